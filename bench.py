#!/usr/bin/env python3
"""bench.py -- M-Gaussians/s for the RAHT hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the codec's transform path over one scene that is already resident in HBM:
forward RAHT -> quantize + reorder -> dequantize + un-reorder -> inverse RAHT (BASELINE.json
configs[2]: "~3M Gaussians, SH deg 3 (59 ch), fwd+inv + quantize").  `--no-quant` times fwd+inv
only.  N > 1: every rank owns one Morton-prefix shard of an N-times larger scene (weak scaling);
the top three octree levels are stitched with one small all-gather over RCCL per direction.
`--workload cfg4` (BASELINE.json configs[3]): every rank codes its own scene of 1-6 M Gaussians,
no collective on the data path (replicas only).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
HIP-event timed, algorithmic bytes) and `cpu_baseline` (the C oracle on the host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-steps", type=int, default=-1,
                    help="untimed steps BEFORE the warmup so that the GPU's clocks have settled when the warmup starts "
                         "(-1: as many as bring settle + warmup to 64; the first ~40 steps after idle run up to 17 %% slower, "
                         "tools/probe_step_transient.py); reported as settle_steps")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg4", "cfg5"],
                    help="cfg3 = headline (3M x 59); cfg2 = 1M x 14; cfg4 = one independent 1-6M x 59 scene per rank; "
                         "cfg5 = 50M x 59 single-GPU equivalent, generated on device")
    ap.add_argument("--no-quant", action="store_true", help="time forward + inverse only")
    ap.add_argument("--engine", default="tile", choices=["tile", "level"])
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--pooled-buffers", type=int, default=0, help="1: carve C/T/Q buffers from one allocation, 64 MiB apart")
    ap.add_argument("--tail-rows", type=int, default=0, help="rows per tile of the stages >= 1 (0 = automatic)")
    ap.add_argument("--tail-ch", type=int, default=0, help="channels per chunk of the stages >= 1 (0 = automatic)")
    ap.add_argument("--top-rows", type=int, default=0, help="entries at which the single-launch top stage takes over (0 = automatic)")
    ap.add_argument("--quant-step", type=float, default=0.01)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-prelude", action="store_true", help="do not time plan build / sort / voxelizer")
    ap.add_argument("--cpu-repeats", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (gloo: ranks may share one GPU; testing only)")
    ap.add_argument("--unfused", action="store_true", help="quantize / dequantize as separate passes")
    ap.add_argument("--ablate", type=int, default=0, help="kernel-timing experiment for the roofline probe only (0 = real kernel)")
    return ap.parse_args()


def cpu_baseline(V, C32, J, repeats):
    """The C oracle (scalar restatement of the reference, float64 like the reference) on this host:
    one core, and all cores (the transform is independent per channel: one thread per channel block,
    ctypes releases the GIL). -> (single-core M-G/s, s per pass, all-cores M-G/s, s per pass, threads, err)"""
    import threading
    from oracle import oracle as orc
    orc.lib()
    p = orc.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
    C64 = C32.astype(np.float64)
    best = None
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        T, _ = orc.raht_fwd(C64, p)
        R = orc.raht_inv(T, p)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    err = float(np.abs(R - C64).max())
    del T, R
    D = C64.shape[1]
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    nthr = max(1, min(ncpu, D, 16))        # every thread re-walks the plan's lists: more, narrower blocks stop paying
    cuts = [round(i * D / nthr) for i in range(nthr + 1)]
    blocks = [np.ascontiguousarray(C64[:, cuts[i]:cuts[i + 1]]) for i in range(nthr)]

    def work(b):
        Tb, _ = orc.raht_fwd(b, p)
        orc.raht_inv(Tb, p)
    best_all = None
    for _ in range(max(1, repeats)):
        th = [threading.Thread(target=work, args=(b,)) for b in blocks]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        best_all = dt if best_all is None else min(best_all, dt)
    n = V.shape[0]
    return n / best / 1e6, best, n / best_all / 1e6, best_all, nthr, err


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch through torch.distributed.run", file=sys.stderr)
        if world == 1 and a.gpus > 1:
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import _lib, synth
    L = _lib.lib()                                     # fails loudly if the HIP library is missing

    n_draws, J, D, seed = synth.CONFIGS[a.workload]
    solo = world == 1 or a.workload == "cfg4"          # this rank runs the whole transform of its own scene
    if a.workload == "cfg4":
        n_draws, seed = synth.CFG4_DRAWS[rank % len(synth.CFG4_DRAWS)], seed + rank
    # ---- synthetic scene (host, seeded), one Morton-prefix shard per rank ----
    if a.workload == "cfg5":
        # 50 M rows: generated on the device (host generation would take minutes); no CPU baseline
        assert world == 1, "cfg5 is a single-GPU scaling data point"
        g5 = torch.Generator(device=dev); g5.manual_seed(seed)
        kraw = torch.randint(0, 1 << (3 * J), (int(n_draws * 1.002),), device=dev, dtype=torch.int64, generator=g5)
        kd5 = torch.unique(kraw)[:n_draws].contiguous()
        del kraw
        keys = None; V = None
        Ch = None
        a.skip_cpu_baseline = True; a.skip_prelude = True
    elif solo:
        V, keys, Ch = synth.scene(n_draws, J, D, seed)
    else:
        per = 512 // world
        V, keys, Ch = synth.scene(n_draws, J, D, seed + 100 * rank, prefix_range=(rank * per, (rank + 1) * per, 9))
    if a.workload == "cfg5":
        N = int(kd5.shape[0])
        kd = kd5
        Cd = torch.empty((N, D), dtype=torch.float32, device=dev)
        for c0 in range(0, D, 8):                               # fill in column blocks: bounded temporaries
            Cd[:, c0:c0 + 8] = torch.randn((N, min(8, D - c0)), device=dev, generator=g5)
    else:
        N = V.shape[0]
        Cd = torch.from_numpy(Ch).to(dev)
        kd = torch.from_numpy(keys.view(np.int64)).to(dev)
    steps_arr = (C.c_float * 1)(a.quant_step)

    if solo:
        plan = R.RahtPlan.from_keys(kd, 3 * J)
        plan.set_engine(a.engine, a.tile_rows, a.tail_rows, a.tail_ch, a.top_rows)
        if a.pooled_buffers:
            # one allocation, buffers 64 MiB apart (buffer placement: the duration of a
            # streaming kernel has a bump over a window of input->output distances that moves with the
            # physical mapping; inside ONE allocation gaps >= 32 MiB were outside it on every box tried)
            nb = N * D * 4
            stride = ((nb + (1 << 21) - 1) >> 21 << 21) + (64 << 20)
            pool = torch.empty(5 * stride, dtype=torch.uint8, device=dev)
            al = (-pool.data_ptr()) % (1 << 21)
            def carve(i, dt):
                return pool[al + i * stride: al + i * stride + nb].view(dt).view(N, D)
            C0 = Cd
            Cd = carve(0, torch.float32); Cd.copy_(C0); del C0
            T, Q, Td, Crec = carve(1, torch.float32), carve(2, torch.int32), carve(3, torch.float32), carve(4, torch.float32)
        else:
            T = torch.empty_like(Cd)
            Q = torch.empty((N, D), dtype=torch.int32, device=dev)
            Td = torch.empty_like(Cd)
            Crec = torch.empty_like(Cd)
        h = plan._h
        vp = C.c_void_p

        def s_():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def fwd():
            _lib.check(L.raht_fwd(h, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, s_()))

        def quant():
            _lib.check(L.raht_quant_reorder(h, vp(T.data_ptr()), D, D, steps_arr, 1, vp(Q.data_ptr()), D, s_()))

        def dequant():
            _lib.check(L.raht_dequant_unreorder(h, vp(Q.data_ptr()), D, D, steps_arr, 1, vp(Td.data_ptr()), D, s_()))

        def inv(src):
            _lib.check(L.raht_inv(h, vp(src.data_ptr()), D, D, vp(Crec.data_ptr()), D, s_()))

        def fwd_quant():
            _lib.check(L.raht_fwd_quant(h, vp(Cd.data_ptr()), D, D, steps_arr, 1, vp(Q.data_ptr()), D, s_()))

        def dequant_inv():
            _lib.check(L.raht_dequant_inv(h, vp(Q.data_ptr()), D, D, steps_arr, 1, vp(Crec.data_ptr()), D, s_()))

        if a.no_quant:
            def step():
                fwd(); inv(T)
        elif a.unfused:
            def step():
                fwd(); quant(); dequant(); inv(Td)
        else:
            def step():
                fwd_quant(); dequant_inv()
        total_rows = N
    else:
        from raht_3dgs_codec_amd import sharded
        sh = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9)
        qs = None if a.no_quant else a.quant_step

        def step():
            sh.step(Cd, qs)
        total_rows = None

    # ---- correctness gate: never report a number for a wrong transform ----
    if solo:
        fwd(); inv(T)
        torch.cuda.synchronize()
        rt_err = (Crec - Cd).abs().max().item() / Cd.abs().max().item()
        assert rt_err <= 1e-5, f"round trip error {rt_err}"
    else:
        rt_err = sh.roundtrip_error(Cd)
        assert rt_err <= 1e-5, f"round trip error {rt_err}"

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    settle = a.settle_steps if a.settle_steps >= 0 else max(0, 64 - a.warmup)
    for _ in range(settle):
        step()
    for _ in range(a.warmup):
        step()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt, float(N)], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = tmax[0].item()
        total_rows = int(tsum[1].item())
    ms_per_step = dt / a.steps * 1e3
    value = total_rows / (dt / a.steps) / 1e6

    out = {
        "metric": "M-Gaussians/s fwd+inv RAHT, 59-ch SH3 3DGS" if D == 59 else "M-Gaussians/s fwd+inv RAHT, 14-ch SH0 3DGS",
        "value": round(value, 2), "unit": "M-Gaussians/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "settle_steps": settle,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"{a.workload}: {total_rows} Gaussians ({'1-6 M' if a.workload == 'cfg4' and world > 1 else n_draws} draws/GPU, J={J}, {D} channels), "
                         + ("fwd + inv RAHT" if a.no_quant else "fwd RAHT + quantize/reorder + dequantize/un-reorder + inv RAHT"
                            + (" (separate passes)" if a.unfused else " (quantization fused into the transform kernels)"))),
            "rows_per_gpu": N, "channels": D, "depth_J": J, "engine": a.engine, "quantize": not a.no_quant,
            "parallelism": "1 GPU" if world == 1 else f"{world} independent scenes, one per GPU, no collective" if solo else f"morton-prefix sharded x{world}, top-3-octree-level all-gather ({'RCCL' if a.backend == 'nccl' else 'gloo, TEST ONLY'})",
            "roundtrip_rel_err": rt_err,
        },
    }

    if rank == 0 and world == 1:
        # ---- per-stage breakdown (HIP events on the launch stream) ----
        def timed(fn, reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps

        reps = min(max(5, a.steps), 100)
        br = {"fwd_ms": timed(fwd, reps), "inv_ms": timed(lambda: inv(T), reps)}
        if not a.no_quant:
            br["quant_reorder_ms"] = timed(quant, reps)
            br["dequant_unreorder_ms"] = timed(dequant, reps)
            br["fwd_quant_fused_ms"] = timed(fwd_quant, reps)
            br["dequant_inv_fused_ms"] = timed(dequant_inv, reps)
        out["breakdown_ms"] = {k: round(v, 4) for k, v in br.items()}
        st = plan.stage_stats(4, D)
        out["config"]["tile_rows"] = st["tile_rows"]
        out["config"]["active_rows_per_stage"] = st["rows_per_stage"]

        # ---- roofline of the dominant kernel: stage-0 LDS-tile kernel, forward and inverse ----
        # ALGORITHMIC bytes per launch (per direction): read N*D*4 + write N*D*4 + 8 B/row of plan
        alg = 8.0 * N * D + 8.0 * N
        if a.engine == "tile":
            fused = not (a.no_quant or a.unfused)
            qp, qs = (vp(Q.data_ptr()), a.quant_step) if fused else (None, 0.0)

            def k_fwd():
                _lib.check(L.raht_debug_run_stage(h, 0, 0, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, qp, D, qs,
                                                  a.ablate, s_()))

            def k_inv():
                _lib.check(L.raht_debug_run_stage(h, 1, 0, vp(T.data_ptr()), D, D, vp(Crec.data_ptr()), D, qp, D, qs,
                                                  a.ablate, s_()))
            k_fwd(); k_inv()
            tf_iso, ti_iso = timed(k_fwd, reps), timed(k_inv, reps)
            tf, ti = tf_iso, ti_iso
            if a.ablate == 0:
                # The same two kernels timed INSIDE real steps: the library records a HIP event pair on the
                # launch stream around the stage-0 launch of each direction (raht_plan_set_stage0_events).
                hip = C.CDLL("libamdhip64.so")
                hip.hipEventCreate.argtypes = [C.POINTER(vp)]
                hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
                hip.hipEventDestroy.argtypes = [vp]

                def new_event():
                    e = vp()
                    assert hip.hipEventCreate(C.byref(e)) == 0
                    return e
                nrep = min(max(5, a.steps), 100)
                evs = [[new_event() for _ in range(4)] for _ in range(nrep)]
                one_fwd, one_inv = ((fwd, lambda: inv(T)) if a.no_quant else
                                    (fwd, lambda: inv(Td)) if a.unfused else (fwd_quant, dequant_inv))
                for e4 in evs:
                    _lib.check(L.raht_plan_set_stage0_events(h, e4[0], e4[1])); one_fwd()
                    if a.unfused and not a.no_quant:
                        _lib.check(L.raht_plan_set_stage0_events(h, None, None)); quant(); dequant()
                    _lib.check(L.raht_plan_set_stage0_events(h, e4[2], e4[3])); one_inv()
                _lib.check(L.raht_plan_set_stage0_events(h, None, None))
                torch.cuda.synchronize()
                ms = C.c_float()
                acc = [0.0, 0.0]
                for e4 in evs:
                    for d_ in (0, 1):
                        assert hip.hipEventElapsedTime(C.byref(ms), e4[2 * d_], e4[2 * d_ + 1]) == 0
                        acc[d_] += ms.value
                    for e in e4:
                        hip.hipEventDestroy(e)
                tf, ti = acc[0] / nrep, acc[1] / nrep
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    traffic = json.load(open(tp)).get(a.workload, {}).get("fused" if fused else "plain", {}).get("fwd_stage0_bytes")
                except Exception:
                    traffic = None
            tq = "true" if fused else "false"
            out["roofline"] = {"kernel": f"raht::tile_kernel<float, false, true, {tq}, 1> (forward, stage 0"
                                         + (", fused quantize+reorder)" if fused else ")"), "bound": "hbm",
                               "achieved": round(alg / (tf * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "alg_bytes_per_launch": alg, "avg_launch_ms": round(tf, 4),
                               "timed": "HIP events around the stage-0 launch inside real steps" if a.ablate == 0 else "isolated launches",
                               "isolated_launch_ms": round(tf_iso, 4)}
            out["roofline_inv"] = {"kernel": f"raht::tile_kernel<float, true, true, {tq}, 1> (inverse, stage 0"
                                             + (", fused un-reorder+dequantize)" if fused else ")"), "bound": "hbm",
                                   "achieved": round(alg / (ti * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(alg / (ti * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "alg_bytes_per_launch": alg, "avg_launch_ms": round(ti, 4),
                                   "isolated_launch_ms": round(ti_iso, 4)}
        # whole fwd+inv against the whole-path algorithmic bytes (16 N D + 16 N)
        tot = br["fwd_ms"] + br["inv_ms"]
        out["path_hbm"] = {"alg_bytes_fwd_inv": 2 * alg, "fwd_inv_ms": round(tot, 4),
                           "achieved_GBs": round(2 * alg / (tot * 1e-3) / 1e9, 1),
                           "frac_of_peak": round(2 * alg / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "fwd_inv_only_MGs": round(N / (tot * 1e-3) / 1e6, 1)}

        # ---- prelude stages, reported separately: plan build, radix sort, voxelizer ----
        if not a.skip_prelude:
            def wall(fn, reps=3):
                best = None
                for _ in range(reps):
                    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
                    dtt = time.perf_counter() - t
                    best = dtt if best is None else min(best, dtt)
                return best * 1e3
            pre = {}
            # plan arrays + the default tile schedule (built speculatively at creation) + destroy
            pre["plan_from_sorted_keys_ms"] = wall(lambda: R.RahtPlan.from_keys(kd, 3 * J))
            g = torch.Generator(device=dev); g.manual_seed(1)
            perm = torch.randperm(N, device=dev, generator=g)
            ku = kd[perm].contiguous()
            pre["radix_sort_%dbit_ms" % (3 * J)] = wall(lambda: R.sort_keys(ku, nbits=3 * J))
            k60 = (ku << (60 - 3 * J)) | (ku & ((1 << (60 - 3 * J)) - 1))
            pre["radix_sort_60bit_ms"] = wall(lambda: R.sort_keys(k60, nbits=60))
            pre["torch_sort_int64_ms"] = wall(lambda: torch.sort(ku))
            # voxelizer on the unsorted cloud: xyz (voxel centres) + the D-3 (or D) attribute columns
            xyz = torch.from_numpy(V.astype(np.float32)).to(dev)[perm] + 0.5
            PC = torch.cat([xyz, Cd[perm][:, : min(D, 56)]], dim=1).contiguous()
            pre["voxelize_ms"] = wall(lambda: R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev, residuals=False, sorted_points=False))
            pre["voxelize_points"] = N
            pre["voxelize_columns"] = int(PC.shape[1])
            out["prelude_ms"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pre.items()}
            del PC, xyz, ku, k60, perm

        if not a.skip_cpu_baseline:
            v1, s1, va, sa, nthr, err = cpu_baseline(V, Ch, J, a.cpu_repeats)
            out["cpu_baseline"] = {"value": round(va, 4), "unit": "M-Gaussians/s", "cores": nthr, "kind": "port",
                                   "sample": (f"the full {a.workload} scene ({N} rows x {D} ch), fwd+inv RAHT only, float64, "
                                              f"best of {a.cpu_repeats}; scalar C oracle (oracle/raht_oracle.c), one thread per "
                                              f"channel block on {nthr} threads: {sa:.2f} s per pass; on one core {s1:.2f} s"),
                                   "single_core_value": round(v1, 4), "roundtrip_abs_err": err}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
