#!/usr/bin/env python3
"""bench.py -- M-Gaussians/s for the RAHT hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the codec's transform path over one scene that is already resident in HBM:
forward RAHT -> quantize + reorder -> dequantize + un-reorder -> inverse RAHT (BASELINE.json
configs[2]: "~3M Gaussians, SH deg 3 (59 ch), fwd+inv + quantize").  `--no-quant` times fwd+inv only.

Workloads (`--workload`):
  cfg3 (default)  N = 1: the headline 3 M x 59 scene. N > 1: WEAK scaling -- every rank owns one 3 M-row
                  Morton-prefix shard of an N-times larger scene; the top three octree levels are stitched
                  with one <= 121 KB all-gather over RCCL per direction.
  cfg2            1 M x 14 (BASELINE configs[1]).
  cfg4            BASELINE configs[3]: every rank codes its own scene of 1-6 M Gaussians, no collective on
                  the data path (replicas only).
  cfg5            BASELINE configs[4]: ONE 50 M-Gaussian scene. N = 1: the whole scene on one GPU. N > 1:
                  STRONG scaling -- the same scene (same seed on every rank) cut into N Morton-prefix shards
                  by `balanced_prefix_cuts`, one shard per rank, top-3-level all-gather per direction.

Correctness gate: no value is printed for a wrong transform. N = 1: the float32 forward coefficients and
the fused quantized integers are compared with the CPU oracle on the WHOLE scene (the float64 restatement
of the reference's RAHT.py; the same run is the cpu_baseline measurement). N > 1: every rank compares its
shard of the sharded transform with the same rows of an unsharded transform of the gathered scene.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel, HIP-event
timed, algorithmic bytes), `cpu_baseline` (the C oracle on the host cores), `oracle_gate`, the reference-
precision leg `f64`, the `cfg2` leg, and `prelude` (plan build / radix sort / voxelizer with their own
algorithmic-bytes fractions).
"""
import argparse
import ctypes as C
import glob
import hashlib
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
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg4", "cfg5"])
    ap.add_argument("--rows", type=int, default=0, help="override the workload's number of draws (rehearsals of the multi-rank modes on small scenes)")
    ap.add_argument("--no-quant", action="store_true", help="time forward + inverse only")
    ap.add_argument("--engine", default="tile", choices=["tile", "level"])
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--pooled-buffers", type=int, default=0, help="1: carve C/T/Q buffers from one allocation, 64 MiB apart")
    ap.add_argument("--tail-rows", type=int, default=0, help="rows per tile of the stages >= 1 (0 = automatic)")
    ap.add_argument("--tail-ch", type=int, default=0, help="channels per chunk of the stages >= 1 (0 = automatic)")
    ap.add_argument("--top-rows", type=int, default=0, help="entries at which the single-launch top stage takes over (0 = automatic)")
    ap.add_argument("--quant-step", type=float, default=0.01)
    ap.add_argument("--skip-cpu-baseline", action="store_true", help="skip the single-core timing and the repeats of the oracle (the gate still runs it once)")
    ap.add_argument("--skip-oracle-gate", action="store_true", help="profiling runs only: the JSON line then says oracle_gate: skipped")
    ap.add_argument("--skip-prelude", action="store_true", help="do not time plan build / sort / voxelizer")
    ap.add_argument("--skip-legs", action="store_true", help="do not run the extra f64 and cfg2 legs")
    ap.add_argument("--cpu-repeats", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (gloo: ranks may share one GPU; testing only)")
    ap.add_argument("--sharded", action="store_true",
                    help="--gpus 1 only: run the Morton-prefix sharded path (ShardedRaht) in a ONE-rank process group of --backend, "
                         "collectives included -- what a one-GPU box can show of the multi-GPU step")
    ap.add_argument("--batch-scenes", type=int, default=0,
                    help="--workload cfg4 --gpus 1: BASELINE configs[3] on ONE GPU -- this many of its 8 scenes (1-6 M Gaussians each) as a batch "
                         "(raht_fwd_quant_batch + raht_dequant_inv_batch: stage k of all scenes in one launch) against one call per scene")
    ap.add_argument("--direct", action="store_true",
                    help="--gpus > 1 / --sharded: the two all-gathers of a step as direct writes into the peers' buffers over hipIpc "
                         "(raht_xchg_*, one launch per direction) instead of the collective backend's all_gather_into_tensor")
    ap.add_argument("--unfused", action="store_true", help="quantize / dequantize as separate passes")
    ap.add_argument("--f32-only", action="store_true",
                    help="frames with xyz columns (14 / 59 channels): time the all-float32 fused kernels instead of the mixed-precision ones "
                         "(raht_fwd_quant_mixed: xyz columns in float64, the reference's integers there)")
    ap.add_argument("--ablate", type=int, default=0, help="kernel-timing experiment for the roofline probe only (0 = real kernel; needs a make ABLATE=1 library)")
    return ap.parse_args()


def kernel_source_hash():
    """sha1 over the kernel sources: profiles/traffic.json records the hash of the build its PMC passes measured,
    and the bench only repeats those HBM bytes next to live timings when the hash is that of the library it runs."""
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "raht-3dgs-codec_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "raht-3dgs-codec_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "raht.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def oracle_pass(V, C32, J, repeats, single_core):
    """The C oracle (scalar restatement of the reference, float64 like the reference) on this host, one thread per
    channel block on every core (oracle/threaded.py) and optionally on one core.
    -> dict(param, T (float64 forward coefficients), all-cores / single-core seconds per fwd+inv pass, threads, err)"""
    from oracle import oracle as orc
    from oracle import threaded
    orc.lib()
    p = orc.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
    C64 = C32.astype(np.float64)
    nthr = threaded.host_threads(C64.shape[1])
    blocks = threaded.split(C64, nthr)
    best_all, Ts, Rs = None, None, None
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        Ts, Rs = threaded.fwd_inv_blocks(orc, blocks, p)
        dt = time.perf_counter() - t0
        best_all = dt if best_all is None else min(best_all, dt)
    err = max(float(np.abs(r - b).max()) for r, b in zip(Rs, blocks))
    T = np.concatenate(Ts, axis=1)
    del Rs, Ts, blocks
    best_one = None
    if single_core:
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            T1, _ = orc.raht_fwd(C64, p)
            orc.raht_inv(T1, p)
            dt = time.perf_counter() - t0
            best_one = dt if best_one is None else min(best_one, dt)
        del T1
    return dict(orc=orc, param=p, T=T, s_all=best_all, s_one=best_one, threads=nthr, err=err)


def oracle_gate(ob, T32, Q32, step, mixed=False):
    """float32 coefficients / fused integers of the HIP path against the oracle on the whole scene. Raises on failure."""
    To = ob["T"]
    N, D = To.shape
    err = np.zeros(D); se = np.zeros(D); colmax = np.zeros(D); sq = np.zeros(D)
    for r0 in range(0, N, 1 << 19):
        b = To[r0:r0 + (1 << 19)]
        d = T32[r0:r0 + (1 << 19)].astype(np.float64) - b
        err = np.maximum(err, np.abs(d).max(axis=0)); se += (d * d).sum(axis=0)
        colmax = np.maximum(colmax, np.abs(b).max(axis=0)); sq += (b * b).sum(axis=0)
    rel = float((err / np.maximum(colmax, 1e-30)).max())
    rms = float((np.sqrt(se / N) / np.maximum(np.sqrt(sq / N), 1e-30)).max())
    g = {"kind": "oracle (float64 C restatement of RAHT.py, whole scene)", "rows": N, "channels": D,
         "max_rel_err_T": rel, "rms_rel_err_T": rms, "tolerance": "per column: max <= 2e-6 * colmax, rms <= 1e-6 * rms"}
    if not (rel <= 2e-6 and rms <= 1e-6):
        raise AssertionError(f"oracle gate: float32 coefficients out of tolerance {g}")
    if Q32 is not None:
        order = ob["param"].order
        Qo = ob["orc"].quant_reorder(To, step, order)
        dq = np.abs(Q32.astype(np.int64) - Qo.astype(np.int64))
        dT = np.abs(T32.astype(np.float64) - To)[order]
        lim = 1.0 + (dT + 1.2e-7 * np.abs(To)[order]) / step
        nbad = int((dq > lim).sum())
        a0 = 3 if D in (14, 59) else 0
        # attribute channels: a differing integer must be +-1, and there may only be as many of them as the float32
        # coefficient error predicts (a rounding boundary between the two quotients: probability |dT| / step each) --
        # "every integer off by one" (a lost +0.5, a wrong rounding direction) fails here, not only in the tests
        nz = dq[:, a0:] != 0
        expected = float(dT[:, a0:].sum() / step)
        g["q_step"] = step
        g["q_mismatch_rate_attr_channels"] = float(nz.mean())
        g["q_mismatches_attr_channels"] = int(nz.sum())
        g["q_mismatches_predicted_from_coefficient_error"] = round(expected, 1)
        g["q_beyond_coefficient_error_bound"] = nbad
        if a0:
            g["q_xyz_columns_max_abs_diff"] = int(dq[:, :a0].max())
            if mixed:
                # raht_fwd_quant_mixed: the xyz columns are carried in float64 -- the oracle's integers, except where 1-ulp transform
                # noise tips an exact rounding tie of the oracle's own quotient (the float64 kernels' bar)
                bad = np.nonzero(dq[:, :a0])
                q = To[order[bad[0]], bad[1]] / step + 0.5
                off_tie = int((np.abs(q - np.round(q)) > 1e-9 * np.maximum(1.0, np.abs(q))).sum())
                g["q_xyz_path"] = "float64 inside the float32 launches (raht_fwd_quant_mixed, n_wide = 3)"
                g["q_xyz_mismatches"] = int(bad[0].size)
                g["q_xyz_mismatches_off_a_rounding_tie"] = off_tie
                if int(dq[:, :a0].max()) > 1 or off_tie:
                    raise AssertionError(f"oracle gate: mixed-precision xyz integers differ from the oracle: max {int(dq[:, :a0].max())}, {off_tie} off a rounding tie")
            else:
                g["q_xyz_note"] = "xyz coefficients reach |T| / step > 2^24 at small steps: float32 integers cannot be exact there (raht_fwd_quant_mixed carries them in float64)"
        if nbad:
            raise AssertionError(f"oracle gate: {nbad} fused quantized integers differ from the oracle by more than the coefficient error allows")
        if int(dq[:, a0:].max()) > 1 or int(nz.sum()) > 4.0 * expected + 8.0 * np.sqrt(expected) + 10:
            raise AssertionError(f"oracle gate: {int(nz.sum())} attribute-channel integers differ from the oracle (max {int(dq[:, a0:].max())}); "
                                 f"the float32 coefficient error accounts for {expected:.1f}")
    return g


def hip_events():
    hip = C.CDLL("libamdhip64.so")
    vp = C.c_void_p
    hip.hipEventCreate.argtypes = [C.POINTER(vp)]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
    hip.hipEventDestroy.argtypes = [vp]
    return hip


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def wall(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        dtt = time.perf_counter() - t
        best = dtt if best is None else min(best, dtt)
    return best * 1e3


def stage0_times(L, _lib, h, one_fwd, one_inv, nrep, between=None):
    """Average duration (ms) of the stage-0 tile kernel of each direction INSIDE real transforms of plan `h`: the library
    records a caller-supplied HIP event pair on the launch stream around that launch (raht_plan_set_stage0_events)."""
    hip = hip_events()
    vp = C.c_void_p

    def new_event():
        e = vp()
        assert hip.hipEventCreate(C.byref(e)) == 0
        return e
    evs = [[new_event() for _ in range(4)] for _ in range(nrep)]
    for e4 in evs:
        _lib.check(L.raht_plan_set_stage0_events(h, e4[0], e4[1])); one_fwd()
        if between is not None:
            _lib.check(L.raht_plan_set_stage0_events(h, None, None)); between()
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
    return acc[0] / nrep, acc[1] / nrep


def sharded_report(a, sh, Cd, qs, dist, world, rank, dev, L, _lib):
    """What separates shard-local time from the exchange in an N-rank step (every figure = MAX over the ranks, ms):
    the two collectives (events on the compute stream around all_gather_into_tensor: they include the hand-over to and
    from the collective's own stream), the step with the collectives and the replicated top tree left out
    (`local_step_ms`: the truncated local transform of both directions), and rank 0's stage-0 roofline."""
    reps = min(max(5, a.steps), 100)
    sh.time_collectives(True)
    for _ in range(reps):
        sh.step(Cd, qs)
    t_f, t_i = sh.collective_ms()
    sh.time_collectives(False)

    def local():
        sh.local_step(Cd, qs)
    for _ in range(3):
        local()
    t_loc = timed(local, reps)
    t_top = timed(lambda: sh.top_only(Cd.shape[1]), reps)
    N, D = int(Cd.shape[0]), int(Cd.shape[1])
    alg = 8.0 * N * D + 8.0 * N
    if qs is None:
        hold = {}
        one_fwd = lambda: hold.__setitem__("T", sh.forward(Cd))           # noqa: E731
        one_inv = lambda: sh.inverse(hold["T"])                           # noqa: E731
    else:
        hold = {}
        one_fwd = lambda: hold.__setitem__("Q", sh.forward_quant(Cd, qs))  # noqa: E731
        one_inv = lambda: sh.dequant_inverse(hold["Q"], qs)                # noqa: E731
    tf = ti = None
    if sh.plan is not None and sh.plan.stage_stats(4, D)["valid"]:
        tf, ti = stage0_times(L, _lib, sh.plan._h, one_fwd, one_inv, reps)
    vals = torch.tensor([t_f, t_i, t_loc, t_top, tf or 0.0, ti or 0.0], dtype=torch.float64)
    if world > 1:
        v = vals.to(dev) if a.backend == "nccl" else vals
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        vals = v.cpu()
    t_f, t_i, t_loc, t_top, tfm, tim = [float(x) for x in vals.tolist()]
    rep = {"exchange": "direct writes into the peers' gather buffers (raht_xchg_gather over hipIpc), one launch per direction" if a.direct
                       else "all_gather_into_tensor of the collective backend",
           "exchange_status": sh.exchange_status() if a.direct else None,
           "rccl_world": dist.get_world_size() if a.backend == "nccl" else None,
           "collective_backend": dist.get_backend(), "ranks": dist.get_world_size(),
           "collective_ms": {"forward_all_gather": round(t_f, 4), "inverse_all_gather": round(t_i, 4),
                             "timed": "events on the compute stream around all_gather_into_tensor, mean over %d steps, max over ranks" % reps},
           "local_step_ms": round(t_loc, 4), "top_tree_ms": round(t_top, 4),
           "gathered_bytes_per_step": sh.gathered_bytes_per_step(D), "gather_rows": sh.gather_rows, "slot_rows": sh.slot,
           "what": "local_step_ms = truncated shard-local fwd(+quant) + (dequant+)inv without collectives and top tree; "
                   "top_tree_ms = the replicated <= 512-row top tree, both directions, with the root quantize/dequantize launches"}
    if tf:
        rep["roofline_stage0"] = {"bound": "hbm", "alg_bytes_per_launch": alg, "rows": N,
                                  "fwd_ms": round(tfm, 4), "inv_ms": round(tim, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "fwd_frac": round(alg / (tfm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "inv_frac": round(alg / (tim * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "note": "slowest rank's stage-0 launch inside real sharded steps; alg bytes of rank 0's shard"}
    return rep


class SoloScene:
    """One scene on one GPU: plan + buffers + the step functions, all through the C ABI."""

    def __init__(self, R, L, _lib, kd, Cd, nbits, a, dev, dtype=torch.float32):
        self.L, self._lib, self.dev = L, _lib, dev
        self.N, self.D = int(Cd.shape[0]), int(Cd.shape[1])
        N, D = self.N, self.D
        self.plan = R.RahtPlan.from_keys(kd, nbits)
        self.plan.set_engine(a.engine, a.tile_rows, a.tail_rows, a.tail_ch, a.top_rows)
        self.f64 = dtype == torch.float64
        es = 8 if self.f64 else 4
        if a.pooled_buffers and not self.f64:
            # one allocation, buffers 64 MiB apart (buffer placement)
            nb = N * D * 4
            stride = ((nb + (1 << 21) - 1) >> 21 << 21) + (64 << 20)
            pool = torch.empty(5 * stride, dtype=torch.uint8, device=dev)
            al = (-pool.data_ptr()) % (1 << 21)

            def carve(i, dt):
                return pool[al + i * stride: al + i * stride + nb].view(dt).view(N, D)
            C0 = Cd
            Cd = carve(0, torch.float32); Cd.copy_(C0); del C0
            self.T, self.Q, self.Td, self.Crec = carve(1, torch.float32), carve(2, torch.int32), carve(3, torch.float32), carve(4, torch.float32)
        else:
            self.T = torch.empty_like(Cd)
            self.Q = torch.empty((N, D), dtype=torch.int32, device=dev)
            self.Td = torch.empty_like(Cd)
            self.Crec = torch.empty_like(Cd)
        self.Cd = Cd
        self.es = es
        self.steps32 = (C.c_float * 1)(a.quant_step)
        self.steps64 = (C.c_double * 1)(a.quant_step)
        # frames that carry their xyz columns (python/voxelize_pc.py:155): those three channels in float64 inside the float32
        # launches (raht_fwd_quant_mixed / raht_dequant_inv_mixed) -- the reference's integers there, where float32 cannot hold them
        self.n_wide = 3
        self.mixed = (not self.f64 and D in (14, 59) and a.engine == "tile" and not getattr(a, "f32_only", False)
                      and self.plan.mixed_stats(D, self.n_wide)["tile_rows"] > 0)

    def s_(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # each of these enqueues kernels through the C ABI only
    def fwd(self):
        f = self.L.raht_fwd_f64 if self.f64 else self.L.raht_fwd
        vp = C.c_void_p
        self._lib.check(f(self.plan._h, vp(self.Cd.data_ptr()), self.D, self.D, vp(self.T.data_ptr()), self.D, None, self.s_()))

    def inv(self, src=None):
        f = self.L.raht_inv_f64 if self.f64 else self.L.raht_inv
        vp = C.c_void_p
        src = self.T if src is None else src
        self._lib.check(f(self.plan._h, vp(src.data_ptr()), self.D, self.D, vp(self.Crec.data_ptr()), self.D, self.s_()))

    def quant(self):
        vp = C.c_void_p
        if self.f64:
            self._lib.check(self.L.raht_quant_reorder_f64(self.plan._h, vp(self.T.data_ptr()), self.D, self.D, self.steps64, 1, vp(self.Q.data_ptr()), self.D, self.s_()))
        else:
            self._lib.check(self.L.raht_quant_reorder(self.plan._h, vp(self.T.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Q.data_ptr()), self.D, self.s_()))

    def dequant(self):
        vp = C.c_void_p
        if self.f64:
            self._lib.check(self.L.raht_dequant_unreorder_f64(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps64, 1, vp(self.Td.data_ptr()), self.D, self.s_()))
        else:
            self._lib.check(self.L.raht_dequant_unreorder(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Td.data_ptr()), self.D, self.s_()))

    def fwd_quant(self):
        vp = C.c_void_p
        if self.f64:
            self._lib.check(self.L.raht_fwd_quant_f64(self.plan._h, vp(self.Cd.data_ptr()), self.D, self.D, self.steps64, 1, vp(self.Q.data_ptr()), self.D, self.s_()))
            return
        if self.mixed:
            self._lib.check(self.L.raht_fwd_quant_mixed(self.plan._h, vp(self.Cd.data_ptr()), self.D, self.D, self.steps64, 1, self.n_wide, vp(self.Q.data_ptr()), self.D, self.s_()))
            return
        self._lib.check(self.L.raht_fwd_quant(self.plan._h, vp(self.Cd.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Q.data_ptr()), self.D, self.s_()))

    def fwd_quant_f32(self):
        vp = C.c_void_p
        self._lib.check(self.L.raht_fwd_quant(self.plan._h, vp(self.Cd.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Q.data_ptr()), self.D, self.s_()))

    def dequant_inv_f32(self):
        vp = C.c_void_p
        self._lib.check(self.L.raht_dequant_inv(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Crec.data_ptr()), self.D, self.s_()))

    def dequant_inv(self):
        vp = C.c_void_p
        if self.f64:
            self._lib.check(self.L.raht_dequant_inv_f64(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps64, 1, vp(self.Crec.data_ptr()), self.D, self.s_()))
            return
        if self.mixed:
            self._lib.check(self.L.raht_dequant_inv_mixed(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps64, 1, self.n_wide, vp(self.Crec.data_ptr()), self.D, self.s_()))
            return
        self._lib.check(self.L.raht_dequant_inv(self.plan._h, vp(self.Q.data_ptr()), self.D, self.D, self.steps32, 1, vp(self.Crec.data_ptr()), self.D, self.s_()))

    def step_fn(self, no_quant, unfused):
        if no_quant:
            return lambda: (self.fwd(), self.inv(self.T))
        if unfused:
            return lambda: (self.fwd(), self.quant(), self.dequant(), self.inv(self.Td))
        return lambda: (self.fwd_quant(), self.dequant_inv())


def two_stream_loop(R, L, _lib, kd, Cd, nbits, step, reps=200, streams=None):
    """_two_stream_loop_on with a side stream that does NOT share a hardware queue with the main one. HIP multiplexes streams onto
    a few hardware queues (4 by default, GPU_MAX_HW_QUEUES); two streams on one queue run their kernels in order whatever the
    events say -- the loop then takes LONGER than on one stream (0.244 against 0.225 ms on the reference's shape; with any other
    side stream: 0.191; tools/probe_stream_pairs.py: which pair collides is fixed per process). Three candidates are probed with a
    few steps each and the best one carries the measurement; the probe's figures go into the line."""
    if streams is not None:
        return _two_stream_loop_on(R, L, _lib, kd, Cd, nbits, step, reps, streams)
    main, sides = torch.cuda.Stream(), [torch.cuda.Stream() for _ in range(3)]
    probe = [_two_stream_loop_on(R, L, _lib, kd, Cd, nbits, step, 40, (main, sb), trials=1)["two_streams_ms_per_step"] for sb in sides]
    r = _two_stream_loop_on(R, L, _lib, kd, Cd, nbits, step, reps, (main, sides[int(np.argmin(probe))]))
    r["side_stream_probe_ms"] = probe
    return r


def _two_stream_loop_on(R, L, _lib, kd, Cd, nbits, step, reps, streams, trials=5):
    """The drivers' loop over quantization steps (python/encode_3dgs.py:199-275) with the two directions on two streams -- forward +
    quantize of step s + 1 next to dequantize + inverse of step s, one workspace set per direction
    (raht_plan_set_concurrent_directions) -- against the same float32 fused calls back to back on one stream."""
    p = R.RahtPlan.from_keys(kd, nbits)
    N, D = int(Cd.shape[0]), int(Cd.shape[1])
    Q = [torch.empty((N, D), dtype=torch.int32, device=Cd.device) for _ in range(2)]
    Cr = torch.empty_like(Cd)
    st = (C.c_float * 1)(step)
    vp = C.c_void_p
    sa, sb = streams

    def fwd(q, s):
        _lib.check(L.raht_fwd_quant(p._h, vp(Cd.data_ptr()), D, D, st, 1, vp(q.data_ptr()), D, vp(s.cuda_stream)))

    def inv(q, s):
        _lib.check(L.raht_dequant_inv(p._h, vp(q.data_ptr()), D, D, st, 1, vp(Cr.data_ptr()), D, vp(s.cuda_stream)))

    def timed_loop(body, n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        body(n)
        sa.wait_stream(sb)
        e1.record(sa); e1.synchronize()
        return e0.elapsed_time(e1) / n

    def serial(n):
        for _ in range(n):
            fwd(Q[0], sa); inv(Q[0], sa)

    def overlapped(n):
        ev_f = [torch.cuda.Event() for _ in range(2)]
        ev_i = [torch.cuda.Event() for _ in range(2)]
        fwd(Q[0], sa); ev_f[0].record(sa)
        for i in range(1, n + 1):
            if i >= 2:
                sa.wait_event(ev_i[i % 2])                 # the inverse that read this Q buffer two steps ago
            if i < n:
                fwd(Q[i % 2], sa); ev_f[i % 2].record(sa)
            sb.wait_event(ev_f[(i - 1) % 2])
            inv(Q[(i - 1) % 2], sb); ev_i[(i - 1) % 2].record(sb)
    serial(40)
    t1 = timed_loop(serial, reps)
    ref = Cr.clone()
    p.set_concurrent_directions(True)
    overlapped(40)
    trials = sorted(timed_loop(overlapped, reps) for _ in range(trials))
    t2 = trials[len(trials) // 2]
    torch.cuda.synchronize()
    assert torch.equal(Cr, ref), "two-stream loop reconstructs differently"
    alg = 2 * (8.0 * N * D + 8.0 * N)
    return {"rows": N, "channels": D, "one_stream_ms_per_step": round(t1, 4), "two_streams_ms_per_step": round(t2, 4),
            "two_streams_trials_ms": [round(t, 4) for t in trials],
            "one_stream_frac_of_peak": round(alg / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "two_streams_frac_of_peak": round(alg / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "two_streams_value": round(N / (t2 * 1e-3) / 1e6, 1), "unit": "M-Gaussians/s", "bit_identical": True}


def device_scene(n_draws, J, D, seed, dev):
    """cfg5: sorted unique 3J-bit keys and N(0,1) attributes generated ON THE DEVICE from `seed` (host generation of
    50 M rows takes minutes). Every rank that calls this with the same seed holds the same scene."""
    g5 = torch.Generator(device=dev); g5.manual_seed(seed)
    kraw = torch.randint(0, 1 << (3 * J), (int(n_draws * 1.002),), device=dev, dtype=torch.int64, generator=g5)
    kd = torch.unique(kraw)[:n_draws].contiguous()
    del kraw
    return kd, g5


def device_attributes(N, D, g5, dev, rows=None):
    """(N, D) float32 N(0,1) in column blocks of 8 (bounded temporaries); rows=(lo, hi) keeps only that row range of
    the SAME matrix (a shard of the scene)."""
    lo, hi = (0, N) if rows is None else rows
    Cd = torch.empty((hi - lo, D), dtype=torch.float32, device=dev)
    for c0 in range(0, D, 8):
        blk = torch.randn((N, min(8, D - c0)), device=dev, generator=g5)
        Cd[:, c0:c0 + 8] = blk[lo:hi]
        del blk
    return Cd


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N fresh rank processes (one per GPU) through
    torch.distributed.run as a CHILD of this process and exit with its status. Nothing in this process has touched the GPU
    yet (importing torch does not), and it never does: a process that has initialised HIP must not be replaced or forked.
    Rank 0 of the children prints the JSON line on the inherited stdout."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cfg4_batch_main(a, R, synth, dev, real_stdout):
    """BASELINE configs[3] when there is ONE GPU: its scenes (1-6 M Gaussians, 59 channels, seeds 10 ...) as a batch on it."""
    from raht_3dgs_codec_amd import ops
    n_draws, J, D, seed = synth.CONFIGS["cfg4"]
    K = min(a.batch_scenes, len(synth.CFG4_DRAWS))
    plans, Cs = [], []
    for i in range(K):
        V, keys, Ch = synth.scene(synth.CFG4_DRAWS[i], J, D, seed + i)
        plans.append(R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).to(dev), 3 * J))
        Cs.append(torch.from_numpy(Ch).to(dev))
        del V, keys, Ch
    rows = sum(p.N for p in plans)
    qs = a.quant_step

    def batched():
        return ops.dequant_inverse_batch(plans, ops.forward_quant_batch(plans, Cs, qs), qs)

    def looped():
        return [p.dequant_inverse(p.forward_quant(c, qs), qs) for p, c in zip(plans, Cs)]
    # gate: the batch is bit-identical to one call per scene (whose kernels the oracle checks scene by scene in the default
    # run and in tests/test_gpu_fullsize.py), and every scene survives the float32 round trip
    Qb = ops.forward_quant_batch(plans, Cs, qs)
    assert all(torch.equal(q, p.forward_quant(c, qs)) for q, p, c in zip(Qb, plans, Cs)), "batch != single-scene calls"
    Rb = ops.inverse_batch(plans, ops.forward_batch(plans, Cs))
    rt = max(float(((r - c).abs().max() / c.abs().max()).item()) for r, c in zip(Rb, Cs))
    assert rt <= 1e-5, f"round trip error {rt}"
    del Qb, Rb
    settle = a.settle_steps if a.settle_steps >= 0 else max(0, 64 - a.warmup)
    for _ in range(settle + a.warmup):
        batched()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batched()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_loop = timed(looped, min(a.steps, 20))
    alg = 2 * (8.0 * rows * D + 8.0 * rows)
    out = {"metric": "M-Gaussians/s fwd+inv RAHT, 59-ch SH3 3DGS", "value": round(rows / (dt / a.steps) / 1e6, 2), "unit": "M-Gaussians/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "settle_steps": settle, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"cfg4 on ONE GPU: {K} scenes of {[p.N for p in plans]} Gaussians (J={J}, {D} channels) as a batch, "
                                  "fwd RAHT + quantize/reorder + dequantize/un-reorder + inv RAHT per scene, stage k of all scenes in one launch",
                      "rows_total": rows, "channels": D, "depth_J": J, "parallelism": "1 GPU, batched launches", "roundtrip_rel_err": rt},
           "oracle_gate": {"kind": "batch == one call per scene (bit for bit) + round trip; the single-scene kernels are oracle-checked on whole scenes in the default run"},
           "path_hbm": {"alg_bytes_fwd_inv": alg, "whole_step_frac_of_peak": round(alg / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)},
           "one_call_per_scene_ms": round(t_loop, 4), "one_call_per_scene_value": round(rows / (t_loop * 1e-3) / 1e6, 2)}
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    # stdout carries ONE line, the JSON record. Libraries write banners there too (RCCL: "RCCL version : ...", gloo:
    # "[Gloo] Rank 0 is connected to ..."): file descriptor 1 points at stderr until the record is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if a.sharded and a.gpus != 1:
        raise SystemExit("bench.py: --sharded is the one-rank rehearsal of the multi-GPU path; with --gpus N > 1 the sharded path is the default")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or a.sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                  # --sharded without a launcher: a one-rank group of its own
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(free_port())
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import _lib, synth
    L = _lib.lib()                                     # fails loudly if the HIP library is missing

    if a.workload == "cfg4" and world == 1 and a.batch_scenes > 0:
        return cfg4_batch_main(a, R, synth, dev, real_stdout)
    n_draws, J, D, seed = synth.CONFIGS[a.workload]
    if a.rows > 0:
        n_draws = a.rows
    solo = (world == 1 and not a.sharded) or a.workload == "cfg4"   # this rank runs the whole transform of its own scene
    scaling = "strong" if (a.workload == "cfg5" and world > 1) else "weak"
    if a.workload == "cfg4":
        n_draws, seed = synth.CFG4_DRAWS[rank % len(synth.CFG4_DRAWS)], seed + rank
    V = keys = Ch = None
    shard_rows = None
    if a.workload == "cfg5":
        kd_all, g5 = device_scene(n_draws, J, D, seed, dev)
        if world == 1:
            kd = kd_all
            Cd = device_attributes(int(kd.shape[0]), D, g5, dev)
            shard_rows = (0, int(kd.shape[0]))
        else:
            from raht_3dgs_codec_amd import sharded
            cuts = sharded.balanced_prefix_cuts(kd_all, 3 * J, world, prefix_bits=9)
            shard_rows = (cuts[rank], cuts[rank + 1])
            kd = kd_all[shard_rows[0]:shard_rows[1]].contiguous()
            Cd = device_attributes(int(kd_all.shape[0]), D, g5, dev, rows=shard_rows)
        del kd_all
    elif solo:
        V, keys, Ch = synth.scene(n_draws, J, D, seed)
    else:
        per = 512 // world
        V, keys, Ch = synth.scene(n_draws, J, D, seed + 100 * rank, prefix_range=(rank * per, (rank + 1) * per, 9))
    if a.workload != "cfg5":
        Cd = torch.from_numpy(Ch).to(dev)
        kd = torch.from_numpy(keys.view(np.int64)).to(dev)
    N = int(kd.shape[0])

    gate = None
    sc = None
    if solo:
        sc = SoloScene(R, L, _lib, kd, Cd, 3 * J, a, dev)
        step = sc.step_fn(a.no_quant, a.unfused)
        total_rows = N
        # ---- correctness gate: never report a number for a wrong transform ----
        sc.fwd(); sc.inv(sc.T)
        torch.cuda.synchronize()
        rt_err = (sc.Crec - sc.Cd).abs().max().item() / sc.Cd.abs().max().item()
        assert rt_err <= 1e-5, f"round trip error {rt_err}"
        ob = None
        if V is not None and rank == 0 and not a.skip_oracle_gate:
            ob = oracle_pass(V, Ch, J, 1 if a.skip_cpu_baseline else a.cpu_repeats, single_core=not a.skip_cpu_baseline)
            T32 = sc.T.cpu().numpy()
            Q32 = None
            if not a.no_quant:
                sc.fwd_quant(); torch.cuda.synchronize()
                Q32 = sc.Q.cpu().numpy()
            gate = oracle_gate(ob, T32, Q32, a.quant_step, mixed=sc.mixed)          # raises -> no JSON line
            gate["order_RAGFT_equal"] = bool(np.array_equal(sc.plan.order_RAGFT.cpu().numpy(), ob["param"].order))
            assert gate["order_RAGFT_equal"], "oracle gate: order_RAGFT differs"
            del T32, Q32
            ob["T"] = None
        elif a.skip_oracle_gate:
            gate = {"kind": "skipped (--skip-oracle-gate: profiling run)"}
        elif V is None:
            # 50 M rows: the oracle would need ~50 GB of float64 and minutes; size-independent properties instead
            sc.fwd_quant_f32(); sc.quant(); torch.cuda.synchronize()
            Q2 = sc.Q.clone(); sc.fwd_quant_f32(); torch.cuda.synchronize()
            assert torch.equal(Q2, sc.Q), "fused != two-call"
            if sc.mixed:
                # the mixed kernels: float32 columns bit-identical to the float32 fused kernels; wide columns within one unit of them
                # here (N(0,1) data: quotients far below 2^24) and decoding back to within half a step
                sc.fwd_quant(); torch.cuda.synchronize()
                assert torch.equal(sc.Q[:, sc.n_wide:], Q2[:, sc.n_wide:]), "mixed: float32 columns differ from the float32 kernels"
                assert int((sc.Q[:, :sc.n_wide] - Q2[:, :sc.n_wide]).abs().max()) <= 1, "mixed: wide columns"
            e_in = sum(float((sc.Cd[:, c].double() ** 2).sum()) for c in range(D))
            e_out = sum(float((sc.T[:, c].double() ** 2).sum()) for c in range(D))
            assert abs(e_in - e_out) <= 1e-5 * e_in, "Parseval"
            gate = {"kind": "properties (no oracle at 50 M rows): round trip, Parseval, fused == two-call", "roundtrip_rel_err": rt_err,
                    "note": "the same kernels are oracle-checked on whole 1 M / 3 M / 6 M scenes (tests/test_gpu_fullsize.py) and in the cfg3 run of this bench"}
            del Q2
    else:
        from raht_3dgs_codec_amd import sharded
        sh = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9, force_collectives=True, direct=a.direct)
        qs = None if a.no_quant else a.quant_step

        def step():
            sh.step(Cd, qs)
        total_rows = N
        rt_err = sh.roundtrip_error(Cd)
        assert rt_err <= 1e-5, f"round trip error {rt_err}"
        # sharded == unsharded: gather the scene, transform it whole on this GPU, compare this rank's rows
        if not a.skip_oracle_gate:
            gate = sh.check_against_unsharded(Cd, qs)
            assert gate["ok"], f"sharded transform differs from the unsharded one: {gate}"
        else:
            gate = {"kind": "skipped (--skip-oracle-gate: profiling run)"}

    def barrier():
        if world > 1:
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
    if not solo and a.direct:
        sh.check_exchange()                                # a direct gather that timed out: no number for stale buffers
    if world > 1:
        tt = torch.tensor([dt, float(N)], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = tmax[0].item()
        total_rows = int(tsum[1].item())
    ms_per_step = dt / a.steps * 1e3
    value = total_rows / (dt / a.steps) / 1e6

    quant_txt = ("fwd + inv RAHT" if a.no_quant else "fwd RAHT + quantize/reorder + dequantize/un-reorder + inv RAHT"
                 + (" (separate passes)" if a.unfused else " (quantization fused into the transform kernels)"))
    use_mixed = bool(sc is not None and sc.mixed and not a.no_quant and not a.unfused)
    if use_mixed:
        quant_txt += "; the 3 xyz columns carried in float64 inside the same launches (raht_fwd_quant_mixed / raht_dequant_inv_mixed): the reference's integers on every column"
    if solo:
        par = "1 GPU" if world == 1 else f"{world} independent scenes, one per GPU, no collective"
    else:
        par = (f"ONE scene morton-prefix sharded x{world} (balanced 9-bit prefix cuts)" if scaling == "strong" else f"morton-prefix sharded x{world}, one shard per rank") \
              + f", top-3-octree-level all-gather ({'RCCL' if a.backend == 'nccl' else 'gloo, TEST ONLY'})"
        if world == 1:
            par = f"ONE-rank rehearsal of the sharded path (--sharded): truncated local tree + {'RCCL' if a.backend == 'nccl' else 'gloo'} all-gathers in a group of one + replicated top tree"
    out = {
        "metric": "M-Gaussians/s fwd+inv RAHT, 59-ch SH3 3DGS" if D == 59 else "M-Gaussians/s fwd+inv RAHT, 14-ch SH0 3DGS",
        "value": round(value, 2), "unit": "M-Gaussians/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "settle_steps": settle,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32 (xyz columns: f64)" if use_mixed else "f32", "data": "synthetic",
        "config": {
            "workload": f"{a.workload}: {total_rows} Gaussians ({'1-6 M' if a.workload == 'cfg4' and world > 1 else n_draws} draws{'' if scaling == 'strong' else '/GPU'}, J={J}, {D} channels), " + quant_txt,
            "rows_per_gpu": N, "channels": D, "depth_J": J, "engine": a.engine, "quantize": not a.no_quant,
            "parallelism": par, "world": world, "backend": None if dist is None else ("RCCL" if a.backend == "nccl" else "gloo"),
            "roundtrip_rel_err": rt_err,
        },
        "oracle_gate": gate,
    }
    if not solo:
        out["config"]["gathered_bytes_per_step"] = sh.gathered_bytes_per_step(D)
        out["config"]["roots_per_rank"] = sh.sizes
        out["multi_gpu"] = sharded_report(a, sh, Cd, qs, dist, world, rank, dev, L, _lib)

    if rank == 0 and world == 1 and solo:
        plan, h = sc.plan, sc.plan._h
        vp = C.c_void_p
        # ---- per-stage breakdown (HIP events on the launch stream) ----
        reps = min(max(5, a.steps), 100)
        br = {"fwd_ms": timed(sc.fwd, reps), "inv_ms": timed(lambda: sc.inv(sc.T), reps)}
        if not a.no_quant:
            br["quant_reorder_ms"] = timed(sc.quant, reps)
            br["dequant_unreorder_ms"] = timed(sc.dequant, reps)
            br["fwd_quant_fused_ms"] = timed(sc.fwd_quant, reps)
            br["dequant_inv_fused_ms"] = timed(sc.dequant_inv, reps)
            if sc.mixed:
                # the all-float32 fused kernels on the same scene (xyz integers off the reference's by tens of units at this step)
                br["fwd_quant_fused_f32_only_ms"] = timed(sc.fwd_quant_f32, reps)
                br["dequant_inv_fused_f32_only_ms"] = timed(sc.dequant_inv_f32, reps)
                sc.fwd_quant()                                 # leave the mixed integers in Q
        out["breakdown_ms"] = {k: round(v, 4) for k, v in br.items()}
        st = plan.mixed_stats(D, sc.n_wide) if use_mixed else plan.stage_stats(4, D)
        out["config"]["tile_rows"] = st["tile_rows"]
        out["config"]["active_rows_per_stage"] = st["rows_per_stage"]
        if use_mixed:
            out["config"]["precision"] = {"channels_0_2": "float64 (float32 input widened exactly, float64 butterflies, IEEE double division)", "channels_3_up": "float32", "n_wide": sc.n_wide}
            f32_step = lambda: (sc.fwd_quant_f32(), sc.dequant_inv_f32())
            for _ in range(20):
                f32_step()
            t32 = timed(f32_step, reps)
            sc.fwd_quant()
            out["f32_only"] = {"what": "the same step through the all-float32 fused kernels (raht_fwd_quant + raht_dequant_inv; --f32-only times it as the headline): "
                                       "xyz integers off the reference's by tens of units at this step (round 3's headline)",
                               "ms_per_step": round(t32, 4), "value": round(N / (t32 * 1e-3) / 1e6, 1), "unit": "M-Gaussians/s",
                               "frac_of_peak": round(2 * (8.0 * N * D + 8.0 * N) / (t32 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        # ---- roofline of the dominant kernel: stage-0 LDS-tile kernel, forward and inverse ----
        # ALGORITHMIC bytes per launch (per direction): read N*D*4 + write N*D*4 + 8 B/row of plan
        alg = 8.0 * N * D + 8.0 * N
        if a.engine == "tile":
            fused = not (a.no_quant or a.unfused)
            qp, qs_ = (vp(sc.Q.data_ptr()), a.quant_step) if fused else (None, 0.0)
            # (raht_debug_run_stage launches the float32 kernels: with the mixed kernels as the step, only the in-step timing below
            # describes the dominant kernel; the isolated figure is then the float32 kernel's, for comparison)

            def k_fwd():
                _lib.check(L.raht_debug_run_stage(h, 0, 0, vp(sc.Cd.data_ptr()), D, D, vp(sc.T.data_ptr()), D, qp, D, qs_, a.ablate, sc.s_()))

            def k_inv():
                _lib.check(L.raht_debug_run_stage(h, 1, 0, vp(sc.T.data_ptr()), D, D, vp(sc.Crec.data_ptr()), D, qp, D, qs_, a.ablate, sc.s_()))
            k_fwd(); k_inv()
            tf_iso, ti_iso = timed(k_fwd, reps), timed(k_inv, reps)
            tf, ti = tf_iso, ti_iso
            if a.ablate == 0:
                # The same two kernels timed INSIDE real steps: the library records a HIP event pair on the
                # launch stream around the stage-0 launch of each direction (raht_plan_set_stage0_events).
                one_fwd, one_inv = ((sc.fwd, lambda: sc.inv(sc.T)) if a.no_quant else
                                    (sc.fwd, lambda: sc.inv(sc.Td)) if a.unfused else (sc.fwd_quant, sc.dequant_inv))
                between = (lambda: (sc.quant(), sc.dequant())) if (a.unfused and not a.no_quant) else None
                tf, ti = stage0_times(L, _lib, h, one_fwd, one_inv, min(max(5, a.steps), 100), between)
            # HBM bytes from the PMC counters: only repeated here when profiles/traffic.json was measured on THIS build
            traffic, traffic_note = None, "no profiles/traffic.json entry for this workload"
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp)).get(a.workload, {})
                    if tj.get("source_hash") == kernel_source_hash():
                        traffic = tj.get("fused" if fused else "plain", {}).get("fwd_stage0_bytes")
                        traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build ({tj.get('source')}, source_hash {tj.get('source_hash')})"
                    else:
                        traffic_note = (f"profiles/traffic.json was measured on another build (source_hash {tj.get('source_hash')} != "
                                        f"{kernel_source_hash()}): not repeated next to live timings")
                except Exception:
                    pass
            tq = "true" if fused else "false"
            kf = "raht::tile_kernel_mx<false, true, 1>" if use_mixed else f"raht::tile_kernel<float, false, true, {tq}, 1>"
            ki = "raht::tile_kernel_mx<true, true, 1>" if use_mixed else f"raht::tile_kernel<float, true, true, {tq}, 1>"
            if use_mixed:
                traffic, traffic_note = None, traffic_note if "another build" in traffic_note else "profiles/traffic.json describes the float32 kernels"
                tp2 = os.path.join(ROOT, "profiles", "traffic.json")
                try:
                    tj = json.load(open(tp2)).get(a.workload, {})
                    if tj.get("source_hash") == kernel_source_hash() and "mixed" in tj:
                        traffic = tj["mixed"].get("fwd_stage0_bytes")
                        traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build ({tj.get('source')}, source_hash {tj.get('source_hash')})"
                except Exception:
                    pass
            out["roofline"] = {"kernel": f"{kf} (forward, stage 0"
                                         + (", fused quantize+reorder)" if fused else ")"), "bound": "hbm",
                               "achieved": round(alg / (tf * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(alg / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
                               "alg_bytes_per_launch": alg, "avg_launch_ms": round(tf, 4),
                               "timed": "HIP events around the stage-0 launch inside real steps" if a.ablate == 0 else "isolated launches",
                               "isolated_launch_ms": round(tf_iso, 4), "isolated_launch_is": "the float32 kernel's (raht_debug_run_stage)" if use_mixed else "this kernel's"}
            out["roofline_inv"] = {"kernel": f"{ki} (inverse, stage 0"
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
                           "fwd_inv_only_MGs": round(N / (tot * 1e-3) / 1e6, 1),
                           "whole_step_frac_of_peak": round(2 * alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

        # ---- prelude stages, reported separately, each against its own algorithmic bytes ----
        if not a.skip_prelude and V is not None:
            def roof(ms, alg_b):
                return {"ms": round(ms, 4), "alg_bytes": alg_b, "achieved_GBs": round(alg_b / (ms * 1e-3) / 1e9, 1),
                        "frac_of_peak": round(alg_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm"}
            pre = {}
            # plan arrays + the default tile schedule (built speculatively at creation) + destroy;
            # >= 8 N read (keys) + ~9 N written (lvl, wl, wr)  -> 17 N
            pre["plan_from_sorted_keys"] = roof(wall(lambda: R.RahtPlan.from_keys(kd, 3 * J)), 17.0 * N)
            g = torch.Generator(device=dev); g.manual_seed(1)
            perm = torch.randperm(N, device=dev, generator=g)
            ku = kd[perm].contiguous()
            # one 8-bit radix pass moves ~24 N (key 8 + index 4, read and written)
            for nb_ in (3 * J, 60):
                kk = ku if nb_ == 3 * J else ((ku << (60 - 3 * J)) | (ku & ((1 << (60 - 3 * J)) - 1)))
                pre["radix_sort_%dbit" % nb_] = roof(wall(lambda: R.sort_keys(kk, nbits=nb_)), 24.0 * N * ((nb_ + 7) // 8))
            pre["torch_sort_int64_ms"] = round(wall(lambda: torch.sort(ku)), 4)
            # voxelizer on the unsorted cloud: xyz (voxel centres) + the attribute columns
            xyz = torch.from_numpy(V.astype(np.float32)).to(dev)[perm] + 0.5
            PC = torch.cat([xyz, sc.Cd[perm][:, : min(D, 56)]], dim=1).contiguous()
            ld = int(PC.shape[1])
            # read the cloud once (gathered in sorted order), write PCvox once, plus the key sort
            vox_alg = 4.0 * N * ld * 2 + 24.0 * N * ((3 * J + 7) // 8)
            pre["voxelize"] = roof(wall(lambda: R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev, residuals=False, sorted_points=False)), vox_alg)
            pre["voxelize"]["points"], pre["voxelize"]["columns"] = N, ld
            # ... and with the reference's secondary outputs (PCsorted, DeltaPC: voxelize_pc.py:103-111, 147-156): the cloud
            # gathered once more, PCvox read per point, two N x ld matrices written
            pre["voxelize_with_residuals"] = roof(wall(lambda: R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev)), vox_alg + 4.0 * N * ld * 4)
            # the per-frame prelude of a dynamic sequence in one call: voxelize, then the plan straight from the voxelizer's
            # sorted voxel keys (borrowed, not copied)
            pre["voxelize_plan"] = roof(wall(lambda: R.voxelize_plan(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev)), vox_alg + 17.0 * N)
            pre["plan_from_sorted_keys_borrowed"] = roof(wall(lambda: R.RahtPlan.from_keys(kd, 3 * J, borrow=True)), 9.0 * N + 8.0 * N)
            # what a frame of a dynamic sequence pays in front of its transform, stage by stage (the round-2 review's target: <= 0.9 ms)
            pre["plan_plus_sort_plus_voxelize_ms"] = round(pre["plan_from_sorted_keys"]["ms"] + pre["radix_sort_%dbit" % (3 * J)]["ms"] + pre["voxelize"]["ms"], 4)
            pre["note"] = ("sort: one histogram launch for every digit + one launch per digit pass whose tiles chain their offsets (scan_sort.hip); voxelize: keys "
                           "inside the sort's histogram launch, voxel starts in two launches, one host round trip; voxelize_with_residuals: raht_voxelize_all, "
                           "means + PCsorted + DeltaPC from one pass over the gathered rows; voxelize_plan: raht_voxelize_plan (one call)")
            out["prelude"] = pre
            del PC, xyz, ku, perm

        # ---- cpu_baseline: the oracle run that fed the gate ----
        if ob is not None:
            nthr = ob["threads"]
            cb = {"value": round(N / ob["s_all"] / 1e6, 4), "unit": "M-Gaussians/s", "cores": nthr, "kind": "port",
                  "sample": (f"the full {a.workload} scene ({N} rows x {D} ch), fwd+inv RAHT only, float64, best of {1 if a.skip_cpu_baseline else a.cpu_repeats}; "
                             f"scalar C oracle (oracle/raht_oracle.c), one thread per channel block on {nthr} threads: {ob['s_all']:.2f} s per pass"
                             + (f"; on one core {ob['s_one']:.2f} s" if ob["s_one"] else "")),
                  "roundtrip_abs_err": ob["err"]}
            if ob["s_one"]:
                cb["single_core_value"] = round(N / ob["s_one"] / 1e6, 4)
            # the reference's OWN CPU path (RAHT2_optimized + inverse_RAHT_optimized, float64, torch CPU) on this same scene:
            # a constant measured in the build container by tools/time_reference_cpu.py -- the reference cannot travel here
            rp = os.path.join(ROOT, "profiles", "reference_cpu.json")
            if os.path.exists(rp):
                try:
                    rj = json.load(open(rp))
                    row = rj["configs"].get(a.workload)
                    if row and row["rows"] == N and row["channels"] == D:
                        cb["reference_torch_cpu"] = {"value": row["M_Gaussians_per_s"], "unit": "M-Gaussians/s", "cores": rj["threads"], "kind": "reference",
                                                     "fwd_s": row["RAHT_s"], "inv_s": row["iRAHT_s"], "RAHT_param_s": row["RAHT_param_s"],
                                                     "provenance": f"constant: tools/time_reference_cpu.py in the build container ({rj['cpu']}, {rj['threads']} threads, torch {rj['torch']}, "
                                                                   f"float64, median of {rj['repeats']}, {rj['date']}); python/RAHT.py:252-336 + python/iRAHT.py:40-114 imported unchanged"}
                except Exception:
                    pass
            out["cpu_baseline"] = cb

        # ---- extra legs on the same box: reference precision (float64), and cfg2 ----
        if not a.skip_legs and a.workload == "cfg3" and a.engine == "tile" and not a.unfused:
            kreps = 50
            s64 = SoloScene(R, L, _lib, kd, sc.Cd.double(), 3 * J, a, dev, dtype=torch.float64)
            f_fi, f_q, f_u = s64.step_fn(True, False), s64.step_fn(False, False), s64.step_fn(False, True)
            for _ in range(10):
                f_q()
            # fused float64 integers == two-call float64 integers, on the benched scene
            s64.fwd_quant(); torch.cuda.synchronize(); Qf = s64.Q.clone()
            s64.fwd(); s64.quant(); torch.cuda.synchronize()
            assert torch.equal(Qf, s64.Q), "float64: fused != two-call"
            del Qf
            t_fi, t_q, t_u = timed(f_fi, kreps), timed(f_q, kreps), timed(f_u, 10)
            t_qf, t_qi = timed(s64.fwd_quant, kreps), timed(s64.dequant_inv, kreps)
            alg64 = 2 * (16.0 * N * D + 8.0 * N)
            out["f64"] = {"what": "the reference's own precision (encode_3dgs.py:82-83): raht_fwd_f64 + raht_inv_f64; with_quant = raht_fwd_quant_f64 + raht_dequant_inv_f64 (float64 quantizer fused into the float64 kernels), unfused = the four-pass sequence",
                          "fwd_inv_ms": round(t_fi, 4), "fwd_inv_MGs": round(N / (t_fi * 1e-3) / 1e6, 1), "alg_bytes_fwd_inv": alg64,
                          "frac_of_peak": round(alg64 / (t_fi * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "with_quant_ms": round(t_q, 4), "with_quant_MGs": round(N / (t_q * 1e-3) / 1e6, 1),
                          "with_quant_frac_of_peak": round((alg64 - 8.0 * N * D) / (t_q * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),      # C 8 B in, Q 4 B out, Q 4 B in, C 8 B out per coefficient
                          "fwd_quant_ms": round(t_qf, 4), "dequant_inv_ms": round(t_qi, 4), "unfused_ms": round(t_u, 4)}
            del s64
            def small_leg(name, n2, J2, D2, seed2, what):
                """the same fused step on a smaller scene (latency-bound regime: few rounds of tiles, the tail stages are a
                third of the step)"""
                V2, keys2, C2 = synth.scene(n2, J2, D2, seed2)
                s2 = SoloScene(R, L, _lib, torch.from_numpy(keys2.view(np.int64)).to(dev), torch.from_numpy(C2).to(dev), 3 * J2, a, dev)
                f2 = s2.step_fn(a.no_quant, False)
                for _ in range(64):
                    f2()
                t2 = timed(f2, 200)
                alg2 = 2 * (8.0 * s2.N * D2 + 8.0 * s2.N)
                leg = {"workload": f"{name}: {s2.N} Gaussians, J={J2}, {D2} channels, same step{what}", "ms_per_step": round(t2, 4),
                       "value": round(s2.N / (t2 * 1e-3) / 1e6, 1), "unit": "M-Gaussians/s", "alg_bytes_fwd_inv": alg2,
                       "frac_of_peak": round(alg2 / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "active_rows_per_stage": s2.plan.stage_stats(4, D2)["rows_per_stage"]}
                if s2.mixed and not a.no_quant:
                    # frames with xyz columns: the step above carries them in float64; the all-float32 step beside it
                    f32 = lambda: (s2.fwd_quant_f32(), s2.dequant_inv_f32())                                    # noqa: E731
                    for _ in range(32):
                        f32()
                    t32 = timed(f32, 200)
                    leg["precision"] = "xyz columns float64 inside the float32 launches (mixed)"
                    leg["f32_only"] = {"ms_per_step": round(t32, 4), "frac_of_peak": round(alg2 / (t32 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del s2
                return leg
            n2, J2, D2, seed2 = synth.CONFIGS["cfg2"]
            out["cfg2"] = small_leg("cfg2", n2, J2, D2, seed2, "")
            # the shape the reference's driver actually runs (python/encode_3dgs.py:20-33: J = 10, one frame of ~1 M voxels,
            # 56 attribute channels; 59 with the xyz columns of PCvox)
            out["reference_shape"] = {"what": "python/encode_3dgs.py:20-33: J = 10, ~1 M voxels per frame, 56 attribute channels (59 with xyz)",
                                      "d56": small_leg("ref56", 1_000_000, 10, 56, 1, ""), "d59": small_leg("ref59", 1_000_000, 10, 59, 1, "")}
            # the drivers' step loop with the two directions on two streams (forward of step s + 1 next to the inverse of step s)
            tsl = {"what": "python/encode_3dgs.py:199-275: forward + quantize of step s + 1 does not depend on dequantize + inverse of step s; one workspace "
                           "set per direction (raht_plan_set_concurrent_directions), the two directions on two streams: one direction's latency-bound tail "
                           "stages run under the other's first stage. float32 fused kernels; same reconstructions bit for bit.",
                   "cfg3": two_stream_loop(R, L, _lib, kd, sc.Cd, 3 * J, a.quant_step)}
            for nm, (n_, J_, D_, sd_) in (("reference_shape_d56", (1_000_000, 10, 56, 1)), ("cfg2", synth.CONFIGS["cfg2"])):
                V_, k_, C_ = synth.scene(n_, J_, D_, sd_)
                tsl[nm] = two_stream_loop(R, L, _lib, torch.from_numpy(k_.view(np.int64)).to(dev), torch.from_numpy(C_).to(dev), 3 * J_, a.quant_step)
            out["frame_loop_two_streams"] = tsl
            # a BATCH of such frames (BASELINE configs[3] is a batch of scenes; so is a dynamic sequence): one call per frame
            # against raht_fwd_quant_batch + raht_dequant_inv_batch (stage k of all frames in one launch)
            from raht_3dgs_codec_amd import ops as _ops
            nb, Db = 8, 56
            plans_b, Cs_b = [], []
            for i in range(nb):
                Vb, kb, Cb = synth.scene(1_000_000, 10, Db, 20 + i)
                plans_b.append(R.RahtPlan.from_keys(torch.from_numpy(kb.view(np.int64)).to(dev), 30))
                Cs_b.append(torch.from_numpy(Cb).to(dev))
            rows_b = sum(p.N for p in plans_b)

            def looped():
                for p, c in zip(plans_b, Cs_b):
                    p.dequant_inverse(p.forward_quant(c, a.quant_step), a.quant_step)

            def batched():
                _ops.dequant_inverse_batch(plans_b, _ops.forward_quant_batch(plans_b, Cs_b, a.quant_step), a.quant_step)
            Qb = _ops.forward_quant_batch(plans_b, Cs_b, a.quant_step)
            assert all(torch.equal(q, p.forward_quant(c, a.quant_step)) for q, p, c in zip(Qb, plans_b, Cs_b)), "batch != single-scene calls"
            del Qb
            for _ in range(10):
                looped(); batched()
            t_l, t_b = timed(looped, 50), timed(batched, 50)
            alg_b = 2 * (8.0 * rows_b * Db + 8.0 * rows_b)
            out["batch"] = {"workload": f"{nb} frames of the reference's shape ({rows_b} Gaussians in all, J=10, {Db} channels), same step per frame",
                            "one_call_per_frame_ms": round(t_l, 4), "batched_ms": round(t_b, 4),
                            "batched_value": round(rows_b / (t_b * 1e-3) / 1e6, 1), "one_call_per_frame_value": round(rows_b / (t_l * 1e-3) / 1e6, 1), "unit": "M-Gaussians/s",
                            "batched_frac_of_peak": round(alg_b / (t_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "one_call_per_frame_frac_of_peak": round(alg_b / (t_l * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "bit_identical_to_single_scene_calls": True}
            del plans_b, Cs_b
            # ---- the whole codec step on a frame of the headline size (tools/e2e_frame.py runs all nine steps): transform +
            # quantize + entropy stage + decode back, with the byte-exact RLGR coder on the host threads (the reference's format:
            # one stream per channel) and with the segmented RLGR coder on the GPU (every segment byte-identical to the
            # reference coder's stream for its slice; own container)
            from raht_3dgs_codec_amd import pipeline as _pl
            Vf, kf, Cf = synth.scene(3_000_000, 12, 56, 2)
            Vt, Ct = torch.from_numpy(Vf), torch.from_numpy(Cf)
            fsteps = [0.04, 0.2]
            _pl.encode_frame(Vt, Ct, 12, fsteps[:1], frame=0, entropy="gpu")
            rg = _pl.encode_frame(Vt, Ct, 12, fsteps, entropy="gpu")
            rh = _pl.encode_frame(Vt, Ct, 12, fsteps, entropy="host")
            assert all(torch.equal(x["C_rec"], y["C_rec"]) for x, y in zip(rg, rh)), "GPU entropy stage reconstructs differently"

            # ... and with ALL nine steps of the frame through every stage at once (one forward pass with nine quantizers, one set
            # of coder launches -- nine times the independent streams --, one decoder launch, the wire copies beside the decode side)
            nine = [0.01, 0.04, 0.08, 0.12, 0.16, 0.20, 0.24, 0.32, 0.64]
            _pl.encode_frame(Vt, Ct, 12, nine, frame=0, entropy="gpu", batch_steps=True, keep_rec=False)
            rb = _pl.encode_frame(Vt, Ct, 12, nine, entropy="gpu", batch_steps=True, keep_rec=False)
            for x in rg:
                y = [r for r in rb if r["Quantization_Step"] == x["Quantization_Step"]][0]
                assert y["size_bytes"] == x["size_bytes"] and y["PSNR_all"] == x["PSNR_all"], "the batched steps code / reconstruct differently"

            def ms(rows_, k):
                return round(float(np.mean([r[k] for r in rows_])) * 1e3, 3)
            out["frame_codec"] = {"workload": f"one voxelized frame, {Vf.shape[0]} Gaussians x 56 channels (J=12), steps {fsteps}: per step forward RAHT + quantize + "
                                              "RLGR encode + RLGR decode + round-trip check + dequantize + inverse RAHT + 5 PSNR columns",
                                  "gpu_entropy": {"step_ms": ms(rg, "Step_wall_time"), "rlgr_encode_ms": ms(rg, "Entropy_enc_time"), "rlgr_decode_ms": ms(rg, "Entropy_dec_time"),
                                                  "container_to_host_ms": ms(rg, "D2H_time"), "bytes": [r["size_bytes"] for r in rg], "segment_symbols": 2048},
                                  "gpu_entropy_all_steps_at_once": {"steps": nine, "step_ms": ms(rb, "Step_wall_time"), "rlgr_encode_ms": ms(rb, "Entropy_enc_time"),
                                                                    "rlgr_decode_ms": ms(rb, "Entropy_dec_time"), "forward_ms": ms(rb, "RAHT_transform_time"),
                                                                    "inverse_and_psnr_ms": ms(rb, "iRAHT_time"),
                                                                    "container_to_host_ms": ms(rb, "D2H_time"),
                                                                    "bytes": [r["size_bytes"] for r in rb], "segment_symbols": 2048,
                                                                    "same_bytes_and_psnr_as_step_by_step": True},
                                  "host_entropy": {"step_ms": ms(rh, "Step_wall_time"), "rlgr_encode_ms": ms(rh, "Entropy_enc_time"), "rlgr_decode_ms": ms(rh, "Entropy_dec_time"),
                                                   "integers_to_host_ms": ms(rh, "D2H_time"), "integers_to_device_ms": ms(rh, "H2D_time"), "bytes": [r["size_bytes"] for r in rh],
                                                   "threads": "the container's CPU quota"},
                                  "same_reconstruction": True}
            del rg, rh, rb, Vt, Ct
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        if not solo:
            sh.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
