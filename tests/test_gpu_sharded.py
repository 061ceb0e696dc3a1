"""GPU tests of the multi-GPU path's building blocks on ONE MI355X:

* row-mapped plans (the replicated top tree working in place on the padded all-gather buffer);
* rows gather / scatter, voxel keys;
* four ranks sharing the one GPU over gloo (host-staged collectives): un-partitioned cloud -> all-to-all by
  Morton prefix -> local HIP voxelizer -> ShardedRaht (HIP local ops) -> against the oracle on the whole cloud.
  RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU bench runs; everything around
  the two collectives is what runs here.
"""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import _lib
    _lib.lib()
    return R


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("engine", ["tile", "level"])
@pytest.mark.parametrize("dtype,D", [("f32", 59), ("f32", 14), ("f64", 11), ("f32", 3), ("f64", 1)])
def test_row_mapped_plan_equals_compact_plan(rt, engine, dtype, D):
    """A weighted <= 512-row tree whose rows sit at scattered positions of a larger matrix (the padded gather
    buffer): forward / inverse through the map == the same plan on the compacted rows, bit for bit; rows outside
    the map are not written."""
    import torch
    rng = np.random.default_rng(17)
    tk = np.unique(rng.integers(0, 512, size=400)).astype(np.int64)
    n = tk.shape[0]
    tw = rng.integers(1, 200000, size=n).astype(np.int64)
    td = torch.float32 if dtype == "f32" else torch.float64
    X = torch.from_numpy(rng.standard_normal((n, D))).to(td).cuda()
    slot, world = 97, 5
    # entries land in `world` padded slots of `slot` rows, rank r holding a run of consecutive entries
    cuts = [0, 60, 60, 157, n - 90, n]
    rows = np.concatenate([r * slot + np.arange(cuts[r + 1] - cuts[r]) for r in range(world)]).astype(np.int64)
    assert rows.shape[0] == n and max(cuts[r + 1] - cuts[r] for r in range(world)) <= slot
    plain = rt.RahtPlan.from_keys(_dev(tk), 9, leaf_weights=_dev(tw))
    plain.set_engine(engine)
    mapped = rt.RahtPlan.from_keys(_dev(tk), 9, leaf_weights=_dev(tw))
    mapped.set_engine(engine)
    mapped.set_row_map(_dev(rows), world * slot)
    T0 = plain.forward(X, want_w=False)
    big = torch.full((world * slot, D), 7.25, dtype=td, device="cuda")
    big[_dev(rows)] = X
    out = torch.full((world * slot, D), -3.5, dtype=td, device="cuda")
    res = mapped.forward(big, want_w=False, out=out)
    assert res.data_ptr() == out.data_ptr()
    assert torch.equal(out[_dev(rows)], T0)
    keep = torch.ones(world * slot, dtype=torch.bool, device="cuda")
    keep[_dev(rows)] = False
    if engine == "tile" and D * X.element_size() >= 16:
        assert bool((out[keep] == -3.5).all())                 # rows outside the map untouched (top kernel)
    back = mapped.inverse(out)
    assert torch.equal(back[_dev(rows)], plain.inverse(T0))
    # errors: node weights and fused quantization are not offered through a map; a bad map is refused
    with pytest.raises(rt.RahtError):
        mapped.forward(big, want_w=True)
    if dtype == "f32":
        with pytest.raises(rt.RahtError):
            mapped.forward_quant(big, 0.5)
    bad = rows.copy(); bad[3] = world * slot
    with pytest.raises(rt.RahtError):
        mapped.set_row_map(_dev(bad), world * slot)
    mapped.set_row_map(None, 0)
    assert torch.equal(mapped.forward(X, want_w=False), T0)


def test_rows_gather_scatter_and_voxel_keys(rt, oracle):
    import torch
    from raht_3dgs_codec_amd import ops
    rng = np.random.default_rng(23)
    for dt in (torch.float32, torch.float64, torch.int32):
        M = torch.from_numpy(rng.integers(-1000, 1000, size=(5000, 59))).to(dt).cuda()
        pos = torch.from_numpy(rng.permutation(5000)[:777].astype(np.int64)).cuda()
        out = torch.zeros((777, 59), dtype=dt, device="cuda")
        ops.rows_gather(M, pos, out)
        assert torch.equal(out, M[pos])
        dst = torch.zeros((5000, 64), dtype=dt, device="cuda")[:, :59]          # strided destination rows
        ops.rows_scatter(out, pos, dst)
        ref = torch.zeros_like(dst); ref[pos] = out
        assert torch.equal(dst, ref)
    # voxel keys == the oracle voxelizer's keys before sorting
    P = (rng.random((20000, 3)) * 9.0 - 2.0).astype(np.float32)
    PC = np.concatenate([P, rng.standard_normal((20000, 4)).astype(np.float32)], axis=1)
    r = oracle.voxelize(PC, 9)
    keys = ops.voxel_keys(_dev(PC), r["vmin"].tolist(), r["width"], 9).cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(keys[r["sort_idx"]], r["keys_sorted"])


# ------------------------------------------------------------------ four ranks, one GPU, gloo collectives
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, J, n, d, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from raht_3dgs_codec_amd import sharded, synth
        from oracle import oracle as orc

        rng = np.random.default_rng(321)
        P = synth.blob_positions(n, seed=19, nblobs=40, sigma=0.04).astype(np.float32) * np.float32(3.0) + np.float32(0.5)
        P[::5] = P[2::5][: P[::5].shape[0]]                           # several points per voxel
        A = rng.standard_normal((n, d)).astype(np.float32)
        PC = np.concatenate([P, A], axis=1)
        bounds = [0] + sorted(rng.choice(np.arange(1, n), size=world - 1, replace=False).tolist()) + [n]
        mine = torch.from_numpy(PC[bounds[rank]:bounds[rank + 1]].copy()).cuda()

        PCvox, keys, info = sharded.exchange_by_prefix(mine, J, prefix_bits=9)          # HIP local ops
        ref = orc.voxelize(PC, J)
        cnt = torch.tensor([PCvox.shape[0]], dtype=torch.int64)
        allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, cnt)
        lo = int(sum(int(c.item()) for c in allc[:rank]))
        nloc = int(PCvox.shape[0])
        assert int(sum(int(c.item()) for c in allc)) == ref["Nvox"]
        ref_keys = ref["keys_sorted"][ref["voxel_indices"]]
        np.testing.assert_array_equal(keys.cpu().numpy().view(np.uint64), ref_keys[lo:lo + nloc])
        np.testing.assert_array_equal(PCvox.cpu().numpy(), ref["PCvox"][lo:lo + nloc])   # float32 means, bit-exact

        sh = sharded.ShardedRaht(keys, 3 * J, prefix_bits=9)
        assert sh.world == world and sh.total_rows == ref["Nvox"]
        C_loc = PCvox[:, 3:].contiguous()
        V = synth.keys_to_coords(ref_keys, J)
        po = orc.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
        Cw = ref["PCvox"][:, 3:].astype(np.float64)
        To, _ = orc.raht_fwd(Cw, po)
        colmax = np.abs(To).max(axis=0)
        T = sh.forward(C_loc)
        err = np.abs(T.cpu().numpy().astype(np.float64) - To[lo:lo + nloc]).max(axis=0)
        assert np.all(err <= 2e-6 * colmax), float((err / colmax).max())
        R = sh.inverse(T)
        assert (R - C_loc).abs().max().item() <= 1e-5 * float(np.abs(Cw).max())
        step = 0.02
        Q = sh.forward_quant(C_loc, step)
        Tq = np.empty((nloc, d), dtype=np.float64)
        Tq[sh.plan.order_RAGFT.cpu().numpy()] = Q.cpu().numpy().astype(np.float64) * step
        assert np.all(np.abs(Tq - To[lo:lo + nloc]) <= 0.5 * step * 1.0001 + 2e-6 * colmax)
        Rq = sh.dequant_inverse(Q, step)
        e = torch.tensor([float(((Rq.double() - C_loc.double()) ** 2).sum()), float(((torch.from_numpy(Tq).cuda() - T.double()) ** 2).sum())], dtype=torch.float64)
        dist.all_reduce(e)
        assert abs(e[0].item() - e[1].item()) <= 1e-3 * max(e[1].item(), 1e-30)         # orthonormal: error energies agree
        chk = sh.check_against_unsharded(C_loc, step)
        assert chk["ok"], chk
        # float64 through the same driver
        T64 = sh.forward(C_loc.double())
        np.testing.assert_allclose(T64.cpu().numpy(), To[lo:lo + nloc], rtol=1e-11, atol=1e-11 * float(colmax.max()))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [4, 5])
def test_four_ranks_share_the_gpu_exchange_and_transform(rt, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, 10, 300000 if world == 4 else 120000, 14, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def _rccl_main(port, q):
    try:
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import raht_3dgs_codec_amd as R
        from raht_3dgs_codec_amd import sharded, synth
        V, keys, C = synth.scene(150000, 11, 59, seed=21)
        kd = torch.from_numpy(keys.view(np.int64)).cuda()
        Cd = torch.from_numpy(C).cuda()
        sh = sharded.ShardedRaht(kd, 33, prefix_bits=9, force_collectives=True)      # the two all-gathers go through RCCL
        p = R.RahtPlan.from_keys(kd, 33)
        T0 = p.forward(Cd, want_w=False)
        T1 = sh.forward(Cd)
        scale = T0.abs().amax(dim=0)
        assert bool(((T1 - T0).abs().amax(dim=0) <= 2e-6 * scale).all())
        for _ in range(3):
            R1 = sh.step(Cd, 0.01)
        assert (R1 - p.dequant_inverse(p.forward_quant(Cd, 0.01), 0.01)).abs().max().item() <= 0.02
        chk = sh.check_against_unsharded(Cd, 0.01)
        assert chk["ok"], chk
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t); dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put("ok")
    except Exception:
        q.put(traceback.format_exc())


def test_rccl_one_rank_group_carries_the_gathers(rt):
    """RCCL on the one GPU of this box: a one-rank NCCL process group, the sharded step with its all-gathers forced
    through the collective (what N ranks do over xGMI, minus the wire)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_main, args=(_free_port(), q))
    p.start()
    msg = q.get(timeout=600)
    p.join(timeout=120)
    assert msg == "ok", msg


# ------------------------------------------------------------------ bench.py starts its own ranks
def _run_bench(args, timeout=900):
    """`python bench.py ...` as a PLAIN subprocess (no launcher around it), the way the driver starts the 1-GPU bench."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, f"bench.py {' '.join(args)} -> rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_its_own_ranks(rt):
    """`python bench.py --gpus 2 ...` with no launcher: bench.py starts the two ranks itself (before it touches the GPU),
    relays rank 0's line and the children's status. gloo, both ranks on this box's one GPU: a strong-scaling cut of a small
    cfg5 scene; the line says how many ranks the collective backend saw and separates the exchange from the local work."""
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "cfg5", "--rows", "400000", "--steps", "3", "--warmup", "1",
                      "--settle-steps", "0"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["oracle_gate"]["ok"] and out["oracle_gate"]["rows_total"] > out["config"]["rows_per_gpu"]
    mg = out["multi_gpu"]
    assert mg["ranks"] == 2 and mg["collective_backend"] == "gloo" and mg["rccl_world"] is None
    assert mg["collective_ms"]["forward_all_gather"] > 0 and mg["collective_ms"]["inverse_all_gather"] > 0
    assert 0 < mg["local_step_ms"] and mg["gathered_bytes_per_step"] == out["config"]["gathered_bytes_per_step"] > 0
    assert 0 < mg["roofline_stage0"]["fwd_frac"] < 1 and 0 < mg["roofline_stage0"]["inv_frac"] < 1


def test_bench_one_rank_rccl_group(rt):
    """The same path through RCCL, as far as one GPU goes: --sharded runs ShardedRaht in a one-rank NCCL group."""
    out = _run_bench(["--gpus", "1", "--backend", "nccl", "--sharded", "--workload", "cfg5", "--rows", "400000", "--steps", "3", "--warmup", "1",
                      "--settle-steps", "0"])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["oracle_gate"]["ok"]
    mg = out["multi_gpu"]
    assert mg["rccl_world"] == 1 and mg["collective_backend"] == "nccl"
    assert mg["collective_ms"]["forward_all_gather"] > 0


def _empty_rank_main(rank, world, port, q):
    """Three gloo ranks on the one GPU; the scene occupies prefixes [0, 300) only, so the last rank owns no row, and rank 1
    starts the exchange with no point: HIP local ops with N = 0, every collective still entered by every rank."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import raht_3dgs_codec_amd as R
        from raht_3dgs_codec_amd import sharded, synth
        J, D = 8, 14
        V, keys, C = synth.scene(40000, J, D, seed=5, prefix_range=(0, 300, 9))
        pref = (keys >> np.uint64(3 * J - 9)).astype(np.int64)
        per = 512 // world
        lo, hi = rank * per, ((rank + 1) * per if rank < world - 1 else 512)
        mine = np.nonzero((pref >= lo) & (pref < hi))[0]
        assert (mine.size == 0) == (rank == world - 1)
        kd = torch.from_numpy(keys[mine].view(np.int64).copy()).cuda()
        Cd = torch.from_numpy(C[mine]).cuda()
        sh = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9)
        full = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64).copy()).cuda(), 3 * J)
        Tf = full.forward(torch.from_numpy(C).cuda(), want_w=False)
        T = sh.forward(Cd)
        if mine.size:
            scale = Tf.abs().amax(dim=0)
            assert bool(((T - Tf[mine[0]: mine[-1] + 1]).abs().amax(dim=0) <= 4e-6 * scale).all())
        else:
            assert sh.plan is None and tuple(T.shape) == (0, D)
        Rr = sh.step(Cd, 0.01)
        assert tuple(Rr.shape) == tuple(Cd.shape)
        chk = sh.check_against_unsharded(Cd, 0.01, keys_sorted=kd)
        assert chk["ok"], chk
        # front end: rank 1 holds no point before the exchange
        n = 30000
        rng = np.random.default_rng(4)
        P = (synth.blob_positions(n, seed=3, nblobs=10, sigma=0.05) * 5.0).astype(np.float32)
        PC = np.concatenate([P, rng.standard_normal((n, 3)).astype(np.float32)], axis=1)
        bounds = [0, n // 2, n // 2, n]
        part = torch.from_numpy(PC[bounds[rank]:bounds[rank + 1]].copy()).cuda()
        PCvox, vkeys, info = sharded.exchange_by_prefix(part, 7, prefix_bits=9)
        cnt = torch.tensor([PCvox.shape[0]], dtype=torch.int64)
        dist.all_reduce(cnt)
        from oracle import oracle as orc
        assert int(cnt.item()) == orc.voxelize(PC, 7)["Nvox"] and info["N_global"] == n
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_rank_without_rows_on_the_gpu(rt):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_empty_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


# ------------------------------------------------------------------ direct exchange (raht_xchg_*)
def _direct_main(rank, world, port, q):
    """ranks share the one GPU; gloo carries the hipIpc handles once, the gathers themselves are direct writes + flags"""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from raht_3dgs_codec_amd import sharded, synth
        J, D = 10, 59
        per = 512 // world
        V, keys, C = synth.scene(60000 + 7000 * rank, J, D, seed=31 + rank, prefix_range=(rank * per, (rank + 1) * per, 9))
        kd = torch.from_numpy(keys.view(np.int64).copy()).cuda()
        Cd = torch.from_numpy(C).cuda()
        ref = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9)                    # gloo all-gathers (host staged)
        drc = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9, direct=True, force_collectives=True)
        T0, Q0 = ref.forward(Cd), ref.forward_quant(Cd, 0.01)
        for it in range(5):                                                    # both buffer parities, several times
            assert torch.equal(drc.forward(Cd), T0), it
            assert torch.equal(drc.forward_quant(Cd, 0.01), Q0), it
        assert torch.equal(drc.inverse(T0), ref.inverse(T0))
        assert torch.equal(drc.dequant_inverse(Q0, 0.01), ref.dequant_inverse(Q0, 0.01))
        T64 = drc.forward(Cd.double())                                         # another (D, dtype): another exchange block
        assert torch.equal(T64, ref.forward(Cd.double()))
        chk = drc.check_against_unsharded(Cd, 0.01, keys_sorted=kd)
        assert chk["ok"], chk
        assert drc.exchange_status() == 0
        drc.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_direct_exchange_equals_the_collective(rt, world):
    """(world 5: the most ranks this box lets share its GPU -- at most 6 processes may use the card at once and this test process is
    one of them; an 8-rank group has only ever run on the CPU, tests/test_sharded_gloo.py.)
    ShardedRaht(direct=True): every rank writes its root slot into each peer's gather buffer (hipIpc-mapped fine-grained
    device memory) and raises a flag -- one launch per direction -- instead of all_gather_into_tensor. Bit-identical results;
    here the 'peers' are processes sharing this box's one GPU (IPC to the same device), on xGMI they are the other GPUs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_direct_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_bench_direct_exchange(rt):
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--direct", "--workload", "cfg5", "--rows", "400000", "--steps", "3", "--warmup", "1",
                      "--settle-steps", "0"])
    assert out["oracle_gate"]["ok"] and out["multi_gpu"]["exchange"].startswith("direct") and out["multi_gpu"]["exchange_status"] == 0
    assert out["multi_gpu"]["collective_ms"]["forward_all_gather"] > 0
