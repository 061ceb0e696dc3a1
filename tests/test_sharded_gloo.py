"""World-size-2, 3 and 8 gloo tests of the Morton-prefix sharded RAHT on CPU (8 = the node the north star names:
no xGMI / RCCL run with more than one rank exists anywhere, so the 8-rank control flow is at least walked here).

The host-side logic under test is raht_3dgs_codec_amd.sharded.ShardedRaht: prefix-range shards,
root directory exchange, ONE all-gather per direction, replicated weighted top tree, write-back of
the rank's own top coefficients. The shard-local arithmetic is injected (tests/numpy_ops.py) because
the product's local ops are GPU-only; the result is checked against the C oracle run on the WHOLE
scene, i.e. against the reference's unsharded transform."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, J, n, D, q, balanced=False, prefix_range=None):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from raht_3dgs_codec_amd import sharded, synth
        from tests.numpy_ops import NumpyLocalOps
        from oracle import oracle as orc

        # the whole scene, same on every rank (prefix_range: only part of the cube is occupied -> some ranks own no row)
        V, keys, C = synth.scene(n, J, D, seed=77, prefix_range=prefix_range)
        nbits, pb = 3 * J, 9
        if balanced:
            # shards cut by the population-balanced helper (512-bin prefix histogram)
            cuts = sharded.balanced_prefix_cuts(torch.from_numpy(keys.view(np.int64).copy()), nbits, world, pb)
            mine = np.arange(cuts[rank], cuts[rank + 1])
        else:
            pref = (keys >> np.uint64(nbits - pb)).astype(np.int64)
            per = (1 << pb) // world
            lo, hi = rank * per, ((rank + 1) * per if rank < world - 1 else 1 << pb)
            mine = np.nonzero((pref >= lo) & (pref < hi))[0]
        assert (mine.size > 0 or prefix_range is not None) and np.all(np.diff(mine) == 1)
        k_loc = torch.from_numpy(keys[mine].view(np.int64).copy())
        C_loc = torch.from_numpy(C[mine].astype(np.float64))

        sh = sharded.ShardedRaht(k_loc, nbits, prefix_bits=pb, local_ops=NumpyLocalOps)
        assert sh.world == world and sh.total_rows == keys.shape[0]

        # reference: the oracle on the WHOLE scene
        po = orc.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
        To, _ = orc.raht_fwd(C.astype(np.float64), po)

        T = sh.forward(C_loc)
        np.testing.assert_allclose(T.numpy(), To[mine], rtol=1e-11, atol=1e-11 * np.abs(To).max())
        R = sh.inverse(T)
        np.testing.assert_allclose(R.numpy(), C[mine].astype(np.float64), rtol=1e-11, atol=1e-11)

        # quantized path: dequantized coefficients of this shard == quantized oracle coefficients
        step = 0.05
        Q = sh.forward_quant(C_loc, step)
        Tq = torch.empty_like(T)
        if mine.size:
            Tq[sh.plan.order_RAGFT] = Q.to(torch.float64) * step
            ref = np.floor(To[mine] / step + 0.5) * step
            bad = np.abs(Tq.numpy() - ref) > 1e-9
            assert bad.mean() < 1e-4            # only rounding ties may differ
        else:
            assert sh.plan is None and tuple(Q.shape) == (0, D) and tuple(T.shape) == (0, D)
        Rq = sh.dequant_inverse(Q, step)
        assert mine.size == 0 or float((Rq - C_loc).abs().max()) < 40 * step
        chk = sh.check_against_unsharded(C_loc, step, keys_sorted=k_loc)
        assert chk["ok"], chk
        # orthonormal transform: global error energy == global quantization error energy
        e_loc = torch.tensor([float(((Rq - C_loc) ** 2).sum()), float(((Tq - T) ** 2).sum())], dtype=torch.float64)
        dist.all_reduce(e_loc)
        assert abs(e_loc[0].item() - e_loc[1].item()) <= 1e-6 * max(e_loc[1].item(), 1e-30)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,J,n,D,balanced", [(2, 6, 6000, 5, False), (2, 10, 4000, 14, False), (3, 5, 3000, 3, False),
                                                  (3, 8, 5000, 7, True), (8, 7, 6000, 5, True), (8, 6, 4000, 3, False)])
def test_sharded_matches_unsharded_oracle(world, J, n, D, balanced):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, J, n, D, q, balanced)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("world,J,n,D,prefix_range", [(3, 6, 3000, 5, (0, 300, 9)), (2, 5, 1500, 3, (256, 512, 9)), (3, 6, 2000, 4, (200, 320, 9)),
                                                      (8, 6, 3000, 4, (0, 200, 9)), (8, 6, 2500, 3, (70, 330, 9))])
def test_rank_without_rows_still_joins_the_collectives(world, J, n, D, prefix_range):
    """A prefix range that holds no point (uneven scenes; fewer occupied prefixes than ranks): that rank has no plan and
    no roots but must enter every collective -- round 2 raised on it and left the other ranks blocked."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, J, n, D, q, False, prefix_range)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def _worker_frontend(rank, world, port, J, n, d, q, empty_rank=-1):
    """Un-partitioned input: every rank holds an arbitrary third / half of an unsorted cloud with duplicates.
    exchange_by_prefix + ShardedRaht must reproduce the single-process pipeline on the WHOLE cloud:
    oracle voxelizer (reference voxelize_pc.py:62-172) -> oracle RAHT."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from raht_3dgs_codec_amd import sharded, synth
        from tests.numpy_ops import NumpyLocalOps
        from oracle import oracle as orc

        rng = np.random.default_rng(123)
        P = synth.blob_positions(n, seed=9, nblobs=12, sigma=0.05).astype(np.float32) * np.float32(7.5) - np.float32(1.25)
        P[::7] = P[1::7][: P[::7].shape[0]]                           # duplicates: several points per voxel
        A = rng.standard_normal((n, d)).astype(np.float32)
        PC = np.concatenate([P, A], axis=1)
        # arbitrary, uneven parts in rank order (their concatenation is the whole cloud)
        bounds = [0] + sorted(rng.choice(np.arange(1, n), size=world - 1, replace=False).tolist()) + [n]
        if empty_rank >= 0:                                           # one rank starts with no point at all
            bounds[empty_rank + 1] = bounds[empty_rank]
            bounds = sorted(bounds)
        mine = torch.from_numpy(PC[bounds[rank]:bounds[rank + 1]].copy())

        PCvox, keys, info = sharded.exchange_by_prefix(mine, J, prefix_bits=9, local_ops=NumpyLocalOps)
        ref = orc.voxelize(PC, J)                                     # whole cloud, one process
        assert info["N_global"] == n
        np.testing.assert_array_equal(info["vmin"].numpy(), ref["vmin"])
        assert info["width"] == ref["width"]
        # this rank's shard = a contiguous run of the reference's voxels, bit for bit
        cnt = torch.tensor([PCvox.shape[0]], dtype=torch.int64)
        allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, cnt)
        lo = int(sum(int(c.item()) for c in allc[:rank]))
        assert int(sum(int(c.item()) for c in allc)) == ref["Nvox"]
        ref_keys = ref["keys_sorted"][ref["voxel_indices"]]
        np.testing.assert_array_equal(keys.numpy().view(np.uint64), ref_keys[lo:lo + PCvox.shape[0]])
        np.testing.assert_array_equal(PCvox.numpy(), ref["PCvox"][lo:lo + PCvox.shape[0]])      # means bit-exact
        sizes = [int(c.item()) for c in allc]
        assert max(sizes) - min(sizes) <= max(np.bincount((ref_keys >> np.uint64(3 * J - 9)).astype(np.int64))) + 1

        # ... and the sharded transform of the exchanged shards == the oracle on the whole voxelized cloud
        sh = sharded.ShardedRaht(keys, 3 * J, prefix_bits=9, local_ops=NumpyLocalOps)
        C_loc = PCvox[:, 3:].to(torch.float64)
        V = synth.keys_to_coords(ref_keys, J)
        po = orc.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
        To, _ = orc.raht_fwd(ref["PCvox"][:, 3:].astype(np.float64), po)
        T = sh.forward(C_loc)
        np.testing.assert_allclose(T.numpy(), To[lo:lo + PCvox.shape[0]], rtol=1e-11, atol=1e-11 * np.abs(To).max())
        chk = sh.check_against_unsharded(C_loc, 0.05)
        assert chk["ok"], chk
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,J,n,d,empty_rank", [(2, 6, 5000, 3, -1), (3, 8, 7000, 5, -1), (3, 6, 4000, 3, 1), (2, 6, 3000, 2, 0),
                                                    (8, 7, 9000, 3, 2)])
def test_unpartitioned_cloud_exchange_then_sharded_transform(world, J, n, d, empty_rank):
    """all-to-all bucket exchange by 9-bit Morton prefix + local radix sort / voxelizer, against the
    oracle's sort of the whole cloud (reference python/voxelize_pc.py:97-118)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_frontend, args=(r, world, port, J, n, d, q, empty_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_single_process_degenerates_to_plain_transform():
    """world = 1 (no process group): the sharded driver is just local + top stages."""
    sys.path.insert(0, ROOT)
    from raht_3dgs_codec_amd import sharded, synth
    from tests.numpy_ops import NumpyLocalOps, NumpyPlan
    V, keys, C = synth.scene(3000, 7, 4, seed=5)
    k = torch.from_numpy(keys.view(np.int64).copy())
    Cd = torch.from_numpy(C.astype(np.float64))
    sh = sharded.ShardedRaht(k, 21, prefix_bits=9, local_ops=NumpyLocalOps)
    full = NumpyPlan(k, 21)
    np.testing.assert_allclose(sh.forward(Cd).numpy(), full.forward(Cd).numpy(), rtol=1e-12, atol=1e-12)
    assert sh.roundtrip_error(Cd) < 1e-12


def test_balanced_prefix_cuts_fall_on_prefix_boundaries_and_balance():
    import numpy as np
    import torch
    from raht_3dgs_codec_amd import sharded, synth
    for seed, world in ((1, 2), (2, 3), (3, 8)):
        keys = torch.from_numpy(synth.sorted_unique_keys(60000, 10, seed).view(np.int64))
        N, nbits = keys.shape[0], 30
        cuts = sharded.balanced_prefix_cuts(keys, nbits, world)
        assert cuts[0] == 0 and cuts[-1] == N and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        pref = keys >> (nbits - 9)
        for b in cuts[1:-1]:
            assert 0 < b < N and pref[b - 1] != pref[b]          # never inside a prefix node
        biggest_bin = int(torch.bincount(pref, minlength=512).max())
        sizes = [b - a for a, b in zip(cuts, cuts[1:])]
        assert max(abs(sz - N / world) for sz in sizes) <= biggest_bin
    # degenerate: everything in one prefix node -> one shard gets it all, the others are empty
    one = torch.arange(100, dtype=torch.int64)
    c = sharded.balanced_prefix_cuts(one, 30, 4)
    assert c[0] == 0 and c[-1] == 100 and sorted(c) == c
