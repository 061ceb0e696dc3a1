#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python operators (CPU, float64).

Runs only in the build container, where /root/reference is mounted (the reference never travels
to the GPU box).  Output: small ``.npz`` fixtures next to this file, committed to the repo.  Each
fixture holds inputs and the reference's outputs (data only -- no reference source text).

    python tests/golden/gen_golden.py            # regenerate everything

Reference entry points exercised (the ``raht_fn`` table of python/encode_3dgs.py:23-27 plus the
voxelizer):  RAHT_param_reorder_fast (RAHT_param.py:190), RAHT2_optimized (RAHT.py:252),
inverse_RAHT_optimized (iRAHT.py:40), get_morton_code / voxelize_pc_batched (voxelize_pc.py:25,62),
and the driver-inline quantize/reorder arithmetic (encode_3dgs.py:204-217, 261-268).
"""
import os
import sys

import numpy as np

REF = os.environ.get("RAHT_REFERENCE", "/root/reference/python")
sys.path.insert(0, REF)
import torch  # noqa: E402

from RAHT import RAHT2_optimized  # noqa: E402
from RAHT_param import RAHT_param, RAHT_param_reorder_fast  # noqa: E402
from iRAHT import inverse_RAHT_optimized  # noqa: E402
from voxelize_pc import get_morton_code, voxelize_pc_batched  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def morton_np(V, J):
    return get_morton_code(torch.from_numpy(V.astype(np.int64)), J).numpy().astype(np.uint64)


def sorted_unique_voxels(V, J):
    """Integer coords -> unique, Morton-sorted (what the encode drivers expect as input)."""
    V = np.unique(V.astype(np.int64), axis=0)
    mc = morton_np(V, J)
    o = np.argsort(mc, kind="stable")
    return V[o]


def blob_cloud(rng, n, J, nblobs=8, sigma=0.05):
    ctr = rng.uniform(0.1, 0.9, size=(nblobs, 3))
    p = ctr[rng.integers(0, nblobs, size=n)] + rng.normal(0, sigma, size=(n, 3))
    p = np.clip(p, 0.0, 1.0 - 1e-9)
    return np.floor(p * (1 << J)).astype(np.int64)


def gaussian_attrs(rng, n, D):
    """3DGS-like attribute matrix (quats, scales, opacity, SH...), float32-representable."""
    cols = []
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    cols.append(q)
    cols.append(np.exp(rng.normal(-4, 1, size=(n, 3))))
    cols.append(1 / (1 + np.exp(-rng.normal(0, 2, size=(n, 1)))))
    cols.append(rng.normal(0, 0.5, size=(n, 3)))
    cols.append(rng.normal(0, 0.1, size=(n, 64)))
    A = np.concatenate(cols, axis=1)[:, :D]
    return A.astype(np.float32)


def transform_case(name, V, J, C32, steps=(), also_slow_param=False):
    """V: sorted unique int64 (N,3); C32: float32 (N,D)."""
    N = V.shape[0]
    Vt = torch.from_numpy(V.astype(np.float64))
    origin = torch.zeros(3, dtype=torch.float64)
    List, Flags, weights, order = RAHT_param_reorder_fast(Vt, origin, 2 ** J, J)
    if also_slow_param and N > 1:
        L2, F2, W2 = RAHT_param(Vt, origin, 2 ** J, J)
        assert len(L2) == len(List)
        for a, b in zip(L2, List):
            assert torch.equal(a.long(), b.long())
        for a, b in zip(F2, Flags):
            assert torch.equal(a.bool(), b.bool())
    Ct = torch.from_numpy(C32.astype(np.float64))
    T, w = RAHT2_optimized(Ct, List, Flags, weights)
    Crec = inverse_RAHT_optimized(T, List, Flags, weights)
    out = dict(
        V=V.astype(np.int32), J=np.int32(J), C=C32,
        morton=morton_np(V, J),
        level_len=np.array([len(l) for l in List], dtype=np.int64),
        list_cat=torch.cat(List).numpy().astype(np.int32),
        flags_cat=np.packbits(torch.cat(Flags).numpy().astype(np.uint8)),
        weights_cat=torch.cat(weights).numpy().astype(np.int32),
        order=(np.array([-1], dtype=np.int64) if order is None else order.numpy().astype(np.int64)),
        order_is_none=np.bool_(order is None),
        T=T.numpy(), w=w.numpy().reshape(-1),
        roundtrip_maxerr=np.float64((Crec - Ct).abs().max().item()),
    )
    for s in steps:                                   # encode_3dgs.py:204-217, 261-275
        enc = torch.floor(T / s + 0.5)
        reord = enc.index_select(0, order)
        q = reord.to(torch.int32)
        dec = q.to(torch.float64) * s
        dec = dec[torch.argsort(order), :]
        rec = inverse_RAHT_optimized(dec, List, Flags, weights)
        out[f"q_step{s}"] = q.numpy()
        out[f"crec_step{s}"] = rec.numpy().astype(np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: N={N} J={J} D={C32.shape[1]} levels={len(Flags)} "
          f"order={'None' if order is None else tuple(order.shape)} rt={out['roundtrip_maxerr']:.2e}")


def voxelize_case(name, PC32, J, vmin=None, width=None):
    PC = torch.from_numpy(PC32)
    vm = None if vmin is None else torch.tensor(vmin, dtype=torch.float32)
    PCvox, PCsorted, vox_idx, DeltaPC, info = voxelize_pc_batched(PC, vm, width, J, device="cpu")
    V = PC[:, :3]
    V0 = V - info["vmin"].unsqueeze(0)
    Vint = torch.clamp(torch.floor(V0 / info["voxel_size"]).long(), 0, 2 ** J - 1)
    M = get_morton_code(Vint, J)
    Ms, _ = torch.sort(M, stable=True)
    out = dict(
        PC=PC32, J=np.int32(J),
        vmin_in=(np.zeros(0, np.float32) if vmin is None else np.asarray(vmin, np.float32)),
        width_in=np.float64(-1.0 if width is None else width),
        Vint=Vint.numpy().astype(np.int32), morton=M.numpy().astype(np.uint64),
        keys_sorted=Ms.numpy().astype(np.uint64),
        sort_idx_ref=info["sort_idx"].numpy().astype(np.int64),
        voxel_indices=vox_idx.numpy().astype(np.int64), PCvox=PCvox.numpy().astype(np.float32),
        Nvox=np.int64(info["Nvox"]), vmin=info["vmin"].numpy().astype(np.float32),
        width=np.float64(info["width"]), voxel_size=np.float64(info["voxel_size"]),
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: N={PC32.shape[0]} J={J} Nvox={info['Nvox']}")


def main():
    rng = np.random.default_rng(20251205)

    # ---- tiny / edge cases of the plan (N = 1, 2, 3, 5; early termination before j = 3) --------
    tiny = [
        ("t_n1", [[0, 0, 0]], 1), ("t_n2_j1", [[0, 0, 0], [0, 0, 1]], 1),
        ("t_n2_j3", [[0, 0, 0], [0, 0, 1]], 3), ("t_n2_y", [[0, 0, 0], [0, 1, 0]], 2),
        ("t_n3", [[0, 0, 0], [0, 0, 1], [0, 1, 0]], 2), ("t_n2_far", [[0, 0, 0], [1, 1, 1]], 1),
        ("t_n5_j1", [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0]], 1),
        ("t_n5_j2", [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0]], 2),
        ("t_n2_hi", [[3, 3, 2], [3, 3, 3]], 2),
    ]
    for name, V, J in tiny:
        V = sorted_unique_voxels(np.array(V), J)
        C = rng.normal(size=(V.shape[0], 2)).astype(np.float32)
        transform_case(name, V, J, C)

    # ---- N = 8 full cube at J = 1 and sparse at J = 2 -----------------------------------------
    V = sorted_unique_voxels(np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)]), 1)
    transform_case("n8_cube_j1", V, 1, rng.normal(size=(8, 3)).astype(np.float32), steps=(1,))
    V = sorted_unique_voxels(rng.integers(0, 4, size=(8, 3)), 2)
    transform_case("n8_j2", V, 2, rng.normal(size=(V.shape[0], 3)).astype(np.float32), also_slow_param=True)

    # ---- dense-ish J = 3 (512 cells), D = 11 -------------------------------------------------
    V = sorted_unique_voxels(rng.integers(0, 8, size=(400, 3)), 3)[:257]
    V = sorted_unique_voxels(V, 3)
    transform_case("n257_j3_d11", V, 3, gaussian_attrs(rng, V.shape[0], 11), steps=(1, 8), also_slow_param=True)

    # ---- cfg2-like: J = 10, D = 14 ----------------------------------------------------------
    V = sorted_unique_voxels(blob_cloud(rng, 1000, 10), 10)
    transform_case("n1000_j10_d14", V, 10, gaussian_attrs(rng, V.shape[0], 14), steps=(1, 4))

    # ---- cfg3-like: J = 12, D = 56 and D = 59, with the driver's quantization steps ----------
    V = sorted_unique_voxels(blob_cloud(rng, 1500, 12), 12)
    transform_case("n1500_j12_d56", V, 12, gaussian_attrs(rng, V.shape[0], 56) * 100, steps=(1, 8, 64))
    V = sorted_unique_voxels(blob_cloud(rng, 2000, 10), 10)
    transform_case("n2000_j10_d59", V, 10, gaussian_attrs(rng, V.shape[0], 59))

    # ---- cfg1-like: RGB-ish 3 channels (0..255), J = 18 (encode_ply.py:17-29) -----------------
    V = sorted_unique_voxels(blob_cloud(rng, 3000, 18, nblobs=3, sigma=0.01), 18)
    C = rng.integers(0, 256, size=(V.shape[0], 3)).astype(np.float32)
    transform_case("n3000_j18_d3", V, 18, C, steps=(1, 16))

    # ---- 60-bit keys: J = 20, D = 1 ----------------------------------------------------------
    V = sorted_unique_voxels(rng.integers(0, 1 << 20, size=(500, 3)), 20)
    transform_case("n500_j20_d1", V, 20, rng.normal(size=(V.shape[0], 1)).astype(np.float32))

    # ---- early root: every point shares the high bits (coords < 8 at J = 10) ------------------
    V = sorted_unique_voxels(rng.integers(0, 8, size=(300, 3)), 10)
    transform_case("early_root_j10", V, 10, rng.normal(size=(V.shape[0], 2)).astype(np.float32), steps=(1,))

    # ---- max coordinate present (2^J - 1 on every axis) and origin present -------------------
    J = 6
    V = np.vstack([rng.integers(0, 1 << J, size=(200, 3)), [[63, 63, 63]], [[0, 0, 0]], [[63, 0, 63]]])
    V = sorted_unique_voxels(V, J)
    transform_case("maxcoord_j6", V, J, rng.normal(size=(V.shape[0], 4)).astype(np.float32))

    # ---- single long chain of strictly increasing / decreasing levels ------------------------
    J = 7
    keys = [0] + [1 << k for k in range(0, 21)]              # d = inf,0,1,2,...  (increasing)
    def key_to_xyz(k):
        x = y = z = 0
        for i in range(J):
            dgt = (k >> (3 * i)) & 7
            z |= (dgt & 1) << i; y |= ((dgt >> 1) & 1) << i; x |= ((dgt >> 2) & 1) << i
        return [x, y, z]
    V = sorted_unique_voxels(np.array([key_to_xyz(k) for k in keys]), J)
    transform_case("chain_inc_j7", V, J, rng.normal(size=(V.shape[0], 3)).astype(np.float32))
    top = (1 << 21) - 1
    keys = [0] + [top - ((1 << k) - 1) for k in range(20, -1, -1)]   # decreasing levels
    V = sorted_unique_voxels(np.array([key_to_xyz(k) for k in keys]), J)
    transform_case("chain_dec_j7", V, J, rng.normal(size=(V.shape[0], 3)).astype(np.float32))

    # ---- voxelizer ------------------------------------------------------------------------
    docs = np.array([[0.1, 0.1, 0.1], [0.9, 0.9, 0.9], [0.15, 0.12, 0.08], [0.5, 0.5, 0.5],
                     [0.52, 0.48, 0.51], [0.3, 0.7, 0.2], [0.85, 0.92, 0.88], [0.0, 0.0, 0.0]],
                    dtype=np.float32)                           # docs/voxelization.md:19-95
    voxelize_case("vox_docs8_given", docs, 2, vmin=[0, 0, 0], width=1.0)
    voxelize_case("vox_docs8_auto", docs, 2)
    PC = np.concatenate([rng.uniform(-3, 5, size=(5000, 3)), rng.normal(size=(5000, 3))], axis=1).astype(np.float32)
    voxelize_case("vox_n5000_j6_d3", PC, 6)
    PC = np.concatenate([rng.normal(0, 1, size=(20000, 3)), gaussian_attrs(rng, 20000, 11)], axis=1).astype(np.float32)
    voxelize_case("vox_n20000_j10_d11", PC, 10)
    # adversarial: coordinates sitting (to float32 rounding) on voxel boundaries
    J = 9
    width = 7.3
    vs = width / (1 << J)
    k = rng.integers(0, 1 << J, size=(4000, 3))
    bump = rng.integers(-1, 2, size=(4000, 3))
    base = (k * vs).astype(np.float32)
    pts = np.nextafter(base, np.where(bump > 0, np.float32(np.inf), np.float32(-np.inf)), dtype=np.float32)
    pts = np.where(bump == 0, base, pts).astype(np.float32)
    pts = np.clip(pts, 0, None)
    pts[0] = 0.0
    PC = np.concatenate([pts, rng.normal(size=(4000, 2)).astype(np.float32)], axis=1)
    voxelize_case("vox_boundary_j9", PC, J, vmin=[0, 0, 0], width=width)
    voxelize_case("vox_boundary_auto_j9", PC, J)
    # positions only
    voxelize_case("vox_posonly_j5", rng.uniform(0, 1, size=(3000, 3)).astype(np.float32), 5)


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") == "":
    main()


def rlgr_cases():
    """RLGR byte streams produced by the REFERENCE's own coder (built from its sources by
    `make -C oracle ref` into oracle/_ref/): inputs and expected bytes only."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import rlgr
    rng = np.random.default_rng(424242)

    def enc(x, flag):
        m = rlgr.membuf()
        m.rlgrWrite([int(v) for v in x], flag)
        m.close()
        return np.array(m.get_buffer(), dtype=np.uint8)

    cases = {
        "lap1": (rng.laplace(0, 1, 30000).astype(np.int64), 1),
        "lap30": (rng.laplace(0, 30, 20000).astype(np.int64), 1),
        "zeros": (np.zeros(70000, np.int64), 1),
        "sparse": (((rng.random(60000) < 0.004) * rng.integers(-300, 300, 60000)).astype(np.int64), 1),
        "escape": (rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, 3000).astype(np.int64), 1),
        "int32_extremes": (np.array([-2 ** 31, 2 ** 31 - 1, 0, 0, -1, 1, -2 ** 31, 0, 0, 0, 0, 0, 0, 0, 0, 5], np.int64), 1),
        "single_nonzero": (np.array([7], np.int64), 1),
        "single_zero": (np.array([0], np.int64), 1),
        "run_then_values": (np.concatenate([np.zeros(9000, np.int64), rng.laplace(0, 500, 4000).astype(np.int64), np.zeros(333, np.int64)]), 1),
        "ends_in_open_run": (np.concatenate([rng.laplace(0, 2, 500).astype(np.int64), np.zeros(37, np.int64)]), 1),
        "unsigned": (rng.integers(0, 12, 8000).astype(np.int64), 0),
        "dc_then_small": (np.concatenate([[123456], rng.laplace(0, 4, 12000).astype(np.int64)]).astype(np.int64), 1),
    }
    out = {}
    for name, (x, flag) in cases.items():
        out[name + "__x"] = x.astype(np.int64)
        out[name + "__flag"] = np.int32(flag)
        out[name + "__bytes"] = enc(x, flag)
        # the reference's decoder must invert its own stream
        r = rlgr.membuf(out[name + "__bytes"].tolist())
        _, back = r.rlgrRead(len(x), flag)
        assert back == x.tolist(), name
    np.savez_compressed(os.path.join(HERE, "rlgr_streams.npz"), **out)
    print("rlgr_streams:", {k: (len(v[0]), len(out[k + "__bytes"])) for k, v in cases.items()})


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "rlgr"):
    rlgr_cases()


def pipeline_case():
    """Replays the reference driver's per-frame loop (python/encode_3dgs.py:126-411) with the
    reference's own operators, its own RLGR build (oracle/_ref) and its own PLY writer / reader, on a
    small synthetic voxelized 3DGS frame. Records what the driver logs: bytes per step, PSNRs."""
    import math
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import rlgr
    from data_util import read_compressed_3dgs_ply
    from quality_eval import save_ply

    rng = np.random.default_rng(777)
    J = 10
    V = sorted_unique_voxels(blob_cloud(rng, 2600, J), J)
    N = V.shape[0]
    A = gaussian_attrs(rng, N, 56)                      # quats(4) scales(3) opacity(1) colors(48)
    voxel_size, vmin = 0.0123, torch.tensor([-1.5, 0.25, 3.0])

    # --- PLY round trip through the reference writer / reader (quality_eval.py:18-117, data_util.py:272-382)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "frame.ply")
        save_ply(path, torch.from_numpy(V).float(), torch.from_numpy(A[:, 0:4]), torch.from_numpy(A[:, 4:7]),
                 torch.from_numpy(A[:, 7]), torch.from_numpy(A[:, 8:]), voxel_size=voxel_size, vmin=vmin)
        ply_bytes = np.frombuffer(open(path, "rb").read(), dtype=np.uint8).copy()
        Vr, Ar, vs_r, vmin_r = read_compressed_3dgs_ply(path)
    assert torch.equal(Vr, torch.from_numpy(V)) and np.array_equal(Ar.numpy(), A)

    # --- the driver loop, float64 (encode_3dgs.py:82-83) -------------------------------------------
    steps = [0.004, 0.02, 0.1, 1]
    C = Ar.to(torch.float64)
    Vd = Vr.to(torch.float64)
    origin = torch.zeros(3, dtype=torch.float64)
    List, Flags, weights, order = RAHT_param_reorder_fast(Vd, origin, 2 ** J, J)
    Coeff, _ = RAHT2_optimized(C, List, Flags, weights)
    out = dict(V=V.astype(np.int32), A=A, J=np.int32(J), voxel_size=np.float64(voxel_size), vmin=vmin.numpy(),
               ply_bytes=ply_bytes, steps=np.array(steps, dtype=np.float64))
    size_bytes, psnrs = [], []
    for s in steps:
        enc = torch.floor(Coeff / s + 0.5)                             # :204
        q = enc.index_select(0, order).to(torch.int32).numpy()         # :210-217
        total = 0
        dec_cols = []
        for ch in range(q.shape[1]):                                   # :229-245
            m = rlgr.membuf(); m.rlgrWrite(q[:, ch].tolist(), 1); m.close()
            buf = m.get_buffer(); total += len(buf)
            _, back = rlgr.membuf(buf).rlgrRead(q.shape[0], 1)
            assert back == q[:, ch].tolist()
            dec_cols.append(back)
        dec = torch.from_numpy(np.stack(dec_cols, axis=1).astype(np.int32)).to(torch.float64) * s     # :255-261
        dec = dec[torch.argsort(order), :]                             # :267-268
        rec = inverse_RAHT_optimized(dec, List, Flags, weights)        # :274

        def ps(a, b):
            return -10 * math.log10(torch.mean((a - b) ** 2).item() + 1e-10)   # :298-310
        psnrs.append([ps(C, rec), ps(C[:, 0:4], rec[:, 0:4]), ps(C[:, 4:7], rec[:, 4:7]), ps(C[:, 7], rec[:, 7]), ps(C[:, 8:], rec[:, 8:])])
        size_bytes.append(total)
    out["size_bytes"] = np.array(size_bytes, dtype=np.int64)
    out["psnr"] = np.array(psnrs, dtype=np.float64)                    # all, quats, scales, opacity, colors
    np.savez_compressed(os.path.join(HERE, "pipeline_small.npz"), **out)
    print("pipeline_small: N=%d bytes=%s psnr_all=%s" % (N, size_bytes, [round(p[0], 2) for p in psnrs]))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "pipeline"):
    pipeline_case()


def policy_case():
    """Per-attribute quantization policy of the reference's debug driver (python/encode_3dgs_debug.py:326-381,
    dequantization :433-442) on the pipeline_small frame, with the reference's operators and its RLGR build.
    The driver itself cannot be imported (it runs at import against /ssd1 paths): its arithmetic is restated
    here statement by statement, each line citing the driver line it follows."""
    import math
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import rlgr
    z = np.load(os.path.join(HERE, "pipeline_small.npz"))
    V, A, J = z["V"].astype(np.int64), z["A"], int(z["J"])
    C = torch.from_numpy(A).to(torch.float64)                                   # encode_3dgs_debug.py: DTYPE float64
    Vd = torch.from_numpy(V).to(torch.float64)
    List, Flags, weights, order = RAHT_param_reorder_fast(Vd, torch.zeros(3, dtype=torch.float64), 2 ** J, J)
    Coeff, _ = RAHT2_optimized(C, List, Flags, weights)
    n_channels = Coeff.shape[1]
    groups = [("quats", 0, 4), ("scales", 4, 7), ("opacity", 7, 8), ("colors", 8, n_channels)]        # :328-333
    imp = {"quats": 1.0 / 21.93, "scales": 1.0 / 26.36, "opacity": 1.0 / 42.22, "colors": 1.0 / 38.67}    # :338-343
    total_imp, budget = sum(imp.values()), 1024                                                       # :347-348
    step_of, levels_of = [], []
    enc = torch.zeros_like(Coeff)
    for name, a, b in groups:
        blk = Coeff[:, a:b]
        rng = blk.max() - blk.min()                                              # :357-358
        levels = max(int(budget * imp[name] / total_imp), 2)                     # :361-362
        step = max(rng / max(levels - 1, 1), 1e-6)                               # :365-366
        step = step.item()                                                       # :369
        enc[:, a:b] = torch.floor(Coeff[:, a:b] / step + 0.5)                    # :378-381
        step_of.append(step); levels_of.append(levels)
    q = enc.index_select(0, order).to(torch.int32).numpy()                       # :388-389
    total, cols = 0, []
    for ch in range(n_channels):
        m = rlgr.membuf(); m.rlgrWrite(q[:, ch].tolist(), 1); m.close()
        buf = m.get_buffer(); total += len(buf)
        _, back = rlgr.membuf(buf).rlgrRead(q.shape[0], 1)
        cols.append(back)
    dec = torch.from_numpy(np.stack(cols, axis=1).astype(np.int32)).to(torch.float64)                 # :427-429
    for (name, a, b), step in zip(groups, step_of):
        dec[:, a:b] = dec[:, a:b] * step                                         # :433-436
    dec = dec[torch.argsort(order), :]                                           # :440-441
    rec = inverse_RAHT_optimized(dec, List, Flags, weights)                      # :442

    def ps(x, y):
        return -10 * math.log10(torch.mean((x - y) ** 2).item() + 1e-10)         # :445-447
    psnr = [ps(C, rec), ps(C[:, 0:4], rec[:, 0:4]), ps(C[:, 4:7], rec[:, 4:7]), ps(C[:, 7], rec[:, 7]), ps(C[:, 8:], rec[:, 8:])]
    np.savez_compressed(os.path.join(HERE, "pipeline_policy.npz"), steps=np.array(step_of), levels=np.array(levels_of, dtype=np.int64),
                        group_start=np.array([g[1] for g in groups], dtype=np.int64), group_end=np.array([g[2] for g in groups], dtype=np.int64),
                        q=q, size_bytes=np.int64(total), psnr=np.array(psnr))
    print("pipeline_policy: steps=%s levels=%s bytes=%d psnr_all=%.3f" % ([round(s, 5) for s in step_of], levels_of, total, psnr[0]))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "policy"):
    policy_case()


def encode_ply_case():
    """BASELINE configs[0]: the reference's encode_ply.py loop (:102-222) on a 10k-point RGB cloud
    (positions U[0,1)^3 voxelized at J=10, RGB randint(0,256), seed 0), with the
    reference's rgb_to_yuv (utils.py:4-33), operators and RLGR build. Records bytes and Y-PSNR per step."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
    import rlgr
    from utils import rgb_to_yuv
    rng = np.random.default_rng(0)
    J = 10
    P = rng.uniform(0, 1, size=(10000, 3))
    rgb_all = rng.integers(0, 256, size=(10000, 3))
    Vall = np.floor(P * (1 << J)).astype(np.int64)
    V, first = np.unique(Vall, axis=0, return_index=True)
    mc = morton_np(V, J)
    o = np.argsort(mc, kind="stable")
    V, rgb = V[o], rgb_all[first][o].astype(np.float32)
    N = V.shape[0]
    Cyuv = rgb_to_yuv(torch.from_numpy(rgb).to(torch.float64)).contiguous()
    Vd = torch.from_numpy(V).to(torch.float64)
    origin = torch.zeros(3, dtype=torch.float64)
    List, Flags, weights, order = RAHT_param_reorder_fast(Vd, origin, 2 ** J, J)
    Coeff, _ = RAHT2_optimized(Cyuv, List, Flags, weights)
    steps = [1, 2, 4, 6, 8, 12, 16, 20, 24, 32, 64]                 # encode_ply.py:29
    sizes, psnrs, recs = [], [], {}
    for s in steps:
        enc = torch.floor(Coeff / s + 0.5)
        Y_hat = enc[:, 0] * s
        mse = (torch.linalg.norm(Coeff[:, 0] - Y_hat) ** 2) / (N * 255 ** 2)      # :150-151
        psnrs.append(float(-10 * torch.log10(mse)))
        q = enc.index_select(0, order).to(torch.int32).numpy()
        total = 0
        for ch in range(3):
            m = rlgr.membuf(); m.rlgrWrite(q[:, ch].tolist(), 1); m.close()
            total += len(m.get_buffer())
        sizes.append(total)
        if s in (1, 16):
            dec = torch.from_numpy(q).to(torch.float64) * s
            recs[s] = inverse_RAHT_optimized(dec[torch.argsort(order), :], List, Flags, weights).numpy()
    np.savez_compressed(os.path.join(HERE, "pipeline_ply_rgb.npz"), V=V.astype(np.int32), rgb=rgb, J=np.int32(J),
                        yuv=Cyuv.numpy(), steps=np.array(steps, dtype=np.float64), size_bytes=np.array(sizes, dtype=np.int64),
                        psnr_y=np.array(psnrs), crec_step1=recs[1], crec_step16=recs[16])
    print("pipeline_ply_rgb: N=%d bytes=%s psnr=%s" % (N, sizes, [round(p, 2) for p in psnrs]))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "ply"):
    encode_ply_case()


def voxel_residual_cases():
    """The voxelizer's secondary outputs PCsorted / DeltaPC (reference python/voxelize_pc.py:103-111, 147-156), straight
    from voxelize_pc_batched on the CPU. torch.sort (:101) is not asked to be stable and is not (equal keys come out in
    another order than they went in), so the fixtures keep the reference's own permutation: consumers compare per POINT
    (row k of the reference <-> point sort_idx[k]), not per sorted position."""
    rng = np.random.default_rng(20262)

    def case(name, PC32, J, vmin=None, width=None):
        PC = torch.from_numpy(PC32)
        vm = None if vmin is None else torch.tensor(vmin, dtype=torch.float32)
        PCvox, PCsorted, vox_idx, DeltaPC, info = voxelize_pc_batched(PC, vm, width, J, device="cpu")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), PC=PC32, J=np.int32(J),
                            vmin_in=(np.zeros(0, np.float32) if vmin is None else np.asarray(vmin, np.float32)),
                            width_in=np.float64(-1.0 if width is None else width),
                            sort_idx=info["sort_idx"].numpy().astype(np.int64), voxel_indices=vox_idx.numpy().astype(np.int64),
                            PCvox=PCvox.numpy().astype(np.float32), PCsorted=PCsorted.numpy().astype(np.float32),
                            DeltaPC=DeltaPC.numpy().astype(np.float32), vmin=info["vmin"].numpy().astype(np.float32),
                            width=np.float64(info["width"]), voxel_size=np.float64(info["voxel_size"]))
        print(f"{name}: N={PC32.shape[0]} J={J} Nvox={info['Nvox']}")

    PC = np.concatenate([rng.uniform(-3, 5, size=(3000, 3)), rng.normal(size=(3000, 4))], axis=1).astype(np.float32)
    PC[::3, :3] = PC[1::3, :3][: PC[::3].shape[0]]                 # several points per voxel
    case("voxres_n3000_j5_d4", PC, 5)
    case("voxres_n3000_j7_given", PC, 7, vmin=[-3.5, -3.5, -3.5], width=9.0)
    case("voxres_posonly_j4", rng.uniform(0, 1, size=(2000, 3)).astype(np.float32), 4)
    # voxel-face positions, non-power-of-two width (the division-semantics case of DESIGN.md 13)
    J, width = 6, 7.3
    vs = width / (1 << J)
    k = rng.integers(0, 1 << J, size=(1500, 3))
    base = (k * vs).astype(np.float32)
    bump = rng.integers(-1, 2, size=(1500, 3))
    pts = np.where(bump == 0, base, np.nextafter(base, np.where(bump > 0, np.float32(np.inf), np.float32(-np.inf)), dtype=np.float32)).astype(np.float32)
    pts = np.clip(pts, 0, None)
    case("voxres_boundary_j6", np.concatenate([pts, rng.normal(size=(1500, 2)).astype(np.float32)], axis=1), J, vmin=[0, 0, 0], width=width)


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "voxres"):
    voxel_residual_cases()


def mixed_cases():
    """Frames as the voxelizer hands them over: columns 0-2 are the integer voxel coordinates (python/voxelize_pc.py:155),
    the rest 3DGS attributes; quantized at the fine steps unit-range attributes need, where the xyz columns' quotients
    exceed 2^24 (what raht_fwd_quant_mixed / raht_dequant_inv_mixed exist for)."""
    rng = np.random.default_rng(20261005)
    for name, n, J, D in (("mx_n2000_j10_d59", 2000, 10, 59), ("mx_n1500_j12_d59", 1500, 12, 59), ("mx_n1000_j10_d14", 1000, 10, 14)):
        V = sorted_unique_voxels(blob_cloud(rng, n, J), J)
        C = np.concatenate([V.astype(np.float32), gaussian_attrs(rng, V.shape[0], D - 3)], axis=1)
        transform_case(name, V, J, C, steps=(0.01, 1))


if __name__ == "__main__" and os.environ.get("GOLDEN_ONLY", "") in ("", "mixed"):
    mixed_cases()
