"""GPU parity tests: the HIP path (through the C ABI) against golden vectors from the reference and
against the CPU oracle on seeded inputs. Run on the MI355X box with ``pytest -m gpu``.

Bars (SURVEY.md 8c / DESIGN.md):
  integers (Morton keys, List/Flags/weights, order_RAGFT, w, voxel indices/coords)  bit-exact
  float64 kernels vs reference                                                       rtol = atol = 1e-12
  float32 kernels vs float64 reference, per column c:
      max|T32 - T64| <= 2e-6 * max|T64[:, c]|      rms <= 1e-6 * rms(T64[:, c])  (floor 1e-30)
      fwd -> inv round trip <= 1e-5 * max|C|
  quantized ints: mismatches only +-1 and only next to a rounding tie of the reference value
"""
import os

import numpy as np
import pytest

from .conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

TRANSFORM = golden_names(exclude_prefix="vox_")
VOX = golden_names(prefix="vox_")
# (engine, stage-0 tile rows, later-stage tile rows, later-stage channels per chunk); 0 = automatic.
# Small explicit tiles force multi-stage / multi-tile / channel-chunked schedules on the small fixtures.
# last entry: finishing single-tile stage up to that many rows (64 keeps the small fixtures multi-stage)
ENGINES = [("tile", 0, 0, 0, 0), ("tile", 64, 64, 0, 64), ("tile", 128, 64, 8, 64), ("tile", 64, 256, 3, 0), ("tile", 64, 64, 0, 0),
           ("level", 0, 0, 0, 0)]


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import _lib
    _lib.lib()                      # must load: no fallback
    return R


def _dev(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _plan(R, g, engine="tile", tile_rows=0, tail_rows=0, tail_ch=0, final_rows=0):
    import torch
    J = int(g["J"])
    V = _dev(g["V"].astype(np.float64))
    p = R.RahtPlan.from_coords(V, torch.zeros(3, dtype=torch.float64), 2 ** J, J)
    p.set_engine(engine, tile_rows, tail_rows, tail_ch, final_rows)
    return p


def _check_f32(T32, T64, what):
    T32 = np.asarray(T32, dtype=np.float64)
    colmax = np.abs(T64).max(axis=0)
    err = np.abs(T32 - T64).max(axis=0)
    assert np.all(err <= 2e-6 * np.maximum(colmax, 1e-30)), (what, float((err / np.maximum(colmax, 1e-30)).max()))
    rms = np.sqrt(((T32 - T64) ** 2).mean(axis=0))
    ref = np.sqrt((T64 ** 2).mean(axis=0))
    assert np.all(rms <= 1e-6 * np.maximum(ref, 1e-30)), (what, float((rms / np.maximum(ref, 1e-30)).max()))


# ------------------------------------------------------------------------------------------------ plan
@pytest.mark.parametrize("name", TRANSFORM)
def test_plan_lists_and_order_match_reference(rt, name):
    g = load_golden(name)
    p = _plan(rt, g)
    List, Flags, weights = p.export_lists()
    assert len(List) == len(g["List"])
    for l in range(len(List)):
        assert np.array_equal(List[l].numpy(), g["List"][l]), f"List[{l}]"
        assert np.array_equal(Flags[l].numpy(), g["Flags"][l]), f"Flags[{l}]"
        assert np.array_equal(weights[l].numpy(), g["weights"][l]), f"weights[{l}]"
    keys, lvl, wl, wr = p.arrays()
    assert np.array_equal(keys, g["morton"])
    N = g["V"].shape[0]
    order = p.order_RAGFT.cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(N))
    ref_sane = (not bool(g["order_is_none"])) and np.array_equal(np.sort(g["order"]), np.arange(N))
    if ref_sane:
        assert np.array_equal(order, g["order"])
    # sum of pairs = N - 1; every row but row 0 is a right sibling exactly once
    assert lvl[0] == 255 and np.all(lvl[1:] < 64)
    assert np.all(wl[1:] >= 1) and np.all(wr[1:] >= 1)


def test_plan_tokens_behave_like_the_reference_lists(rt):
    """The drivers re-map the lists with .to(device) and pass them back opaquely (encode_3dgs.py:153-159)."""
    import torch
    g = load_golden("n1000_j10_d14")
    J = int(g["J"])
    V = _dev(g["V"].astype(np.float64))
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device="cuda")
    ListC, FlagsC, weightsC, order = rt.raht_fn["RAHT_param"](V, origin, 2 ** J, J)
    ListC = [t.to(device="cuda", non_blocking=True) for t in ListC]
    FlagsC = [t.to(device="cuda", non_blocking=True) for t in FlagsC]
    weightsC = [t.to(device="cuda", non_blocking=True) for t in weightsC]
    C = _dev(g["C"].astype(np.float64))
    Coeff, w = rt.raht_fn["RAHT"](C, ListC, FlagsC, weightsC)
    assert Coeff.dtype == torch.float64 and w.shape == (C.shape[0], 1)
    np.testing.assert_allclose(Coeff.cpu().numpy(), g["T"], rtol=1e-12, atol=1e-12)
    Crec = rt.raht_fn["iRAHT"](Coeff, ListC, FlagsC, weightsC)
    assert torch.allclose(C, Crec, rtol=1e-5, atol=1e-8)              # encode_3dgs.py:195 (strict)
    assert np.array_equal(order.cpu().numpy(), g["order"])
    # a copied token (different storage) still resolves through its contents
    Coeff2, _ = rt.raht_fn["RAHT"](C, [ListC[0].clone()], FlagsC, weightsC)
    assert torch.equal(Coeff, Coeff2)


# ------------------------------------------------------------------------------------------- transform
@pytest.mark.parametrize("eng", ENGINES)
@pytest.mark.parametrize("name", TRANSFORM)
def test_forward_inverse_f64(rt, name, eng):
    g = load_golden(name)
    p = _plan(rt, g, *eng)
    C = _dev(g["C"].astype(np.float64))
    T, w = p.forward(C)
    # (frames with xyz columns, mx_*: inputs up to 4095, coefficients up to 1e5 -- the absolute part of the bar scales with the
    # column's magnitude there, as in tests/test_gpu_fullsize.py; a high-pass coefficient is a difference of such numbers)
    atol = 1e-12 * np.maximum(1.0, np.abs(g["T"]).max(axis=0)) if name.startswith("mx_") else 1e-12
    assert np.all(np.abs(T.cpu().numpy() - g["T"]) <= atol + 1e-12 * np.abs(g["T"])), name
    assert np.array_equal(w.cpu().numpy().reshape(-1), g["w"])
    Crec = p.inverse(T)
    np.testing.assert_allclose(Crec.cpu().numpy(), g["C"].astype(np.float64), rtol=1e-12,
                               atol=1e-12 * max(1.0, float(np.abs(g["C"]).max())))


@pytest.mark.parametrize("eng", ENGINES)
@pytest.mark.parametrize("name", TRANSFORM)
def test_forward_inverse_f32(rt, name, eng):
    g = load_golden(name)
    p = _plan(rt, g, *eng)
    C = _dev(g["C"])
    T, w = p.forward(C)
    _check_f32(T.cpu().numpy(), g["T"], name)
    assert np.array_equal(w.cpu().numpy().reshape(-1).astype(np.float64), g["w"])
    Crec = p.inverse(T).cpu().numpy()
    assert np.abs(Crec - g["C"]).max() <= 1e-5 * max(float(np.abs(g["C"]).max()), 1e-30)
    # inverse of the REFERENCE coefficients (decoder side on its own)
    C2 = p.inverse(_dev(g["T"].astype(np.float32))).cpu().numpy()
    assert np.abs(C2 - g["C"]).max() <= 1e-5 * max(float(np.abs(g["C"]).max()), 1e-30)


def test_strided_and_unaligned_rows(rt):
    import torch
    g = load_golden("n2000_j10_d59")
    p = _plan(rt, g)
    N, D = g["C"].shape
    big = torch.zeros((N, 64), dtype=torch.float32, device="cuda")
    big[:, :D] = _dev(g["C"])
    T1, _ = p.forward(big[:, :D])                 # row stride 64
    T0, _ = p.forward(_dev(g["C"]))
    assert torch.equal(T0, T1)
    flat = torch.zeros(N * D + 1, dtype=torch.float32, device="cuda")
    flat[1:] = _dev(g["C"]).reshape(-1)
    T2, _ = p.forward(flat[1:].view(N, D))        # base pointer only 4-byte aligned
    assert torch.equal(T0, T2)
    assert torch.equal(p.inverse(T0), p.inverse(flat[1:].view(N, D) * 0 + T0))


def test_many_channels_are_chunked(rt):
    """D > 64 is split into channel chunks (the transform is independent per channel)."""
    import torch
    g = load_golden("n1000_j10_d14")
    p = _plan(rt, g)
    C14 = _dev(g["C"].astype(np.float64))
    C = torch.cat([C14] * 10, dim=1)[:, :131].contiguous()
    T, _ = p.forward(C)
    ref = np.concatenate([g["T"]] * 10, axis=1)[:, :131]
    np.testing.assert_allclose(T.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(p.inverse(T).cpu().numpy(), C.cpu().numpy(), rtol=1e-12, atol=1e-12)


# ---------------------------------------------------------------------------------------------- quant
@pytest.mark.parametrize("name", [n for n in TRANSFORM if any(k.startswith("q_step") for k in load_golden(n))])
def test_quantize_reorder_and_back(rt, name):
    g = load_golden(name)
    p = _plan(rt, g)
    T, _ = p.forward(_dev(g["C"]))
    order = g["order"]
    for key in [k for k in g if k.startswith("q_step")]:
        step = float(key[len("q_step"):])
        Q = p.quant_reorder(T, step).cpu().numpy()
        ref = g[key]
        bad = np.argwhere(Q != ref)
        pre = g["T"][order] / step + 0.5
        scale = np.abs(g["T"]).max(axis=0) / step
        if name.startswith("mx_"):
            # frames with xyz columns at fine steps: |T| / step exceeds 2^24 there, float32 integers are not the reference's
            # (include/raht.h at raht_fwd_quant; tests/test_gpu_mixed.py covers these columns through raht_fwd_quant_mixed)
            bad = bad[(bad[:, 1] >= 3) | (scale[bad[:, 1]] < 2.0 ** 22)]
        for r, c in bad:
            assert abs(int(Q[r, c]) - int(ref[r, c])) == 1
            # fp32 coefficient error (<= 2e-6 of the column max) can only flip values that close to a tie
            assert abs(pre[r, c] - np.round(pre[r, c])) <= 4e-6 * max(scale[c], 1.0), (key, r, c)
        # decoder side from the reference integers
        Td = p.dequant_unreorder(_dev(ref), step)
        Crec = p.inverse(Td).cpu().numpy()
        ref_rec = g["crec_step" + key[len("q_step"):]]
        assert np.abs(Crec - ref_rec).max() <= 1e-5 * max(float(np.abs(ref_rec).max()), 1.0)
    # per-channel steps
    D = g["C"].shape[1]
    steps = [1.0 + 0.5 * c for c in range(D)]
    Qc = p.quant_reorder(T, steps).cpu().numpy()
    Tn = T.cpu().numpy()
    exp = np.floor(Tn[order] / np.asarray(steps, np.float32) + np.float32(0.5)).astype(np.int32)
    assert np.array_equal(Qc, exp)


@pytest.mark.parametrize("tile_rows,tail_rows,tail_ch,final_rows", [(0, 0, 0, 0), (64, 64, 0, 64), (128, 64, 8, 64), (64, 256, 3, 0)])
@pytest.mark.parametrize("name", ["n257_j3_d11", "n1000_j10_d14", "n1500_j12_d56", "n3000_j18_d3", "early_root_j10", "n8_cube_j1"])
def test_fused_quant_equals_two_call_sequence(rt, name, tile_rows, tail_rows, tail_ch, final_rows):
    """raht_fwd_quant == raht_fwd + raht_quant_reorder and raht_dequant_inv == dequant + raht_inv, bit for bit."""
    import torch
    g = load_golden(name)
    p = _plan(rt, g, "tile", tile_rows, tail_rows, tail_ch, final_rows)
    C = _dev(g["C"])
    D = C.shape[1]
    for steps in (1.0, 0.37, [0.5 + 0.25 * c for c in range(D)]):
        T, _ = p.forward(C)
        Q2 = p.quant_reorder(T, steps)
        Q1 = p.forward_quant(C, steps)
        assert torch.equal(Q1, Q2)
        C2 = p.inverse(p.dequant_unreorder(Q2, steps))
        C1 = p.dequant_inverse(Q1, steps)
        assert torch.equal(C1, C2)


@pytest.mark.parametrize("tile_rows,tail_rows,tail_ch,final_rows", [(0, 0, 0, 0), (64, 64, 0, 64), (128, 64, 8, 64), (64, 256, 3, 0)])
@pytest.mark.parametrize("name", ["n257_j3_d11", "n1000_j10_d14", "n1500_j12_d56", "n2000_j10_d59", "n3000_j18_d3", "early_root_j10", "n8_cube_j1"])
def test_fused_quant_float64_equals_two_call_sequence(rt, name, tile_rows, tail_rows, tail_ch, final_rows):
    """The same at the reference's precision: raht_fwd_quant_f64 (float64 tile kernels with the float64 quantizer in their
    write-back) == raht_fwd_f64 + raht_quant_reorder_f64, raht_dequant_inv_f64 == dequant + raht_inv_f64, bit for bit --
    and the integers are floor(T64 / step + 0.5) of the float64 coefficients (encode_3dgs.py:204,210,215)."""
    import torch
    g = load_golden(name)
    p = _plan(rt, g, "tile", tile_rows, tail_rows, tail_ch, final_rows)
    C = _dev(g["C"].astype(np.float64))
    D = C.shape[1]
    order = p.order_RAGFT.cpu().numpy()
    for steps in (1.0, 0.37, [0.5 + 0.25 * c for c in range(D)]):
        T, _ = p.forward(C)
        assert T.dtype == torch.float64
        Q2 = p.quant_reorder(T, steps)
        Q1 = p.forward_quant(C, steps)
        assert Q1.dtype == torch.int32 and torch.equal(Q1, Q2)
        exp = np.floor(T.cpu().numpy()[order] / np.asarray(steps, np.float64) + 0.5).astype(np.int32)
        assert np.array_equal(Q1.cpu().numpy(), exp)
        C2 = p.inverse(p.dequant_unreorder(Q2, steps, dtype=torch.float64))
        C1 = p.dequant_inverse(Q1, steps, dtype=torch.float64)
        assert C1.dtype == torch.float64 and torch.equal(C1, C2)


def test_fused_quant_strided_input(rt):
    import torch
    g = load_golden("n2000_j10_d59")
    p = _plan(rt, g)
    N, D = g["C"].shape
    big = torch.zeros((N, 64), dtype=torch.float32, device="cuda")
    big[:, :D] = _dev(g["C"])
    assert torch.equal(p.forward_quant(big[:, :D], 0.01), p.forward_quant(_dev(g["C"]), 0.01))


@pytest.mark.parametrize("geom", [(0, 0, 0, 0), (64, 64, 0, 64)])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_strided_rows_through_every_tile_kernel(rt, geom, dtype):
    """Row stride > D on the input side of every direction (the tile kernels' LDS-direct loads take one global address per
    lane: row * ld + channel): forward, inverse, fused forward -- same results as from contiguous matrices, bit for bit."""
    import torch
    g = load_golden("n2000_j10_d59")
    p = _plan(rt, g, "tile", *geom)
    td = torch.float32 if dtype == "f32" else torch.float64
    C = _dev(g["C"]).to(td)
    N, D = C.shape
    wide = torch.full((N, D + 7), float("nan"), dtype=td, device="cuda")
    wide[:, :D] = C
    T = p.forward(C, want_w=False)
    assert torch.equal(p.forward(wide[:, :D], want_w=False), T)
    wideT = torch.full((N, D + 3), float("nan"), dtype=td, device="cuda")
    wideT[:, :D] = T
    assert torch.equal(p.inverse(wideT[:, :D]), p.inverse(T))
    assert torch.equal(p.forward_quant(wide[:, :D], 0.02), p.forward_quant(C, 0.02))


# ------------------------------------------------------------------------------------------- voxelizer
@pytest.mark.parametrize("name", VOX)
def test_voxelize(rt, name):
    g = load_golden(name)
    vmin = None if g["vmin_in"].size == 0 else g["vmin_in"].tolist()
    width = None if float(g["width_in"]) < 0 else float(g["width_in"])
    J = int(g["J"])
    PC = _dev(g["PC"])
    PCvox, PCsorted, vidx, DeltaPC, info = rt.voxelize_pc_batched(PC, vmin, width, J, device="cuda")
    assert info["Nvox"] == int(g["Nvox"])
    assert np.array_equal(info["keys_sorted"].cpu().numpy().view(np.uint64), g["keys_sorted"])
    assert np.array_equal(vidx.cpu().numpy(), g["voxel_indices"])
    assert np.array_equal(info["vmin"].cpu().numpy(), g["vmin"])
    assert info["width"] == float(g["width"]) and info["voxel_size"] == float(g["voxel_size"])
    pv = PCvox.cpu().numpy()
    assert np.array_equal(pv[:, :3], g["PCvox"][:, :3])
    np.testing.assert_allclose(pv[:, 3:], g["PCvox"][:, 3:], rtol=2e-6, atol=1e-6)
    idx = info["sort_idx"].cpu().numpy()
    assert np.array_equal(np.sort(idx), np.arange(g["PC"].shape[0]))
    assert np.array_equal(g["morton"][idx], g["keys_sorted"])
    # stable: equal keys keep their input order
    same = g["keys_sorted"][1:] == g["keys_sorted"][:-1]
    assert np.all(idx[1:][same] > idx[:-1][same])
    assert np.array_equal(PCsorted.cpu().numpy(), g["PC"][idx])
    m = rt.get_morton_code(_dev(g["Vint"].astype(np.int64)), J).cpu().numpy().view(np.uint64)
    assert np.array_equal(m, g["morton"])


@pytest.mark.parametrize("name", golden_names(prefix="voxres_"))
def test_voxelizer_residuals(rt, oracle, name):
    """PCsorted / DeltaPC (voxelize_pc.py:103-111, 147-156) through raht_voxelize_residuals: bit-identical to the oracle
    (same stable order, same float32 arithmetic), and per point against the reference's own outputs."""
    from .test_oracle_golden import check_residuals_against_reference
    g = load_golden(name)
    vmin = None if g["vmin_in"].size == 0 else g["vmin_in"].tolist()
    width = None if float(g["width_in"]) < 0 else float(g["width_in"])
    PCvox, PCsorted, vidx, DeltaPC, info = rt.voxelize_pc_batched(_dev(g["PC"]), vmin, width, int(g["J"]), device="cuda")
    r = oracle.voxelize(g["PC"], int(g["J"]), vmin=None if vmin is None else g["vmin_in"], width=width)
    pcs, dl = oracle.voxel_residuals(g["PC"], r)
    assert np.array_equal(info["sort_idx"].cpu().numpy(), r["sort_idx"])
    np.testing.assert_array_equal(PCsorted.cpu().numpy(), pcs)
    np.testing.assert_array_equal(DeltaPC.cpu().numpy(), dl)
    check_residuals_against_reference(g, PCsorted.cpu().numpy(), DeltaPC.cpu().numpy(), r["sort_idx"], vidx.cpu().numpy())
    # the optional outputs stay optional
    out = rt.voxelize_pc_batched(_dev(g["PC"]), vmin, width, int(g["J"]), device="cuda", residuals=True, sorted_points=False)
    assert out[1] is None and np.array_equal(out[3].cpu().numpy(), dl)
    out = rt.voxelize_pc_batched(_dev(g["PC"]), vmin, width, int(g["J"]), device="cuda", residuals=False, sorted_points=True)
    assert out[3] is None and np.array_equal(out[1].cpu().numpy(), pcs)


@pytest.mark.parametrize("d,J,n", [(5, 5, 30000), (11, 6, 40000), (56, 9, 60000), (56, 4, 20000), (61, 7, 30000), (70, 6, 20000), (130, 6, 9000)])
def test_voxelizer_all_outputs_in_one_pass(rt, oracle, d, J, n):
    """raht_voxelize_all (clouds with >= 5 attribute columns: means, PCsorted and DeltaPC from ONE pass over the gathered rows)
    against the oracle's voxelizer bit for bit, and against the two-call sequence raht_voxelize + raht_voxelize_residuals;
    voxels with many points (J = 4: ~5 per voxel), row lengths that are / are not multiples of four, rows longer than a wave's
    64 chunks' worth of lanes."""
    import ctypes as C
    import torch
    from raht_3dgs_codec_amd import _lib
    rng = np.random.default_rng(1000 + d + J)
    P = (rng.random((n, 3)) * 3.0 - 0.5).astype(np.float32)
    P[::5] = P[1::5][: P[::5].shape[0]]                       # exact duplicates as well
    PC = np.concatenate([P, rng.standard_normal((n, d)).astype(np.float32)], axis=1)
    PCvox, PCsorted, vidx, DeltaPC, info = rt.voxelize_pc_batched(_dev(PC), None, None, J, device="cuda")
    r = oracle.voxelize(PC, J)
    pcs, dl = oracle.voxel_residuals(PC, r)
    assert info["Nvox"] == r["Nvox"]
    assert np.array_equal(info["sort_idx"].cpu().numpy(), r["sort_idx"]) and np.array_equal(vidx.cpu().numpy(), r["voxel_indices"])
    np.testing.assert_array_equal(PCvox.cpu().numpy(), r["PCvox"])
    np.testing.assert_array_equal(PCsorted.cpu().numpy(), pcs)
    np.testing.assert_array_equal(DeltaPC.cpu().numpy(), dl)
    # the two-call sequence through the C ABI
    L = _lib.lib()
    vp = C.c_void_p
    PCd = _dev(PC)
    ld = 3 + d
    keys = torch.empty(n, dtype=torch.int64, device="cuda"); idx = torch.empty_like(keys); vi2 = torch.empty_like(keys)
    pcv2 = torch.empty((n, ld), dtype=torch.float32, device="cuda")
    nv = C.c_int64(); vmin = (C.c_float * 3)(); w = C.c_double(); vs = C.c_double()
    _lib.check(L.raht_voxelize(vp(PCd.data_ptr()), ld, n, d, None, -1.0, J, vp(keys.data_ptr()), vp(idx.data_ptr()), vp(vi2.data_ptr()),
                               vp(pcv2.data_ptr()), None, C.byref(nv), vmin, C.byref(w), C.byref(vs), None))
    pcs2 = torch.empty((n, ld), dtype=torch.float32, device="cuda"); dl2 = torch.empty_like(pcs2)
    _lib.check(L.raht_voxelize_residuals(vp(PCd.data_ptr()), ld, n, d, vp(keys.data_ptr()), vp(idx.data_ptr()), vp(pcv2.data_ptr()), vmin,
                                         vs.value, vp(pcs2.data_ptr()), vp(dl2.data_ptr()), None))
    torch.cuda.synchronize()
    assert nv.value == r["Nvox"] and torch.equal(pcv2[: nv.value], PCvox) and torch.equal(pcs2, PCsorted) and torch.equal(dl2, DeltaPC)


@pytest.mark.parametrize("n,d,J,mode", [(1, 6, 5, "rand"), (2, 6, 5, "same"), (63, 9, 3, "rand"), (5000, 7, 1, "rand"), (4097, 5, 6, "same"),
                                        (2049, 56, 10, "rand"), (300, 0, 4, "rand"), (300, 3, 4, "same")])
def test_voxelizer_edge_shapes(rt, oracle, n, d, J, mode):
    """One point, two identical points, every point in one voxel (one run of equal keys spanning several blocks of the run-start
    kernels), a cloud of one voxel per point, position-only clouds, fewer points than a wave -- all outputs against the oracle."""
    rng = np.random.default_rng(n * 31 + d)
    P = (rng.random((n, 3)) * 2.0).astype(np.float32)
    if mode == "same":
        P[1:] = P[0]
    PC = np.concatenate([P, rng.standard_normal((n, d)).astype(np.float32)], axis=1)
    vmin, width = ([0.0, 0.0, 0.0], 2.0) if (n < 3 or mode == "same") else (None, None)          # (identical points have no extent of their own)
    PCvox, PCsorted, vidx, DeltaPC, info = rt.voxelize_pc_batched(_dev(PC), vmin, width, J, device="cuda")
    r = oracle.voxelize(PC, J, vmin=None if vmin is None else np.asarray(vmin, np.float32), width=width)
    pcs, dl = oracle.voxel_residuals(PC, r)
    assert info["Nvox"] == r["Nvox"] and (mode != "same" or info["Nvox"] == 1)
    assert np.array_equal(info["sort_idx"].cpu().numpy(), r["sort_idx"]) and np.array_equal(vidx.cpu().numpy(), r["voxel_indices"])
    np.testing.assert_array_equal(PCvox.cpu().numpy(), r["PCvox"])
    np.testing.assert_array_equal(PCsorted.cpu().numpy(), pcs)
    np.testing.assert_array_equal(DeltaPC.cpu().numpy(), dl)
    if d >= 1 and n >= 2:
        PCvox2, plan, info2 = rt.voxelize_plan(_dev(PC), vmin, width, J)
        assert plan.N == r["Nvox"] and np.array_equal(PCvox2.cpu().numpy(), r["PCvox"])


def test_voxelize_matches_oracle_bitwise(rt, oracle):
    """Same stable order and sequential float32 sums as the C oracle -> means are bit-identical."""
    rng = np.random.default_rng(5)
    PC = np.concatenate([rng.normal(0, 1, (50000, 3)), rng.normal(0, 1, (50000, 7))], axis=1).astype(np.float32)
    r = oracle.voxelize(PC, 7)
    PCvox, _, vidx, _, info = rt.voxelize_pc_batched(_dev(PC), None, None, 7, device="cuda", residuals=False)
    assert info["Nvox"] == r["Nvox"] and r["Nvox"] < 50000
    assert np.array_equal(info["sort_idx"].cpu().numpy(), r["sort_idx"])
    assert np.array_equal(vidx.cpu().numpy(), r["voxel_indices"])
    assert np.array_equal(PCvox.cpu().numpy(), r["PCvox"])


def test_radix_sort_60bit_keys(rt):
    import torch
    rng = np.random.default_rng(11)
    k = rng.integers(0, 1 << 60, size=300001, dtype=np.int64)
    k[::7] = k[3]                                       # many duplicates -> stability matters
    ks, idx = rt.sort_keys(_dev(k), nbits=60)
    o = np.argsort(k, kind="stable")
    assert np.array_equal(idx.cpu().numpy(), o)
    assert np.array_equal(ks.cpu().numpy(), k[o])


# ------------------------------------------------------------------------------------------ edge cases
def test_rejects_bad_input(rt):
    import torch
    g = load_golden("n1000_j10_d14")
    J = int(g["J"])
    V = g["V"].astype(np.float64)
    z = torch.zeros(3, dtype=torch.float64)
    with pytest.raises(rt.RahtError) as e:
        rt.RahtPlan.from_coords(_dev(V[::-1].copy()), z, 2 ** J, J)          # unsorted
    assert e.value.code == -2
    Vd = V.copy(); Vd[500] = Vd[499]
    with pytest.raises(rt.RahtError) as e:
        rt.RahtPlan.from_coords(_dev(Vd), z, 2 ** J, J)                      # duplicate voxel
    assert e.value.code == -2 and "row 500" in str(e.value)
    Vb = V.copy(); Vb[-1, 0] = 2 ** J
    with pytest.raises(rt.RahtError) as e:
        rt.RahtPlan.from_coords(_dev(Vb), z, 2 ** J, J)                      # out of bounds
    assert e.value.code == -3
    with pytest.raises(RuntimeError):
        rt.RAHT_param_reorder_fast(torch.from_numpy(V), z, 2 ** J, J)        # CPU tensor: no CPU path
    p = rt.RahtPlan.from_coords(_dev(V), z, 2 ** J, J)
    with pytest.raises(ValueError):
        p.forward(torch.zeros((5, 3), device="cuda"))


def test_single_point(rt):
    import torch
    p = rt.RahtPlan.from_coords(torch.zeros((1, 3), dtype=torch.float64, device="cuda"), [0, 0, 0], 2, 1)
    C = torch.tensor([[1.5, -2.0]], device="cuda")
    T, w = p.forward(C)
    assert torch.equal(T, C) and w.item() == 1.0 and p.order_RAGFT.tolist() == [0]
    assert torch.equal(p.inverse(T), C)


# ------------------------------------------------------------------------ oracle on larger seeded inputs
@pytest.mark.parametrize("n,J,D,seed", [(60000, 10, 59, 3), (200000, 12, 14, 4), (30000, 18, 3, 5)])
def test_against_oracle_seeded(rt, oracle, n, J, D, seed):
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, C = synth.scene(n, J, D, seed)
    po = oracle.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
    To, wo = oracle.raht_fwd(C.astype(np.float64), po)
    p = rt.RahtPlan.from_coords(_dev(V.astype(np.float64)), [0, 0, 0], 2 ** J, J)
    assert np.array_equal(p.order_RAGFT.cpu().numpy(), po.order)
    assert p.levels == po.nlevels
    T, w = p.forward(_dev(C))
    _check_f32(T.cpu().numpy(), To, "seeded")
    assert np.array_equal(w.cpu().numpy().reshape(-1).astype(np.float64), wo.reshape(-1))
    T64, _ = p.forward(_dev(C.astype(np.float64)))
    # 1e-12 relative to the column scale (xyz columns carry a DC of ~1e5 next to entries near 0)
    colmax = np.abs(To).max(axis=0, keepdims=True)
    assert np.all(np.abs(T64.cpu().numpy() - To) <= 1e-12 * np.maximum(colmax, 1.0) + 1e-12 * np.abs(To))
    p.set_engine("level")
    Tl, _ = p.forward(_dev(C))
    _check_f32(Tl.cpu().numpy(), To, "seeded-level")
    st = p.stage_stats(4, D)
    assert st["valid"] and st["rows_per_stage"][0] == V.shape[0]


# ------------------------------------------------------- full-size, size-independent properties (cfg3)
def test_full_size_properties_cfg3(rt):
    """3M-Gaussian, 59-channel scene: round trip, Parseval, DC coefficient, linearity."""
    import torch
    from raht_3dgs_codec_amd import synth
    n, J, D, seed = synth.CONFIGS["cfg3"]
    V, keys, C = synth.scene(n, J, D, seed)
    N = V.shape[0]
    p = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 3 * J)
    Cd = _dev(C)
    T, w = p.forward(Cd)
    Crec = p.inverse(T)
    cmax = Cd.abs().max().item()
    assert (Crec - Cd).abs().max().item() <= 1e-5 * cmax                      # encode_3dgs.py:186-195
    e_in = (Cd.double() ** 2).sum(dim=0)
    e_out = (T.double() ** 2).sum(dim=0)
    assert torch.allclose(e_in, e_out, rtol=1e-5)                             # energy, encode_3dgs.py:183-184
    dc = Cd.double().sum(dim=0) / np.sqrt(N)
    assert torch.allclose(T[0].double(), dc, rtol=1e-4, atol=1e-4 * dc.abs().max().item())   # utils.py:46-57
    assert w[0].item() == float(N)
    X = torch.roll(Cd, 1, dims=0)
    Tx, _ = p.forward(X)
    Tsum, _ = p.forward(Cd + 2 * X)
    scale = T.abs().max().item()
    assert (Tsum - (T + 2 * Tx)).abs().max().item() <= 2e-5 * scale            # linearity
    order = p.order_RAGFT
    assert torch.equal(torch.sort(order)[0], torch.arange(N, device="cuda"))
    Q = p.quant_reorder(T, 0.01)
    Td = p.dequant_unreorder(Q, 0.01)
    assert (Td - T).abs().max().item() <= 0.005 * 1.0001 + 1e-6 * scale
    assert torch.equal(p.forward_quant(Cd, 0.01), Q)                          # fused == two-call
    assert torch.equal(p.dequant_inverse(Q, 0.01), p.inverse(Td))
    st = p.stage_stats(4, D)
    assert st["valid"] and len(st["rows_per_stage"]) <= 6


# ------------------------------------------------------------- truncated trees, roots, sharded scenes
@pytest.mark.parametrize("engine,tile_rows,tail_rows,tail_ch,final_rows", [("tile", 0, 0, 0, 0), ("tile", 64, 64, 0, 64), ("tile", 64, 128, 4, 64), ("level", 0, 0, 0, 0)])
def test_truncated_plan_roots_and_weights(rt, engine, tile_rows, tail_rows, tail_ch, final_rows):
    """top_level cut + compact root buffer + weighted leaves, against the numpy formulation."""
    import torch
    from raht_3dgs_codec_amd import synth
    from tests.numpy_ops import NumpyPlan
    V, keys, C = synth.scene(40000, 8, 7, seed=9)
    nbits, top = 24, 15
    kd = _dev(keys.view(np.int64))
    ref = NumpyPlan(torch.from_numpy(keys.view(np.int64).copy()), nbits, top_level=top)
    p = rt.RahtPlan.from_keys(kd, nbits, top_level=top)
    p.set_engine(engine, tile_rows, tail_rows, tail_ch, final_rows)
    assert np.array_equal(p.root_rows.cpu().numpy(), ref.root_rows.numpy()) and p.n_roots > 100
    C64 = torch.from_numpy(C.astype(np.float64))
    r_ref = torch.empty((ref.n_roots, 7), dtype=torch.float64)
    T_ref = ref.forward(C64, roots=r_ref)
    roots = torch.empty((p.n_roots, 7), dtype=torch.float64, device="cuda")
    T = p.forward(C64.cuda(), want_w=False, roots=roots)
    scale = float(T_ref.abs().max())
    np.testing.assert_allclose(T.cpu().numpy(), T_ref.numpy(), rtol=1e-12, atol=1e-12 * scale)
    np.testing.assert_allclose(roots.cpu().numpy(), r_ref.numpy(), rtol=1e-12, atol=1e-12 * scale)
    # inverse reads the roots from the buffer, not from T
    T2 = T.clone()
    T2[p.root_rows] = 123.0
    np.testing.assert_allclose(p.inverse(T2, roots=roots).cpu().numpy(), C.astype(np.float64), rtol=1e-11, atol=1e-11 * scale)
    # weighted leaves (what the top stage of a sharded scene uses)
    rng = np.random.default_rng(3)
    tk = np.unique(rng.integers(0, 512, size=300)).astype(np.int64)
    tw = rng.integers(1, 100000, size=tk.shape[0]).astype(np.int64)
    X = rng.normal(size=(tk.shape[0], 5))
    wp = rt.RahtPlan.from_keys(_dev(tk), 9, leaf_weights=_dev(tw))
    wp.set_engine(engine, tile_rows, tail_rows, tail_ch, final_rows)
    wr = NumpyPlan(torch.from_numpy(tk), 9, leaf_weights=torch.from_numpy(tw))
    Tw, w = wp.forward(_dev(X))
    np.testing.assert_allclose(Tw.cpu().numpy(), wr.forward(torch.from_numpy(X)).numpy(), rtol=1e-12, atol=1e-12)
    assert w[0].item() == float(tw.sum())
    np.testing.assert_allclose(wp.inverse(Tw).cpu().numpy(), X, rtol=1e-11, atol=1e-11)


def test_sharded_driver_single_rank_equals_plain_plan(rt):
    import torch
    from raht_3dgs_codec_amd import sharded, synth
    V, keys, C = synth.scene(120000, 11, 59, seed=21)
    kd = _dev(keys.view(np.int64))
    Cd = _dev(C)
    sh = sharded.ShardedRaht(kd, 33, prefix_bits=9)
    p = rt.RahtPlan.from_keys(kd, 33)
    T0, _ = p.forward(Cd)
    T1 = sh.forward(Cd)
    assert sh.n_roots > 50 and sh.total_rows == keys.shape[0]
    scale = T0.abs().max(dim=0)[0]
    assert bool(((T1 - T0).abs().max(dim=0)[0] <= 2e-6 * scale).all())
    assert (sh.inverse(T1) - Cd).abs().max().item() <= 1e-5 * Cd.abs().max().item()
    Q0, Q1 = p.forward_quant(Cd, 0.01), sh.forward_quant(Cd, 0.01)
    assert (Q0 != Q1).float().mean().item() < 1e-5 and (Q0 - Q1).abs().max().item() <= 1
    R1 = sh.dequant_inverse(Q1, 0.01)
    assert (R1 - p.dequant_inverse(Q0, 0.01)).abs().max().item() <= 0.02


def test_two_prefix_shards_stitched_by_the_top_stage(rt, oracle):
    """Emulates two ranks on one GPU: shard-local truncated transforms + one weighted top tree ==
    the oracle's transform of the whole scene."""
    import torch
    from raht_3dgs_codec_amd import synth
    J, D = 9, 11
    V, keys, C = synth.scene(90000, J, D, seed=31)
    nbits, pb = 3 * J, 9
    po = oracle.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
    To, _ = oracle.raht_fwd(C.astype(np.float64), po)
    pref = (keys >> np.uint64(nbits - pb)).astype(np.int64)
    cut = int(np.searchsorted(pref, 200))
    shards = [(0, cut), (cut, keys.shape[0])]
    plans, roots, metas = [], [], []
    for a, b in shards:
        pl = rt.RahtPlan.from_keys(_dev(keys[a:b].view(np.int64)), nbits, top_level=nbits - pb)
        rb = torch.empty((pl.n_roots, D), dtype=torch.float32, device="cuda")
        Tl = pl.forward(_dev(C[a:b]), want_w=False, roots=rb)
        rr = pl.root_rows.cpu().numpy()
        cnt = np.diff(np.concatenate([rr, [b - a]]))
        plans.append((pl, Tl, rr)); roots.append(rb); metas.append((pref[a:b][rr], cnt))
    tp = np.concatenate([m[0] for m in metas]); tc = np.concatenate([m[1] for m in metas])
    top = rt.RahtPlan.from_keys(_dev(tp.astype(np.int64)), pb, leaf_weights=_dev(tc.astype(np.int64)))
    Ttop = top.forward(torch.cat(roots), want_w=False)
    off = 0
    for (pl, Tl, rr), (a, b) in zip(plans, shards):
        Tl[pl.root_rows] = Ttop[off: off + len(rr)]
        off += len(rr)
        # tolerance relative to the WHOLE scene's column scale (the DC lives in one shard only)
        err = np.abs(Tl.cpu().numpy().astype(np.float64) - To[a:b]).max(axis=0)
        assert np.all(err <= 2e-6 * np.abs(To).max(axis=0)), float((err / np.abs(To).max(axis=0)).max())


def test_transform_entry_points_are_graph_capturable(rt):
    """After raht_plan_prepare the transform entry points only enqueue kernels on the given stream:
    they can be captured in a hipGraph (torch.cuda.CUDAGraph) and replayed."""
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, C = synth.scene(50000, 10, 59, seed=8)
    p = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 30)
    p.prepare(59)
    Cd = _dev(C)
    C0 = Cd.clone()
    ref_Q = p.forward_quant(Cd, 0.02)
    ref_R = p.dequant_inverse(ref_Q, 0.02)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                     # warm-up on the side stream, as torch recommends
        p.dequant_inverse(p.forward_quant(Cd, 0.02), 0.02)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        Qg = p.forward_quant(Cd, 0.02)
        Rg = p.dequant_inverse(Qg, 0.02)
    Cd.add_(1.0)                                   # new input, same buffers
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(Qg, p.forward_quant(Cd, 0.02)) and not torch.equal(Qg, ref_Q)
    assert torch.equal(Rg, p.dequant_inverse(Qg, 0.02))
    Cd.copy_(C0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(Qg, ref_Q) and torch.equal(Rg, ref_R)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RAHT_SOAK_SEEDS", "40"))))
def test_randomised_configurations(rt, seed):
    """Random N / key width / channel count / dtype / truncation / tile geometry against the numpy
    formulation of the list-free transform (tests/numpy_ops.py)."""
    import torch
    from tests.numpy_ops import NumpyPlan
    rng = np.random.default_rng(5000 + seed)
    nbits = int(rng.integers(1, 64))
    nmax = min(20000, 1 << min(nbits, 20))
    N = int(rng.integers(1, nmax + 1))
    D = int(rng.choice([1, 2, 3, 5, 16, 31, 32, 33, 59, 63, 64, 65, 100, 128, 130]))
    f64 = bool(rng.integers(0, 2))
    hi = (1 << nbits) - 1
    if nbits <= 20:
        keys = np.sort(rng.choice(1 << nbits, size=N, replace=False)).astype(np.int64)
    else:
        keys = np.unique(rng.integers(0, hi, size=N + 64, dtype=np.int64, endpoint=True))[:N]
        N = keys.shape[0]
    top = int(rng.integers(1, nbits + 1)) if rng.random() < 0.4 else None
    kt = torch.from_numpy(keys.copy())
    ref = NumpyPlan(kt, nbits, top_level=top)
    p = rt.RahtPlan.from_keys(kt.cuda(), nbits, top_level=top)
    geo = [(0, 0, 0, 0), (64, 64, 0, 64), (128, 64, 8, 64), (64, 128, 5, 128), (192, 0, 0, 0)][int(rng.integers(0, 5))]
    p.set_engine("tile" if rng.random() < 0.85 else "level", *geo)
    assert np.array_equal(p.root_rows.cpu().numpy(), ref.root_rows.numpy())
    C = rng.normal(size=(N, D)) * 10 ** rng.uniform(-2, 3)
    Cd = torch.from_numpy(C if f64 else C.astype(np.float32)).cuda()
    roots = torch.empty((p.n_roots, D), dtype=Cd.dtype, device="cuda")
    T = p.forward(Cd, want_w=False, roots=roots)
    r_ref = torch.empty((ref.n_roots, D), dtype=torch.float64)
    T_ref = ref.forward(torch.from_numpy(Cd.cpu().numpy().astype(np.float64)), roots=r_ref).numpy()
    scale = np.maximum(np.abs(T_ref).max(axis=0), 1e-300)
    # float32 bar, derived: a coefficient passes through <= nbits butterflies, each rounding it by <= 2.5 eps32 (a, b rounded
    # once, two products, one sum) of the running magnitude: worst case 2.5 * 6e-8 * 63 = 9.4e-6 of the column maximum, a
    # random walk ~sqrt(63) * 1.5e-7 = 1.2e-6. SURVEY 8c's 2e-6 is the bar for 3DGS-like scenes (tests/test_gpu_fullsize.py:
    # measured 1.7e-7); these adversarial draws (weighted leaves, truncated trees, 3-row tiles) get 3e-6, inside the worst case.
    tol = 1e-12 if f64 else 3e-6
    err = np.abs(T.cpu().numpy().astype(np.float64) - T_ref).max(axis=0) / scale
    assert err.max() <= tol, (seed, N, D, nbits, top, geo, float(err.max()))
    assert (np.abs(roots.cpu().numpy().astype(np.float64) - r_ref.numpy()).max(axis=0) / scale).max() <= tol
    R = p.inverse(T, roots=roots)
    assert (R - Cd).abs().max().item() <= (1e-11 if f64 else 2e-5) * max(Cd.abs().max().item(), 1e-30)
    if not f64 and D <= 256:
        steps = float(10 ** rng.uniform(-3, 0)) * float(np.abs(T_ref).max() + 1e-3) / 1000.0
        Q1 = p.forward_quant(Cd, steps)
        Q2 = p.quant_reorder(p.forward(Cd, want_w=False), steps)
        assert torch.equal(Q1, Q2)
        assert torch.equal(p.dequant_inverse(Q1, steps), p.inverse(p.dequant_unreorder(Q1, steps)))
    if f64 and D <= 256:
        steps = float(10 ** rng.uniform(-3, 0)) * float(np.abs(T_ref).max() + 1e-3) / 1000.0
        Q1 = p.forward_quant(Cd, steps)                                        # float64 quantizer fused into the float64 kernels
        Q2 = p.quant_reorder(p.forward(Cd, want_w=False), steps)
        assert Q1.dtype == torch.int32 and torch.equal(Q1, Q2)
        assert torch.equal(p.dequant_inverse(Q1, steps, dtype=torch.float64), p.inverse(p.dequant_unreorder(Q1, steps, dtype=torch.float64)))
        # ... and with the roots travelling through a caller buffer (truncated trees: what a sharded scene does)
        r1 = torch.empty((p.n_roots, D), dtype=torch.float64, device="cuda")
        Q3 = p.forward_quant(Cd, steps, roots=r1)
        assert torch.equal(r1, roots)                                           # the same float64 low-pass rows as the plain forward
        R3 = p.dequant_inverse(Q3, steps, roots=r1, dtype=torch.float64)
        # orthonormal inverse of coefficients that are off by at most half a step each (the roots not at all)
        assert (R3 - Cd).abs().max().item() <= 1e-11 * max(Cd.abs().max().item(), 1e-30) + 0.5 * steps * np.sqrt(N) * 1.001


def test_fused_quantization_divides_exactly(rt):
    """The fused forward replaces x / step by a hoisted-reciprocal refinement (transform.hip, P5); it
    must round like the IEEE division of encode_3dgs.py:204 for every coefficient. Keys that are all
    even with top_level = 1 leave no butterfly below the truncation level, so T == C and the quantizer
    sees exactly the adversarial values placed in C (ties, neighbours of ties, tiny, huge, signed zero)."""
    import torch
    N, D = 40000, 59
    rng = np.random.default_rng(77)
    keys = torch.arange(N, dtype=torch.int64, device="cuda") * 2
    p = rt.RahtPlan.from_keys(keys, 24, top_level=1)
    assert p.n_roots == N
    steps = torch.from_numpy(np.exp(rng.uniform(np.log(1e-6), np.log(1e3), size=D)).astype(np.float32)).cuda()
    steps[0], steps[1], steps[2], steps[3] = 0.01, 1.0, 3.0, 2.0 ** -20
    k = torch.from_numpy(rng.integers(-2 ** 20, 2 ** 20, size=(N, D))).to(torch.float32).cuda()
    base = (k + 0.5) * steps                                  # at / next to rounding ties
    C = base.clone()
    sel = torch.from_numpy(rng.integers(0, 6, size=(N, D))).cuda()
    C = torch.where(sel == 1, torch.nextafter(base, torch.full_like(base, float("inf"))), C)
    C = torch.where(sel == 2, torch.nextafter(base, torch.full_like(base, float("-inf"))), C)
    C = torch.where(sel == 3, torch.from_numpy(rng.normal(size=(N, D)).astype(np.float32)).cuda() * steps * 1000, C)
    C = torch.where(sel == 4, torch.from_numpy((rng.normal(size=(N, D)) * 1e-38).astype(np.float32)).cuda(), C)
    C = torch.where(sel == 5, k * steps, C)
    C[0, :] = 0.0
    C[1, :] = -0.0
    C[2, :] = 1e-45
    C = C.contiguous()
    want = torch.floor(C / steps + 0.5).clamp(-2.0 ** 31, 2.0 ** 31 - 1).to(torch.int32)[p.order_RAGFT]
    for st in (steps, 0.01, 0.3, 2.0 ** -7):
        if not torch.is_tensor(st):
            # a TENSOR divisor: torch turns division by a Python scalar into a multiplication by 1 / st
            sv = torch.full((D,), st, dtype=torch.float32, device="cuda")
            want_s = torch.floor(C / sv + 0.5).clamp(-2.0 ** 31, 2.0 ** 31 - 1).to(torch.int32)[p.order_RAGFT]
        else:
            want_s = want
        Q = p.forward_quant(C, st)
        assert torch.equal(Q, p.quant_reorder(C, st))
        assert torch.equal(Q, want_s)
    # steps outside the fast divider's range take the plain division path
    tiny = torch.full((D,), 1e-38, dtype=torch.float32, device="cuda")
    Cs = (C * 1e-36).contiguous()
    assert torch.equal(p.forward_quant(Cs, tiny), p.quant_reorder(Cs, tiny))


@pytest.mark.parametrize("N,D,top_rows", [(6000, 59, 8192), (6000, 59, 4096), (6000, 14, 1), (8192, 7, 8192), (8193, 64, 8192),
                                          (30000, 59, 8192), (30000, 59, 300), (1, 5, 0), (2, 4, 0), (65, 100, 0)])
def test_top_stage_thresholds(rt, N, D, top_rows):
    """The single-launch top stage (top_kernel) at every take-over point, against the level engine:
    whole tree in one launch (N <= top_rows), behind one / several tile stages, and switched off in
    favour of tile stages down to the roots (top_rows = 1)."""
    import torch
    rng = np.random.default_rng(N * 7 + D)
    nbits = 27
    keys = np.sort(rng.choice(1 << nbits, size=N, replace=False)).astype(np.int64) if N > 2 else np.arange(N, dtype=np.int64) * 5
    kd = torch.from_numpy(keys).cuda()
    ref = rt.RahtPlan.from_keys(kd, nbits)
    ref.set_engine("level")
    p = rt.RahtPlan.from_keys(kd, nbits)
    p.set_engine("tile", 0, 0, 0, top_rows)
    st = p.stage_stats(4, D)
    assert st["valid"]
    # float32: tile engine against LEVEL engine, both float32 -- each within 2e-6 of the float64 transform, so
    # within 4e-6 of each other by the triangle inequality; 3e-6 asserted
    for dt, tol in ((torch.float32, 3e-6), (torch.float64, 1e-12)):
        C = torch.from_numpy(rng.normal(size=(N, D))).to(dt).cuda()
        T, w = p.forward(C)
        Tr, wr = ref.forward(C)
        scale = Tr.abs().amax(dim=0).clamp_min(1e-30)
        assert ((T - Tr).abs().amax(dim=0) / scale).max().item() <= tol
        assert torch.equal(w, wr)
        R = p.inverse(T)
        assert (R - C).abs().max().item() <= (2e-5 if dt == torch.float32 else 1e-11) * C.abs().max().item()
        if dt == torch.float32:
            Q = p.forward_quant(C, 0.05)
            assert torch.equal(Q, p.quant_reorder(T, 0.05))
            assert torch.equal(p.dequant_inverse(Q, 0.05), p.inverse(p.dequant_unreorder(Q, 0.05)))


def test_plan_memory_cache_survives_churn(rt):
    """Plans are built and dropped per frame: their device blocks are recycled through the library's
    cache (raht_release_cached_memory empties it) and every new plan still computes the same thing."""
    import torch
    from raht_3dgs_codec_amd import _lib
    rng = np.random.default_rng(5)
    first = None
    for it in range(6):
        N = 50000 + 37 * (it % 3)
        keys = torch.from_numpy(np.sort(rng.choice(1 << 30, size=N, replace=False)).astype(np.int64)).cuda()
        C = torch.from_numpy(rng.normal(size=(N, 16)).astype(np.float32)).cuda()
        p = rt.RahtPlan.from_keys(keys, 30)
        T, _ = p.forward(C)
        assert (p.inverse(T) - C).abs().max().item() <= 2e-5 * C.abs().max().item()
        p.set_engine("level")
        Tl, _ = p.forward(C)
        assert (T - Tl).abs().max().item() <= 3e-6 * Tl.abs().max().item()      # two float32 engines: <= 2 x 2e-6 (triangle inequality)
        del p
        if it == 2:
            _lib.check(_lib.lib().raht_release_cached_memory())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fused_quantization_wide_step_range(rt, seed):
    """The hoisted-reciprocal divider over most of its admitted step range ([2^-100, 2^100]; here 2^+-80) and
    quotients from denormal to far beyond int32: every integer must equal floor(IEEE x / step + 0.5)
    (saturated like v_cvt_i32_f32)."""
    import torch
    N, D = 30000, 64
    rng = np.random.default_rng(900 + seed)
    keys = torch.arange(N, dtype=torch.int64, device="cuda") * 2
    p = rt.RahtPlan.from_keys(keys, 24, top_level=1)            # no butterflies: T == C
    steps = torch.from_numpy(np.exp2(rng.uniform(-80, 80, size=D)).astype(np.float32)).cuda()
    mant = torch.from_numpy(rng.uniform(0.5, 1.0, size=(N, D)).astype(np.float32)).cuda()
    expo = torch.from_numpy(rng.integers(-40, 36, size=(N, D)).astype(np.float32)).cuda()
    sign = torch.from_numpy(rng.choice([-1.0, 1.0], size=(N, D)).astype(np.float32)).cuda()
    C = (sign * mant * torch.exp2(expo) * steps).contiguous()      # quotients 2^-41 .. 2^36, finite everywhere
    half = (torch.floor(mant * 64) + 0.5) * steps                   # exact-looking ties k + 0.5
    C = torch.where(torch.from_numpy(rng.random((N, D)) < 0.2).cuda(), sign * half, C).contiguous()
    assert torch.isfinite(C).all()
    want = torch.floor(C / steps + 0.5).clamp(-2.0 ** 31, 2.0 ** 31 - 1).to(torch.int32)[p.order_RAGFT]
    Q = p.forward_quant(C, steps)
    assert torch.equal(Q, want)
    assert torch.equal(Q, p.quant_reorder(C, steps))


def test_stage0_events_bracket_the_dominant_kernel(rt):
    """raht_plan_set_stage0_events: the caller's HIP events are recorded around the stage-0 launch of a
    real transform (bench.py times the dominant kernel inside its steps with them)."""
    import ctypes as C
    import torch
    from raht_3dgs_codec_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    vp = C.c_void_p
    hip.hipEventCreate.argtypes = [C.POINTER(vp)]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
    hip.hipEventDestroy.argtypes = [vp]
    a, b = vp(), vp()
    assert hip.hipEventCreate(C.byref(a)) == 0 and hip.hipEventCreate(C.byref(b)) == 0
    N, D = 300000, 59
    keys = torch.arange(N, dtype=torch.int64, device="cuda") * 3 + 1
    p = rt.RahtPlan.from_keys(keys, 24)
    Cm = torch.randn(N, D, device="cuda")
    ref = p.forward_quant(Cm, 0.1)
    L = _lib.lib()
    _lib.check(L.raht_plan_set_stage0_events(p._h, a, b))
    Q = p.forward_quant(Cm, 0.1)
    _lib.check(L.raht_plan_set_stage0_events(p._h, None, None))
    torch.cuda.synchronize()
    ms = C.c_float()
    assert hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
    assert 0.0 < ms.value < 5.0                       # a 71 MB launch: tens of microseconds
    assert torch.equal(Q, ref)
    assert L.raht_plan_set_stage0_events(p._h, a, None) != 0        # both or none
    hip.hipEventDestroy(a); hip.hipEventDestroy(b)


# ---------------------------------------------------------------- node extents, straight from their definition
def _extents_by_definition(keys):
    """lvl / wl / wr of include/raht.h from sorted Python-int keys (bisect on the prefix ranges)."""
    import bisect
    n = len(keys)
    lvl, wl, wr = [255] + [0] * (n - 1), [0] * n, [0] * n
    for i in range(1, n):
        l = (keys[i] ^ keys[i - 1]).bit_length() - 1
        lvl[i] = l
        end = bisect.bisect_left(keys, ((keys[i] >> l) + 1) << l)              # first row past the node starting at i
        start = bisect.bisect_left(keys, (keys[i - 1] >> l) << l)              # first row of the node ending at i - 1
        wr[i], wl[i] = end - i, i - start
    return lvl, wl, wr


@pytest.mark.parametrize("n", [2, 63, 64, 65, 127, 129, 1023, 1024, 1025, 2049, 5000])
@pytest.mark.parametrize("pattern", ["uniform60", "clustered", "staircase"])
def test_node_extents_match_their_definition(rt, n, pattern):
    """Sizes around the wave (64 rows) and workgroup (1024 rows) boundaries of the extent kernel, keys whose
    nodes span many of them, and level sequences that leave whole waves unresolved (staircase)."""
    import torch
    rng = np.random.default_rng(n * 7 + len(pattern))
    nbits = 60
    if pattern == "uniform60":
        ks = set(int(x) for x in rng.integers(0, 1 << 60, size=2 * n, dtype=np.uint64))
    elif pattern == "clustered":
        # a few deep clusters: long runs of rows differing only in low bits, joined at very high levels
        ks = set()
        centres = [int(x) for x in rng.integers(0, 1 << 60, size=5, dtype=np.uint64)]
        while len(ks) < n:
            c = centres[int(rng.integers(0, 5))]
            ks.add((c & ~((1 << 14) - 1)) | int(rng.integers(0, 1 << 14)))
    else:
        # lvl strictly decreasing, then increasing, over long stretches: nodes nested like a staircase
        ks, k = set(), 0
        for i in range(n):
            b = 59 - (i % 60) if (i // 60) % 2 == 0 else (i % 60)
            k += 1 << b
            k &= (1 << 60) - 1
            ks.add(k | (i & 1))
        ks |= set(int(x) for x in rng.integers(0, 1 << 60, size=n, dtype=np.uint64))
    keys = sorted(ks)[:n]
    assert len(keys) == n
    kd = torch.tensor(np.array(keys, dtype=np.uint64).view(np.int64), device="cuda")
    p = rt.RahtPlan.from_keys(kd, nbits)
    _, lvl, wl, wr = p.arrays()
    rl, rwl, rwr = _extents_by_definition(keys)
    assert np.array_equal(lvl, np.array(rl, dtype=np.uint8))
    assert np.array_equal(wr, np.array(rwr, dtype=np.int32))
    assert np.array_equal(wl, np.array(rwl, dtype=np.int32))


def test_very_wide_row_stride_takes_the_level_engine(rt):
    """Row strides above 2^18 elements do not fit the tile kernel's 32-bit in-tile offsets: the entry points
    fall back to the level engine and return the same coefficients (include/raht.h)."""
    import torch
    g = load_golden("n257_j3_d11")
    p = _plan(rt, g)
    N, D = g["C"].shape
    ld = (1 << 18) + 5
    big = torch.zeros((N, ld), dtype=torch.float32, device="cuda")
    big[:, :D] = _dev(g["C"])
    T0, _ = p.forward(_dev(g["C"]))
    T1, _ = p.forward(big[:, :D])
    _check_f32(T1.cpu().numpy(), g["T"], "wide stride")
    assert float((T1 - T0).abs().max()) <= 2e-6 * float(T0.abs().max())
    Q0 = p.forward_quant(_dev(g["C"]), 0.5)
    Q1 = p.forward_quant(big[:, :D], 0.5)
    assert int((Q0 != Q1).sum()) <= 2                      # two engines: a rounding tie may fall either way
    C1 = p.dequant_inverse(Q1, 0.5)
    assert float((C1 - _dev(g["C"])).abs().max()) < 40 * 0.5


# ------------------------------------------------------------------ several scenes in one set of launches
def _batch_scenes(R, sizes, D, seed0=50):
    import torch
    from raht_3dgs_codec_amd import synth
    plans, Cs = [], []
    for i, (n, J) in enumerate(sizes):
        V, keys, C = synth.scene(n, J, D, seed=seed0 + i)
        plans.append(R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 3 * J))
        Cs.append(torch.from_numpy(C).cuda())
    return plans, Cs


@pytest.mark.parametrize("D", [59, 14, 56])
@pytest.mark.parametrize("sizes", [
    [(40000, 10), (300, 6), (150000, 11), (5000, 9)],                       # 3 stages, one top stage, 4 stages, 2 stages
    [(20000 + 3000 * i, 10) for i in range(11)],                            # more scenes than one launch carries
    [(70000, 12)],                                                          # a batch of one
])
def test_batch_equals_single_scene_calls_bit_for_bit(rt, D, sizes):
    """raht_*_batch (stage k of every scene in ONE launch, top stages with blockIdx.y = scene) against n calls of the
    single-scene entry points: the same kernels on the same tiles, so every output is bit-identical."""
    import torch
    from raht_3dgs_codec_amd import ops
    plans, Cs = _batch_scenes(rt, sizes, D)
    step = 0.02
    Tb = ops.forward_batch(plans, Cs)
    Qb = ops.forward_quant_batch(plans, Cs, step)
    for p, C, T, Q in zip(plans, Cs, Tb, Qb):
        assert torch.equal(T, p.forward(C, want_w=False))
        assert torch.equal(Q, p.forward_quant(C, step))
    Cb = ops.inverse_batch(plans, Tb)
    Cq = ops.dequant_inverse_batch(plans, Qb, step)
    for p, C, T, Q, c1, c2 in zip(plans, Cs, Tb, Qb, Cb, Cq):
        assert torch.equal(c1, p.inverse(T))
        assert torch.equal(c2, p.dequant_inverse(Q, step))
        assert (c1 - C).abs().max().item() <= 1e-5 * C.abs().max().item()
    # per-channel steps travel with the batch as well
    steps = [0.01 * (1 + (c % 5)) for c in range(D)]
    Qs = ops.forward_quant_batch(plans, Cs, steps)
    for p, C, Q in zip(plans, Cs, Qs):
        assert torch.equal(Q, p.forward_quant(C, steps))


def test_batch_mixes_engines_and_geometries(rt):
    """Scenes outside the tile engine (level engine selected; D below one 16-byte chunk) and scenes with another tile
    geometry run inside the same call: the level-engine scenes through their own entry point, the others in launches of
    their own shape."""
    import torch
    from raht_3dgs_codec_amd import ops
    plans, Cs = _batch_scenes(rt, [(30000, 10), (30000, 10), (30000, 10), (30000, 10)], 14, seed0=70)
    plans[1].set_engine("level")
    plans[2].set_engine("tile", 64, 64, 0, 64)
    ref = [p.forward_quant(C, 0.05) for p, C in zip(plans, Cs)]
    out = ops.forward_quant_batch(plans, Cs, 0.05)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    back = ops.dequant_inverse_batch(plans, out, 0.05)
    for p, Q, c in zip(plans, out, back):
        assert torch.equal(c, p.dequant_inverse(Q, 0.05))
    # D = 3 float32: rows shorter than one chunk -> every scene on the level engine
    p3, C3 = _batch_scenes(rt, [(5000, 8), (900, 7)], 3, seed0=80)
    for T, p, C in zip(ops.forward_batch(p3, C3), p3, C3):
        assert torch.equal(T, p.forward(C, want_w=False))
    # errors: a plan twice in one batch, mismatched D
    with pytest.raises(rt.RahtError):
        ops.forward_batch([plans[0], plans[0]], [Cs[0], Cs[0]])
    with pytest.raises(ValueError):
        ops.forward_batch([plans[0], p3[0]], [Cs[0], C3[0]])


def test_voxelize_plan_feeds_the_plan_from_the_voxelizers_keys(rt, oracle):
    """ops.voxelize_plan: unsorted cloud -> PCvox + plan in one call; the plan built from the (borrowed) voxel keys is the
    plan RAHT_param_reorder_fast builds from the voxel coordinates (order_RAGFT, transform bit for bit)."""
    import torch
    rng = np.random.default_rng(77)
    n, d, J = 60000, 11, 9
    P = (rng.random((n, 3)) * 5.0 - 1.0).astype(np.float32)
    P[::6] = P[1::6][: P[::6].shape[0]]
    PC = np.concatenate([P, rng.standard_normal((n, d)).astype(np.float32)], axis=1)
    PCvox, plan, info = rt.voxelize_plan(torch.from_numpy(PC).cuda(), None, None, J)
    ref = oracle.voxelize(PC, J)
    assert plan.N == ref["Nvox"] and np.array_equal(PCvox.cpu().numpy(), ref["PCvox"])
    # the one-call entry point (raht_voxelize_plan) hands out what the two-call sequence would: voxel starts and voxel keys
    assert np.array_equal(info["voxel_indices"].cpu().numpy(), ref["voxel_indices"])
    assert np.array_equal(info["voxel_keys"].cpu().numpy().view(np.uint64), np.asarray(ref["keys_sorted"]).view(np.uint64)[ref["voxel_indices"]])
    assert info["Nvox"] == ref["Nvox"] and abs(info["voxel_size"] - float(ref["voxel_size"])) <= 1e-12 * float(ref["voxel_size"])
    V = PCvox[:, :3].double()
    ListC, _, _, order = rt.raht_fn["RAHT_param"](V, torch.zeros(3, dtype=torch.float64, device="cuda"), 2 ** J, J)
    assert torch.equal(order, plan.order_RAGFT)
    C = PCvox[:, 3:].contiguous()
    T0, _ = rt.raht_fn["RAHT"](C, ListC, None, None)
    assert torch.equal(plan.forward(C, want_w=False), T0)
    del info, PCvox            # the plan keeps the borrowed key tensor alive
    torch.cuda.empty_cache()
    assert torch.equal(plan.inverse(T0), rt.plan_of(ListC).inverse(T0))


# ---- the key sort in its three forms (csrc/scan_sort.hip) --------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"RAHT_SORT_TICKET": "1"}, {"RAHT_SORT_ONESWEEP": "0"}, {"RAHT_SORT_ROUNDS": "8"}, {"RAHT_SORT_ROUNDS": "12"},
                                 {"RAHT_SORT_DEBUG_FAIL_TILE": "0"}, {"RAHT_SORT_DEBUG_FAIL_TILE": "37"}],
                         ids=["one-sweep", "one-sweep, ticketed tiles", "pass by pass", "2048-item tiles", "3072-item tiles",
                              "tile 0 gives up -> fallback", "tile 37 gives up -> fallback"])
def test_key_sort_forms_equal_a_stable_sort(env):
    """raht_sort_keys == torch.sort(stable=True), keys AND permutation, for sizes around the tile edges (1 ... 3 000 017), every
    digit-pass count (1 ... 63 key bits) and inputs full of duplicates -- in the default one-sweep form (tiles numbered by
    workgroup index), with ticketed tiles, and in the pass-by-pass form the library falls back to; and with one tile of every
    one-sweep pass made to give up (RAHT_SORT_DEBUG_FAIL_TILE): the tiles after it pass the error on and write nothing, the grid
    drains, the host finds the error word and repeats the sort pass by pass -- same results, and the voxelizer (whose mean
    kernel is enqueued BEHIND the sort without a host round trip) must not gather through the stale indices. The knobs are read
    once per process: each form runs tools/check_sort.py in its own interpreter."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_sort.py")], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "correctness: 0 mismatches" in r.stdout
    # raht_sort_fallbacks(): 0 unless a tile was made to give up (then > 0: the fallback is observable, not silent)
    if "RAHT_SORT_DEBUG_FAIL_TILE" in env:
        assert "sort fallbacks: 0" not in r.stdout
    else:
        assert "sort fallbacks: 0" in r.stdout


# ---- raht_dequant_inv_sqdiff: the fused inverse that measures its own distortion (encode_3dgs.py:274,298-310) -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("tile_rows,tail_rows,final_rows", [(0, 0, 0), (64, 64, 64), (128, 64, 0), (256, 128, 256)])
@pytest.mark.parametrize("name", ["n1000_j10_d14", "n1500_j12_d56", "n2000_j10_d59", "n257_j3_d11", "n3000_j18_d3", "n8_cube_j1"])
def test_dequant_inverse_sqdiff_equals_the_two_passes(rt, name, tile_rows, tail_rows, final_rows):
    """C_rec bit-identical to raht_dequant_inv; the per-column sums equal raht_sqdiff_columns(C, C_rec) up to the order of the
    float64 additions -- with and without writing C_rec; one-launch trees and narrow rows take the two-pass path inside."""
    import ctypes as C
    import torch
    from raht_3dgs_codec_amd import _lib
    g = load_golden(name)
    p = _plan(rt, g, "tile", tile_rows, tail_rows, 0, final_rows)
    Cd = _dev(g["C"])
    N, D = Cd.shape
    for steps in (0.37, [0.05 + 0.01 * c for c in range(D)]):
        Q = p.forward_quant(Cd, steps)
        ref = p.dequant_inverse(Q, steps)
        want = torch.empty(D, dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib().raht_sqdiff_columns(C.c_void_p(Cd.data_ptr()), D, C.c_void_p(ref.data_ptr()), D, N, D, _lib.RAHT_F32,
                                                  C.c_void_p(want.data_ptr()), None))
        rec, ssd = p.dequant_inverse_sqdiff(Q, steps, Cd)
        assert torch.equal(rec, ref)
        assert torch.allclose(ssd, want, rtol=1e-12, atol=1e-300), (name, float((ssd - want).abs().max()))
        exact = ((ref.double() - Cd.double()) ** 2).sum(dim=0)          # (differences in float32 first, as the drivers' float32 frames do)
        f32d = ((ref - Cd).double() ** 2).sum(dim=0)
        assert torch.allclose(ssd, f32d, rtol=1e-9, atol=1e-300) and torch.allclose(ssd, exact, rtol=1e-3, atol=1e-12)
        none, ssd2 = p.dequant_inverse_sqdiff(Q, steps, Cd, want_rec=False)
        assert none is None and torch.equal(ssd2, ssd)                    # a deterministic sum


@pytest.mark.gpu
def test_dequant_inverse_sqdiff_strided_reference_and_big_scene(rt):
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, Ch = synth.scene(300000, 10, 56, seed=4)
    p = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 30)
    N, D = Ch.shape
    big = torch.zeros((N, 64), dtype=torch.float32, device="cuda")
    big[:, :D] = _dev(Ch)
    Q = p.forward_quant(big[:, :D], 0.02)
    rec, ssd = p.dequant_inverse_sqdiff(Q, 0.02, big[:, :D])
    assert torch.equal(rec, p.dequant_inverse(Q, 0.02))
    f32d = ((rec - big[:, :D]).double() ** 2).sum(dim=0)
    assert torch.allclose(ssd, f32d, rtol=1e-9)
    # orthonormal transform: the distortion equals the quantization error of the coefficients (Parseval), ~ N step^2 / 12 per column
    assert 0.5 < float(ssd.mean()) / (N * 0.02 ** 2 / 12) < 1.5


# ---- one workspace set per direction: forward of step s + 1 next to the inverse of step s (encode_3dgs.py:199-275) ----------
@pytest.mark.gpu
@pytest.mark.parametrize("mixed", [False, True])
def test_forward_and_inverse_of_one_plan_on_two_streams(rt, mixed):
    """raht_plan_set_concurrent_directions: a forward-direction and an inverse-direction call of the same plan in flight at the
    same time on two streams, many steps in a row -- every reconstruction bit-identical to the one-stream loop."""
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, Ch = synth.scene(400000, 10, 59, seed=8)
    p = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 30)
    Cs = [_dev(Ch) * (1.0 + 0.1 * i) for i in range(6)]
    steps = [0.01 * (i + 1) for i in range(6)]
    fq = (lambda c, s: p.forward_quant_mixed(c, s, 3)) if mixed else p.forward_quant
    di = (lambda q, s: p.dequant_inverse_mixed(q, s, 3)) if mixed else p.dequant_inverse
    want = [di(fq(c, s), s) for c, s in zip(Cs, steps)]
    torch.cuda.synchronize()
    p.set_concurrent_directions(True)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    got, qs, evs = [], [], []
    for rep in range(3):
        got.clear(); qs.clear(); evs.clear()
        for i in range(len(Cs) + 1):
            if i < len(Cs):
                with torch.cuda.stream(sa):
                    qs.append(fq(Cs[i], steps[i]))
                    e = torch.cuda.Event(); e.record(sa); evs.append(e)
            if i >= 1:
                with torch.cuda.stream(sb):
                    sb.wait_event(evs[i - 1])
                    got.append(di(qs[i - 1], steps[i - 1]))
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b), rep
    p.set_concurrent_directions(False)
    assert torch.equal(di(fq(Cs[0], steps[0]), steps[0]), want[0])


# ---- raht_fwd_quant_multi: one forward pass, one quantization per step (encode_3dgs.py:28,199-217) -------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tile_rows,tail_rows,final_rows", [(0, 0, 0), (64, 64, 64), (128, 64, 0)])
@pytest.mark.parametrize("name", ["n1000_j10_d14", "n1500_j12_d56", "n2000_j10_d59", "n257_j3_d11", "n3000_j18_d3", "n8_cube_j1", "t_n1"])
def test_forward_quant_multi_equals_single_step_calls(rt, name, tile_rows, tail_rows, final_rows):
    import torch
    g = load_golden(name)
    p = _plan(rt, g, "tile", tile_rows, tail_rows, 0, final_rows)
    C = _dev(g["C"])
    for steps in ([0.37], [1.0, 4.0, 8.0], [0.01 * s for s in (1, 4, 8, 12, 16, 20, 24, 32, 64)], [0.05 * (i + 1) for i in range(14)], [1e-38, 3e38, 1.0]):
        Qs = p.forward_quant_multi(C, steps)
        assert len(Qs) == len(steps)
        for q, s in zip(Qs, steps):
            assert torch.equal(q, p.forward_quant(C, s)), (name, s)


@pytest.mark.gpu
def test_forward_quant_multi_level_engine_and_errors(rt):
    import torch
    from raht_3dgs_codec_amd._lib import RahtError
    g = load_golden("n1000_j10_d14")
    p = _plan(rt, g, "level")
    C = _dev(g["C"])
    Qs = p.forward_quant_multi(C, [0.5, 2.0])
    assert torch.equal(Qs[0], p.forward_quant(C, 0.5)) and torch.equal(Qs[1], p.forward_quant(C, 2.0))
    with pytest.raises(RahtError):
        p.forward_quant_multi(C, [0.5, 0.0])
