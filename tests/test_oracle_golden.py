"""Pin the CPU oracle (oracle/raht_oracle.c) against golden vectors produced by the reference's
own Python (tests/golden/gen_golden.py).  CPU-only; this is what makes the oracle trustworthy as
the checker for the HIP path on the GPU box, where the reference itself cannot travel.

Bars: List/Flags/weights/order/Morton/voxel indices bit-exact
(crosscheck.py:200-270 compares lists exactly); float64 coefficients rtol = atol = 1e-12
(crosscheck.py:363-366)."""
import numpy as np
import pytest

from .conftest import golden_names, load_golden

TRANSFORM = golden_names(exclude_prefix="vox_")
VOX = golden_names(prefix="vox_")


def _param(orc, g, quirks=True):
    V = g["V"].astype(np.float64)
    J = int(g["J"])
    return orc.raht_param(V, np.zeros(3), 2 ** J, J, ref_quirks=quirks)


@pytest.mark.parametrize("name", TRANSFORM)
def test_param_lists_exact(oracle, name):
    g = load_golden(name)
    p = _param(oracle, g)
    assert p.nlevels == len(g["List"])
    for l in range(p.nlevels):
        assert np.array_equal(p.List[l], g["List"][l]), f"List[{l}]"
        assert np.array_equal(p.Flags[l], g["Flags"][l]), f"Flags[{l}]"
        assert np.array_equal(p.weights[l], g["weights"][l]), f"weights[{l}]"
    assert np.array_equal(p.morton, g["morton"])
    if bool(g["order_is_none"]):
        assert p.order is None
    else:
        assert np.array_equal(p.order, g["order"])


@pytest.mark.parametrize("name", TRANSFORM)
def test_order_without_quirks_is_permutation(oracle, name):
    g = load_golden(name)
    p = _param(oracle, g, quirks=False)
    N = g["V"].shape[0]
    assert p.order is not None and np.array_equal(np.sort(p.order), np.arange(N))
    if not bool(g["order_is_none"]) and np.array_equal(np.sort(g["order"]), np.arange(N)):
        assert np.array_equal(p.order, g["order"])     # identical wherever the reference is sane


@pytest.mark.parametrize("name", TRANSFORM)
def test_forward_inverse_fp64(oracle, name):
    g = load_golden(name)
    p = _param(oracle, g)
    C = g["C"].astype(np.float64)
    T, w = oracle.raht_fwd(C, p)
    np.testing.assert_allclose(T, g["T"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(w.reshape(-1), g["w"])
    Crec = oracle.raht_inv(T, p)
    np.testing.assert_allclose(Crec, C, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(C).max()))
    # reference invariants: energy preservation (encode_3dgs.py:183-184) and DC (utils.py:46-57)
    assert abs(np.linalg.norm(T) - np.linalg.norm(C)) <= 1e-10 * max(1.0, np.linalg.norm(C))
    np.testing.assert_allclose(T[0], C.sum(axis=0) / np.sqrt(C.shape[0]), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("name", [n for n in TRANSFORM if any(k.startswith("q_step") for k in load_golden(n))])
def test_quantize_reorder_roundtrip(oracle, name):
    g = load_golden(name)
    p = _param(oracle, g)
    T, _ = oracle.raht_fwd(g["C"].astype(np.float64), p)
    for key in [k for k in g if k.startswith("q_step")]:
        step = float(key[len("q_step"):])
        Q = oracle.quant_reorder(T, step, p.order)
        bad = np.argwhere(Q != g[key])
        # torch's vectorised sqrt/div differ from IEEE scalar math in the last ulp, so integers may
        # differ only where the reference's own pre-floor value sits on a rounding tie, and by 1.
        pre = g["T"][p.order] / step + 0.5
        for r, c in bad:
            assert abs(int(Q[r, c]) - int(g[key][r, c])) == 1
            assert abs(pre[r, c] - np.round(pre[r, c])) < 1e-9, (key, r, c, pre[r, c])
        assert len(bad) <= 0.02 * Q.size
        Q = g[key]
        Tdec = oracle.dequant_unreorder(Q, step, p.order)
        Crec = oracle.raht_inv(Tdec, p)
        np.testing.assert_allclose(Crec, g["crec_step" + key[len("q_step"):]], rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("name", VOX)
def test_voxelize(oracle, name):
    g = load_golden(name)
    vmin = None if g["vmin_in"].size == 0 else g["vmin_in"]
    width = None if float(g["width_in"]) < 0 else float(g["width_in"])
    J = int(g["J"])
    r = oracle.voxelize(g["PC"], J, vmin=vmin, width=width)
    assert np.array_equal(oracle.morton(g["Vint"].astype(np.int64), J), g["morton"])
    assert np.array_equal(r["keys_sorted"], g["keys_sorted"])
    assert r["Nvox"] == int(g["Nvox"])
    assert np.array_equal(r["voxel_indices"], g["voxel_indices"])
    assert np.array_equal(r["vmin"], g["vmin"])
    assert r["width"] == float(g["width"]) and r["voxel_size"] == float(g["voxel_size"])
    assert np.array_equal(r["PCvox"][:, :3], g["PCvox"][:, :3])            # integer voxel coords
    # torch.sort is unstable (voxelize_pc.py:101): permutations agree up to order inside a voxel
    assert np.array_equal(g["morton"][r["sort_idx"]], g["keys_sorted"])
    np.testing.assert_allclose(r["PCvox"][:, 3:], g["PCvox"][:, 3:], rtol=2e-6, atol=1e-6)


def test_docs_worked_example(oracle):
    """docs/voxelization.md:19-95 -- sort_idx, voxel_indices and PCvox of the 8-point example."""
    g = load_golden("vox_docs8_given")
    r = oracle.voxelize(g["PC"], 2, vmin=[0, 0, 0], width=1.0)
    assert r["sort_idx"].tolist() == [0, 2, 7, 5, 4, 3, 1, 6]
    assert r["voxel_indices"].tolist() == [0, 3, 4, 5, 6]
    assert r["Vvox"].tolist() == [[0, 0, 0], [1, 2, 0], [2, 1, 2], [2, 2, 2], [3, 3, 3]]
    assert r["keys_sorted"].tolist() == [0, 0, 0, 20, 42, 56, 63, 63]     # code-true Morton keys


def _by_point(rows, sort_idx):
    out = np.empty_like(rows)
    out[sort_idx] = rows
    return out


def check_residuals_against_reference(g, PCsorted, DeltaPC, sort_idx, voxel_indices):
    """The reference sorts with an UNSTABLE torch.sort (voxelize_pc.py:101): rows inside a voxel come in another order, so
    outputs are compared per point. Position residuals: bit-exact. Attribute residuals subtract the voxel mean, whose
    float32 sum depends on the order of the voxel's points once there are three or more: exact for voxels of <= 2 points,
    1e-6 of the attribute scale otherwise."""
    PC = g["PC"]
    N, ld = PC.shape
    np.testing.assert_array_equal(_by_point(PCsorted, sort_idx), PC)
    np.testing.assert_array_equal(_by_point(g["PCsorted"], g["sort_idx"]), PC)
    mine, ref = _by_point(DeltaPC, sort_idx), _by_point(g["DeltaPC"], g["sort_idx"])
    np.testing.assert_array_equal(mine[:, :3], ref[:, :3])
    if ld > 3:
        counts = np.diff(np.concatenate([voxel_indices, [N]]))
        per_sorted = np.repeat(counts, counts)                       # voxel population of every sorted point
        small = _by_point(per_sorted.reshape(-1, 1), sort_idx).reshape(-1) <= 2
        np.testing.assert_array_equal(mine[small, 3:], ref[small, 3:])
        np.testing.assert_allclose(mine[~small, 3:], ref[~small, 3:], rtol=0, atol=1e-6 * max(1.0, float(np.abs(PC[:, 3:]).max())))


@pytest.mark.parametrize("name", golden_names(prefix="voxres_"))
def test_voxelizer_residuals_match_reference(oracle, name):
    g = load_golden(name)
    vmin = None if g["vmin_in"].size == 0 else g["vmin_in"]
    width = None if float(g["width_in"]) < 0 else float(g["width_in"])
    r = oracle.voxelize(g["PC"], int(g["J"]), vmin=vmin, width=width)
    assert np.array_equal(r["voxel_indices"], g["voxel_indices"]) and r["voxel_size"] == float(g["voxel_size"])
    np.testing.assert_array_equal(r["vmin"], g["vmin"])
    pcs, dl = oracle.voxel_residuals(g["PC"], r)
    check_residuals_against_reference(g, pcs, dl, r["sort_idx"], r["voxel_indices"])
