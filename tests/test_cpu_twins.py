"""Host twins of the C ABI (last row): oracle/_build/libraht_cpu.so exports raht_cpu_* with the
parameter lists of their namesakes in include/raht.h, so a host can swap the CPU restatement and the MI355X
path by symbol name. Checked here: (1) the prototypes, textually, against include/raht.h; (2) the twins'
results against the golden vectors from the reference; (3) on the GPU box, the same ctypes call sequence
through both libraries. The twins are test infrastructure: the product package never loads them."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from .conftest import golden_names, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _protos(path, prefix):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int64_t|int|const char \*)\s*(" + prefix + r"\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        out[m.group(2)] = (m.group(1).strip(), re.sub(r"\s+", " ", m.group(3)).strip())
    return out


@pytest.fixture(scope="module")
def twins(oracle):
    from oracle import oracle as orc
    L = C.CDLL(orc.CPU_TWINS_SO)
    L.raht_cpu_last_error.restype = C.c_char_p
    L.raht_cpu_plan_size.restype = C.c_int64
    return L


def test_twin_prototypes_equal_the_product_prototypes(twins):
    prod = _protos(os.path.join(ROOT, "include", "raht.h"), "raht_")
    twin = _protos(os.path.join(ROOT, "oracle", "raht_cpu.h"), "raht_cpu_")
    assert len(twin) >= 25
    for name, (ret, args) in twin.items():
        pname = name.replace("raht_cpu_", "raht_")
        assert pname in prod, f"{name}: no product entry {pname}"
        pret, pargs = prod[pname]
        assert ret == pret, name
        assert args.replace("raht_cpu_plan", "raht_plan") == pargs, f"{name}:\n  twin    {args}\n  product {pargs}"
        assert hasattr(twins, name), f"{name} not exported"
    # the path's core entry points all have a twin
    for core in ("raht_plan_create", "raht_plan_create_from_keys", "raht_plan_destroy", "raht_plan_levels", "raht_plan_export_level",
                 "raht_plan_order", "raht_fwd", "raht_fwd_f64", "raht_inv", "raht_inv_f64", "raht_fwd_quant", "raht_dequant_inv",
                 "raht_quant_reorder", "raht_dequant_unreorder", "raht_voxelize", "raht_morton", "raht_sort_keys",
                 "raht_plan_create_from_keys_borrowed", "raht_fwd_batch", "raht_inv_batch", "raht_fwd_quant_batch", "raht_dequant_inv_batch"):
        assert core.replace("raht_", "raht_cpu_", 1) in twin, core


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Api:
    """One ctypes call sequence, parameterised by the library and the symbol prefix: `raht_` + device pointers
    (the product) or `raht_cpu_` + host pointers (the twins)."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix

    def fn(self, name):
        f = getattr(self.lib, self.prefix + name)
        return f

    def err(self):
        e = getattr(self.lib, self.prefix + "last_error")
        e.restype = C.c_char_p
        return e().decode()


def _run_path(api, ptr, V, Cm, J, step, alloc, fetch):
    """plan -> levels / lists / order -> fwd (f32, f64) -> quantize (fused) -> dequantize + inverse. `ptr` turns a
    buffer into a pointer argument, `alloc(shape, dtype)` makes an output buffer, `fetch` brings it to numpy."""
    N, D = Cm.shape
    h = C.c_void_p()
    minV = (C.c_double * 3)(0.0, 0.0, 0.0)
    Vb = alloc(V.shape, np.float64, V)
    rc = api.fn("plan_create")(ptr(Vb), 1, C.c_int64(N), minV, C.c_double(2.0 ** J), J, None, C.byref(h))
    assert rc == 0, api.err()
    out = {"levels": api.fn("plan_levels")(h)}
    lists = []
    for l in range(out["levels"]):
        n = C.c_int64()
        assert api.fn("plan_export_level")(h, l, None, None, None, C.byref(n)) == 0
        a, f, w = np.empty(n.value, np.int64), np.empty(n.value, np.uint8), np.empty(n.value, np.int64)
        assert api.fn("plan_export_level")(h, l, _vp(a), _vp(f), _vp(w), C.byref(n)) == 0
        lists.append((a, f.astype(bool), w))
    out["lists"] = lists
    ob = alloc((N,), np.int64)
    assert api.fn("plan_order")(h, ptr(ob), None) == 0
    out["order"] = fetch(ob)
    C32, C64 = alloc((N, D), np.float32, Cm.astype(np.float32)), alloc((N, D), np.float64, Cm.astype(np.float64))
    T32, T64 = alloc((N, D), np.float32), alloc((N, D), np.float64)
    w64 = alloc((N,), np.float64)
    assert api.fn("fwd")(h, ptr(C32), C.c_int64(D), D, ptr(T32), C.c_int64(D), None, None) == 0, api.err()
    assert api.fn("fwd_f64")(h, ptr(C64), C.c_int64(D), D, ptr(T64), C.c_int64(D), ptr(w64), None) == 0, api.err()
    out["T32"], out["T64"], out["w"] = fetch(T32), fetch(T64), fetch(w64)
    Q = alloc((N, D), np.int32)
    st32, st64 = (C.c_float * 1)(step), (C.c_double * 1)(step)
    assert api.fn("fwd_quant_f64")(h, ptr(C64), C.c_int64(D), D, st64, 1, ptr(Q), C.c_int64(D), None) == 0, api.err()
    out["Q64"] = fetch(Q).copy()
    assert api.fn("fwd_quant")(h, ptr(C32), C.c_int64(D), D, st32, 1, ptr(Q), C.c_int64(D), None) == 0, api.err()
    out["Q32"] = fetch(Q).copy()
    R = alloc((N, D), np.float64)
    Qin = alloc((N, D), np.int32, out["Q64"])
    assert api.fn("dequant_inv_f64")(h, ptr(Qin), C.c_int64(D), D, st64, 1, ptr(R), C.c_int64(D), None) == 0, api.err()
    out["R64"] = fetch(R)
    Ri = alloc((N, D), np.float64)
    assert api.fn("inv_f64")(h, ptr(T64), C.c_int64(D), D, ptr(Ri), C.c_int64(D), None) == 0, api.err()
    out["C_back"] = fetch(Ri)
    assert api.fn("plan_destroy")(h) == 0
    return out


def _host_api(twins):
    api = Api(twins, "raht_cpu_")
    twins.raht_cpu_plan_destroy.argtypes = [C.c_void_p]
    twins.raht_cpu_plan_levels.argtypes = [C.c_void_p]
    twins.raht_cpu_plan_export_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    twins.raht_cpu_plan_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    twins.raht_cpu_plan_create.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    for n in ("fwd", "fwd_f64"):
        getattr(twins, "raht_cpu_" + n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    for n in ("inv", "inv_f64"):
        getattr(twins, "raht_cpu_" + n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    for n in ("fwd_quant", "dequant_inv"):
        getattr(twins, "raht_cpu_" + n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    for n in ("fwd_quant_f64", "dequant_inv_f64"):
        getattr(twins, "raht_cpu_" + n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_int64, C.c_void_p]

    def alloc(shape, dtype, init=None):
        a = np.empty(shape, dtype)
        if init is not None:
            a[...] = init
        return a
    return api, _vp, alloc, (lambda a: a)


@pytest.mark.parametrize("name", [n for n in golden_names(exclude_prefix="vox_") if "q_step1" in load_golden(n) and load_golden(n)["V"].shape[0] >= 8])
def test_twins_reproduce_the_reference(twins, name):
    g = load_golden(name)
    J = int(g["J"])
    api, ptr, alloc, fetch = _host_api(twins)
    o = _run_path(api, ptr, g["V"].astype(np.float64), g["C"], J, 1.0, alloc, fetch)
    assert o["levels"] == len(g["List"])
    for (a, f, w), La, Fa, Wa in zip(o["lists"], g["List"], g["Flags"], g["weights"]):
        assert np.array_equal(a, La) and np.array_equal(f, Fa) and np.array_equal(w, Wa)
    assert np.array_equal(o["order"], g["order"])
    np.testing.assert_allclose(o["T64"], g["T"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(o["w"], g["w"].reshape(-1))
    colmax = np.abs(g["T"]).max(axis=0)
    assert np.all(np.abs(o["T32"].astype(np.float64) - g["T"]).max(axis=0) <= 2e-7 * np.maximum(colmax, 1e-30))   # float64 rounded once
    bad = np.nonzero(o["Q64"] != g["q_step1"])
    if bad[0].size:                                       # only where the reference's quotient sits on a rounding tie
        q = g["T"][g["order"][bad[0]], bad[1]] + 0.5      # (integer-valued inputs produce many exact ties)
        assert np.all(np.abs(o["Q64"].astype(np.int64) - g["q_step1"])[bad] == 1)
        assert np.all(np.abs(q - np.round(q)) <= 1e-9 * np.maximum(1.0, np.abs(q)))
    np.testing.assert_allclose(o["C_back"], g["C"].astype(np.float64), rtol=1e-10, atol=1e-10 * max(1.0, float(np.abs(g["C"]).max())))


def test_twins_share_the_error_behaviour(twins):
    api, ptr, alloc, fetch = _host_api(twins)
    h = C.c_void_p()
    minV = (C.c_double * 3)(0.0, 0.0, 0.0)
    V = np.array([[0, 0, 1], [0, 0, 0]], dtype=np.float64)                       # unsorted
    assert api.fn("plan_create")(_vp(V), 1, C.c_int64(2), minV, C.c_double(4.0), 2, None, C.byref(h)) == -2
    V = np.array([[0, 0, 0], [0, 0, 0]], dtype=np.float64)                       # duplicate
    assert api.fn("plan_create")(_vp(V), 1, C.c_int64(2), minV, C.c_double(4.0), 2, None, C.byref(h)) == -2
    V = np.array([[0, 0, 0], [4, 0, 0]], dtype=np.float64)                       # out of bounds
    assert api.fn("plan_create")(_vp(V), 1, C.c_int64(2), minV, C.c_double(4.0), 2, None, C.byref(h)) == -3
    assert b"out of" in twins.raht_cpu_last_error()


@pytest.mark.gpu
def test_same_call_sequence_through_both_libraries(twins):
    """The swap by symbol name, end to end: one ctypes call sequence, `raht_` + device buffers vs `raht_cpu_` +
    host buffers, same results (integers equal; float64 1e-12; float32 2e-6 of the column max)."""
    import torch
    from raht_3dgs_codec_amd import _lib, synth
    L = _lib.lib()
    V, keys, Cm = synth.scene(20000, 9, 14, seed=5)
    host = _run_path(*_host_api(twins)[:2], V.astype(np.float64), Cm, 9, 0.5, *_host_api(twins)[2:])

    keep = []

    def alloc(shape, dtype, init=None):
        t = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device="cuda")
        if init is not None:
            t.copy_(torch.from_numpy(np.ascontiguousarray(init, dtype=dtype)))
        keep.append(t)
        return t
    dev = _run_path(Api(L, "raht_"), lambda t: C.c_void_p(t.data_ptr()), V.astype(np.float64), Cm, 9, 0.5, alloc,
                    lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1])
    assert dev["levels"] == host["levels"] and np.array_equal(dev["order"], host["order"])
    for (a, f, w), (b, g_, x) in zip(dev["lists"], host["lists"]):
        assert np.array_equal(a, b) and np.array_equal(f, g_) and np.array_equal(w, x)
    assert np.array_equal(dev["w"], host["w"])
    colmax = np.maximum(np.abs(host["T64"]).max(axis=0), 1.0)
    assert np.all(np.abs(dev["T64"] - host["T64"]).max(axis=0) <= 1e-12 * colmax)
    assert np.all(np.abs(dev["T32"].astype(np.float64) - host["T64"]).max(axis=0) <= 2e-6 * colmax)
    bad = np.nonzero(dev["Q64"] != host["Q64"])
    if bad[0].size:                                       # exact ties (integer xyz columns) only
        q = host["T64"][host["order"][bad[0]], bad[1]] / 0.5 + 0.5
        assert np.all(np.abs(dev["Q64"].astype(np.int64) - host["Q64"])[bad] == 1)
        assert np.all(np.abs(q - np.round(q)) <= 1e-9 * np.maximum(1.0, np.abs(q)))
    assert np.all(np.abs(dev["C_back"] - host["C_back"]).max(axis=0) <= 1e-10 * np.maximum(np.abs(Cm).max(axis=0), 1.0))


def test_batch_twins_equal_scene_by_scene_calls(twins):
    """raht_cpu_*_batch (the twins of the multi-scene entry points): n scenes through one call == one call per scene, and the
    borrowed-keys constructor builds the same plan."""
    rng = np.random.default_rng(8)
    D, n = 5, 3
    plans, Cs, keys_all = [], [], []
    for i in range(n):
        N = 500 + 130 * i
        keys = np.sort(rng.choice(1 << 15, size=N, replace=False)).astype(np.uint64)
        h = C.c_void_p()
        f = twins.raht_cpu_plan_create_from_keys_borrowed if i == 1 else twins.raht_cpu_plan_create_from_keys
        assert f(_vp(keys), C.c_int64(N), 15, None, None, C.byref(h)) == 0, twins.raht_cpu_last_error()
        plans.append(h); keys_all.append(keys)
        Cs.append(rng.standard_normal((N, D)).astype(np.float32))
    step = (C.c_float * 1)(0.05)
    vp = C.c_void_p
    hp = (vp * n)(*[p.value for p in plans])
    cp = (vp * n)(*[c.ctypes.data for c in Cs])
    ld = (C.c_int64 * n)(*[D] * n)
    Qb = [np.zeros((c.shape[0], D), np.int32) for c in Cs]
    Tb = [np.zeros_like(c) for c in Cs]
    qp = (vp * n)(*[q.ctypes.data for q in Qb])
    tp = (vp * n)(*[t.ctypes.data for t in Tb])
    assert twins.raht_cpu_fwd_quant_batch(n, hp, cp, ld, D, step, 1, qp, ld, None) == 0, twins.raht_cpu_last_error()
    assert twins.raht_cpu_fwd_batch(n, hp, cp, ld, D, tp, ld, None) == 0
    Rb = [np.zeros_like(c) for c in Cs]
    rp = (vp * n)(*[r.ctypes.data for r in Rb])
    assert twins.raht_cpu_dequant_inv_batch(n, hp, qp, ld, D, step, 1, rp, ld, None) == 0
    for i in range(n):
        Q1 = np.zeros_like(Qb[i]); T1 = np.zeros_like(Tb[i]); R1 = np.zeros_like(Rb[i])
        assert twins.raht_cpu_fwd_quant(plans[i], _vp(Cs[i]), C.c_int64(D), D, step, 1, _vp(Q1), C.c_int64(D), None) == 0
        assert twins.raht_cpu_fwd(plans[i], _vp(Cs[i]), C.c_int64(D), D, _vp(T1), C.c_int64(D), None, None) == 0
        assert twins.raht_cpu_dequant_inv(plans[i], _vp(Q1), C.c_int64(D), D, step, 1, _vp(R1), C.c_int64(D), None) == 0
        assert np.array_equal(Q1, Qb[i]) and np.array_equal(T1, Tb[i]) and np.array_equal(R1, Rb[i])
        twins.raht_cpu_plan_destroy(plans[i])
    assert twins.raht_cpu_fwd_batch(0, hp, cp, ld, D, tp, ld, None) != 0


def _voxelize_calls(lib, prefix, ptr, alloc, fetch, PC, J):
    """raht[_cpu]_voxelize_all and raht[_cpu]_voxelize_plan on one cloud -> dict of their outputs (numpy)."""
    n, ld = PC.shape
    d = ld - 3
    vp = C.c_void_p
    P = alloc((n, ld), np.float32, PC)
    keys, idx, vi = alloc((n,), np.int64), alloc((n,), np.int64), alloc((n,), np.int64)
    pcv, pcs, dl = alloc((n, ld), np.float32), alloc((n, ld), np.float32), alloc((n, ld), np.float32)
    nv = C.c_int64(); vmin = (C.c_float * 3)(); w = C.c_double(); vs = C.c_double()
    f = getattr(lib, prefix + "voxelize_all")
    f.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_double, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                  C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    assert f(ptr(P), ld, n, d, None, -1.0, J, ptr(keys), ptr(idx), ptr(vi), ptr(pcv), None, ptr(pcs), ptr(dl), C.byref(nv), vmin,
             C.byref(w), C.byref(vs), None) == 0
    out = dict(nv=nv.value, keys=fetch(keys).copy(), idx=fetch(idx).copy(), vi=fetch(vi)[: nv.value].copy(), pcv=fetch(pcv)[: nv.value].copy(),
               pcs=fetch(pcs).copy(), dl=fetch(dl).copy(), vmin=list(vmin), width=w.value, vs=vs.value)
    g = getattr(lib, prefix + "voxelize_plan")
    g.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_float), C.c_double, C.c_int, vp, vp, vp, C.POINTER(C.c_int64),
                  C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double), vp, C.POINTER(vp)]
    vk, vi2, pcv2 = alloc((n,), np.int64), alloc((n,), np.int64), alloc((n, ld), np.float32)
    h = vp()
    assert g(ptr(P), ld, n, d, None, -1.0, J, ptr(vk), ptr(vi2), ptr(pcv2), C.byref(nv), vmin, C.byref(w), C.byref(vs), None, C.byref(h)) == 0
    size = getattr(lib, prefix + "plan_size"); size.argtypes = [vp]; size.restype = C.c_int64
    order = alloc((nv.value,), np.int64)
    po = getattr(lib, prefix + "plan_order"); po.argtypes = [vp, vp, vp]
    assert size(h) == nv.value and po(h, ptr(order), None) == 0
    out.update(vkeys=fetch(vk)[: nv.value].copy(), vi2=fetch(vi2)[: nv.value].copy(), pcv2=fetch(pcv2)[: nv.value].copy(), order=fetch(order).copy())
    pd = getattr(lib, prefix + "plan_destroy"); pd.argtypes = [vp]
    assert pd(h) == 0
    return out


def _cloud(seed, n, d):
    rng = np.random.default_rng(seed)
    P = (rng.random((n, 3)) * 4.0 - 1.0).astype(np.float32)
    P[::7] = P[1::7][: P[::7].shape[0]]
    return np.concatenate([P, rng.standard_normal((n, d)).astype(np.float32)], axis=1)


def test_voxelizer_twins_one_call_forms(twins):
    """raht_cpu_voxelize_all == raht_cpu_voxelize + raht_cpu_voxelize_residuals (which the golden fixtures pin), and
    raht_cpu_voxelize_plan builds its plan from the keys of the voxels' first points."""
    from oracle import oracle as orc
    PC = _cloud(21, 4000, 6)
    o = _voxelize_calls(twins, "raht_cpu_", _vp, lambda s, t, i=None: (np.ascontiguousarray(i, dtype=t).copy() if i is not None else np.zeros(s, t)), lambda a: a, PC, 6)
    r = orc.voxelize(PC, 6)
    pcs, dl = orc.voxel_residuals(PC, r)
    assert o["nv"] == r["Nvox"] and np.array_equal(o["idx"], r["sort_idx"]) and np.array_equal(o["vi"], r["voxel_indices"])
    assert np.array_equal(o["pcv"], r["PCvox"]) and np.array_equal(o["pcs"], pcs) and np.array_equal(o["dl"], dl)
    assert np.array_equal(o["vkeys"].view(np.uint64), np.asarray(r["keys_sorted"]).view(np.uint64)[r["voxel_indices"]])
    assert np.array_equal(o["vi2"], o["vi"]) and np.array_equal(o["pcv2"], o["pcv"])
    assert np.array_equal(np.sort(o["order"]), np.arange(o["nv"]))


@pytest.mark.gpu
@pytest.mark.parametrize("d", [2, 11, 56])
def test_voxelizer_one_call_forms_through_both_libraries(twins, d):
    """raht_voxelize_all / raht_voxelize_plan on the MI355X against their host twins: every output bit for bit (narrow clouds take
    the two-call sequence inside, wide ones the one-pass kernel)."""
    import torch
    from raht_3dgs_codec_amd import _lib
    PC = _cloud(30 + d, 30000, d)
    host = _voxelize_calls(twins, "raht_cpu_", _vp, lambda s, t, i=None: (np.ascontiguousarray(i, dtype=t).copy() if i is not None else np.zeros(s, t)), lambda a: a, PC, 7)
    keep = []

    def alloc(shape, dtype, init=None):
        t = torch.zeros(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device="cuda")
        if init is not None:
            t.copy_(torch.from_numpy(np.ascontiguousarray(init, dtype=dtype)))
        keep.append(t)
        return t
    dev = _voxelize_calls(_lib.lib(), "raht_", lambda t: C.c_void_p(t.data_ptr()), alloc, lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1], PC, 7)
    for k in host:
        if isinstance(host[k], np.ndarray):
            assert np.array_equal(dev[k], host[k]), k
        else:
            assert dev[k] == host[k], k
