"""GPU regression / robustness tests for conditions that once broke (or could break) the operator boundary:

* the round-1 abort: an INVERSE transform whose last TILE stage has survivors (the roots)
  and therefore no stage above it (wsn == nullptr);
* the level-engine fallback when a tile schedule is abandoned (raht_plan_set_max_stages);
* the exits of the device-driven schedule chain (a stage larger than its buffer, a tree that reaches the top stage
  early, many small stages inside the one tail launch);
* schedules of several geometries on one plan (float32 then float64 then another D: the schedule cache grows
  while earlier schedules are still referenced);
* plans on two devices in one process (skipped on a one-GPU box);
* the float64 quantizer (reference precision) against the reference's integers.
"""
import os

import numpy as np
import pytest

from .conftest import golden_names, load_golden

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


# ----------------------------------------------------------------------------- the round-1 abort, pinned
@pytest.mark.parametrize("with_root_buffer", [False, True])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_last_tile_stage_with_survivors_has_no_stage_above(rt, oracle, with_root_buffer, dtype):
    """8192 dense keys, tiles of 64 rows, tree truncated at level 6: every tile holds one complete 64-row node,
    so stage 0 leaves ONLY the 128 roots -> stage 0 is the last stage, it is a TILE stage (8192 > 4096 rows:
    no top_kernel), and its survivors have no workspace above them. The inverse must take them from T (or from
    the caller's root buffer), never from wsn."""
    import torch
    N, J, D, top = 8192, 5, 11, 6
    keys = np.arange(N, dtype=np.uint64)                      # 15-bit dense Morton keys
    rng = np.random.default_rng(5)
    C = rng.standard_normal((N, D)).astype(np.float32 if dtype == "f32" else np.float64)
    td = torch.float32 if dtype == "f32" else torch.float64
    plan = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 3 * J, top_level=top)
    plan.set_engine("tile", 64, 64, 0, 0)
    st = plan.stage_stats(4 if dtype == "f32" else 8, D)
    assert st["valid"] and st["rows_per_stage"] == [N], st     # one stage: a tile stage that is also the last
    assert plan.n_roots == N // 64
    Cd = _dev(C)
    roots = torch.empty((plan.n_roots, D), dtype=td, device="cuda") if with_root_buffer else None
    T = plan.forward(Cd, want_w=False, roots=roots)
    # reference: every 64-row node transformed on its own (the truncated tree) = the level engine of the same plan
    plan.set_engine("level")
    Tl = plan.forward(Cd, want_w=False)
    plan.set_engine("tile", 64, 64, 0, 0)
    tol = 1e-12 if dtype == "f64" else 2e-6
    nonroot = torch.ones(N, dtype=torch.bool, device="cuda")
    nonroot[plan.root_rows] = False
    assert (T - Tl)[nonroot].abs().max().item() <= tol * Tl.abs().max().item()
    if with_root_buffer:
        assert (roots - Tl[plan.root_rows]).abs().max().item() <= tol * Tl.abs().max().item()
        Crec = plan.inverse(T, roots=roots)
    else:
        assert (T - Tl).abs().max().item() <= tol * Tl.abs().max().item()
        Crec = plan.inverse(T)                                  # <- the launch that aborted the process in round 1
    torch.cuda.synchronize()
    assert (Crec - Cd).abs().max().item() <= (1e-12 if dtype == "f64" else 1e-5) * Cd.abs().max().item()
    if dtype == "f32" and not with_root_buffer:
        Q = plan.forward_quant(Cd, 0.01)
        assert torch.equal(Q, plan.quant_reorder(T, 0.01))
        assert torch.equal(plan.dequant_inverse(Q, 0.01), plan.inverse(plan.dequant_unreorder(Q, 0.01)))
    # each 64-row node is a complete 6-level subtree: its root coefficient is the node's sum / 8
    want = Cd.double().reshape(N // 64, 64, D).sum(dim=1) / 8.0
    got = (roots if with_root_buffer else T[plan.root_rows]).double()
    assert (got - want).abs().max().item() <= (1e-12 if dtype == "f64" else 1e-5) * want.abs().max().item()


# ------------------------------------------------------------------- abandoned schedule -> level engine
def test_abandoned_tile_schedule_falls_back_to_the_level_engine(rt, oracle):
    """A schedule that needs more stages than the plan allows is marked invalid and every entry point runs
    the one-launch-per-level engine instead: same results (fwd, inv, fused quantize, fused dequantize)."""
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, C = synth.scene(40000, 10, 14, seed=21)
    po = oracle.raht_param(V.astype(np.float64), np.zeros(3), 2 ** 10, 10)
    To, _ = oracle.raht_fwd(C.astype(np.float64), po)
    plan = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 30)
    plan.set_engine("tile", 64, 64, 0, 64)                    # small tiles: the full schedule has several stages
    full = plan.stage_stats(4, 14)
    assert full["valid"] and len(full["rows_per_stage"]) >= 3
    Cd = _dev(C)
    T_tile = plan.forward(Cd, want_w=False)
    plan.set_max_stages(2)
    cut = plan.stage_stats(4, 14)
    assert not cut["valid"]                                   # abandoned -> fallback
    T_fb = plan.forward(Cd, want_w=False)
    colmax = np.abs(To).max(axis=0)
    assert np.all(np.abs(T_fb.cpu().numpy().astype(np.float64) - To).max(axis=0) <= 2e-6 * colmax)
    assert (T_fb - T_tile).abs().max().item() <= 4e-6 * float(colmax.max())
    Crec = plan.inverse(T_fb)
    assert (Crec - Cd).abs().max().item() <= 1e-5 * Cd.abs().max().item()
    Q = plan.forward_quant(Cd, 0.05)
    assert torch.equal(Q, plan.quant_reorder(T_fb, 0.05))
    assert torch.equal(plan.dequant_inverse(Q, 0.05), plan.inverse(plan.dequant_unreorder(Q, 0.05)))
    T64 = plan.forward(_dev(C.astype(np.float64)), want_w=False)
    np.testing.assert_allclose(T64.cpu().numpy(), To, rtol=1e-12, atol=1e-12 * float(colmax.max()))
    plan.set_max_stages(24)
    assert plan.stage_stats(4, 14)["valid"]
    assert torch.equal(plan.forward(Cd, want_w=False), T_tile)


# ------------------------------------------------- the device-driven schedule chain and its exits
@pytest.mark.parametrize("case", ["half_roots", "early_top", "many_small_stages"])
def test_schedule_chain_exits(rt, case):
    """The schedule is built by a chain of launches that never returns to the host (plan.hip: build_schedule_fast):
    multi-workgroup stages, then ONE launch for every small stage and the top stage. Its exits must all end in a
    correct transform:
    * half_roots: a tree truncated so that half the rows are roots -- stage 0 keeps more entries than the buffer of
      stage 1 holds (1/3 of the stage before): the chain must STOP there (it once went on and read past the buffer:
      a GPU memory fault) and the exact builder takes over;
    * early_top: tiles of 512 rows shrink a 30 000-row tree below the top stage's size one stage earlier than the
      host expected when it enqueued the chain;
    * many_small_stages: 64-row tiles, 64-entry top stage: two tile stages and the top stage run inside the one tail launch."""
    import torch
    from tests.numpy_ops import NumpyPlan
    rng = np.random.default_rng({"half_roots": 1, "early_top": 2, "many_small_stages": 3}[case])
    N, nbits, D = 30000, 30, 7
    keys = np.unique(rng.integers(0, (1 << nbits) - 1, size=N + 64, dtype=np.int64))[:N]
    top, geom = None, (0, 0, 0, 0)
    if case == "half_roots":
        ref_all = NumpyPlan(torch.from_numpy(keys.copy()), nbits)
        lv = np.sort(np.asarray(ref_all.lvl)[1:])
        top = int(lv[len(lv) // 2])                               # about half of the rows sit at or above this level
    elif case == "early_top":
        geom = (512, 512, 0, 4096)
    else:
        geom = (64, 64, 0, 64)
    kt = torch.from_numpy(keys.copy())
    ref = NumpyPlan(kt, nbits, top_level=top)
    p = rt.RahtPlan.from_keys(kt.cuda(), nbits, top_level=top)
    p.set_engine("tile", *geom)
    st = p.stage_stats(4, D)
    assert st["valid"]
    if case == "half_roots":
        assert 0.3 * N < p.n_roots < 0.7 * N
    if case == "many_small_stages":
        assert len(st["rows_per_stage"]) >= 4
    assert np.array_equal(p.root_rows.cpu().numpy(), ref.root_rows.numpy())
    C = rng.normal(size=(N, D))
    Cd = torch.from_numpy(C).cuda()
    roots = torch.empty((p.n_roots, D), dtype=Cd.dtype, device="cuda")
    T = p.forward(Cd, want_w=False, roots=roots)
    r_ref = torch.empty((ref.n_roots, D), dtype=torch.float64)
    T_ref = ref.forward(torch.from_numpy(C), roots=r_ref).numpy()
    scale = np.maximum(np.abs(T_ref).max(axis=0), 1e-300)
    assert (np.abs(T.cpu().numpy() - T_ref).max(axis=0) / scale).max() <= 1e-12
    assert (np.abs(roots.cpu().numpy() - r_ref.numpy()).max(axis=0) / scale).max() <= 1e-12
    R = p.inverse(T, roots=roots)
    assert (R - Cd).abs().max().item() <= 1e-11 * Cd.abs().max().item()


def test_deep_schedules_take_several_height_launches(rt, monkeypatch):
    """The butterfly heights of a schedule's tile stages come from one launch per 8 stages (plan.hip: launch_stage_heights);
    schedules with more tile stages (deep / unbalanced key sets, small tail tiles) used to be refused, failing every transform
    of the plan. Here the per-launch group is cut to 2 so that an ordinary four-stage schedule walks the several-launches
    path: same heights, hence bit-identical transforms."""
    import torch
    rng = np.random.default_rng(11)
    N, nbits, D = 30000, 30, 9
    keys = torch.from_numpy(np.unique(rng.integers(0, (1 << nbits) - 1, size=N + 64, dtype=np.int64))[:N]).cuda()
    C = torch.from_numpy(rng.normal(size=(N, D)).astype(np.float32)).cuda()
    p0 = rt.RahtPlan.from_keys(keys, nbits)
    p0.set_engine("tile", 64, 64, 0, 64)
    T0 = p0.forward(C, want_w=False)
    Q0 = p0.forward_quant(C, 0.02)
    monkeypatch.setenv("RAHT_HEIGHT_STAGES_PER_LAUNCH", "2")
    p1 = rt.RahtPlan.from_keys(keys, nbits)
    p1.set_engine("tile", 64, 64, 0, 64)
    st = p1.stage_stats(4, D)
    assert st["valid"] and len(st["rows_per_stage"]) >= 4
    assert torch.equal(p1.forward(C, want_w=False), T0)
    assert torch.equal(p1.forward_quant(C, 0.02), Q0)
    assert torch.equal(p1.dequant_inverse(Q0, 0.02), p0.dequant_inverse(Q0, 0.02))


# ----------------------------------------------------------- several schedules alive on one plan
def test_schedule_cache_growth_keeps_earlier_schedules_usable(rt):
    """float32 D = 59 (default schedule), float64, D = 200 (channel chunks), forced geometries: each adds a
    schedule to the plan's cache; the earlier ones must stay valid (they are referenced again afterwards)."""
    import torch
    from raht_3dgs_codec_amd import synth
    V, keys, C = synth.scene(30000, 10, 59, seed=31)
    plan = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 30)
    Cd = _dev(C)
    T0 = plan.forward(Cd, want_w=False)
    results = []
    for geom in [(0, 0, 0, 0), (64, 64, 0, 64), (128, 64, 8, 64), (192, 128, 16, 256), (256, 64, 4, 1024)]:
        plan.set_engine("tile", *geom)
        results.append(plan.forward(Cd, want_w=False))
        plan.forward(Cd.double(), want_w=False)
        wide = torch.cat([Cd, Cd, Cd, Cd[:, :23]], dim=1).contiguous()             # D = 200
        Tw = plan.forward(wide, want_w=False)
        if geom[0]:        # forced geometry: the schedule does not depend on D, every butterfly runs in the same kernel
            assert torch.equal(Tw[:, :59], results[-1]) and torch.equal(Tw[:, 59:118], results[-1])
        else:              # automatic: D = 200 picks other tile sizes, a butterfly may move between tile and top kernel
            assert (Tw[:, :59] - results[-1]).abs().max().item() <= 4e-6 * T0.abs().max().item()
    plan.set_engine("tile", 0, 0, 0, 0)
    assert torch.equal(plan.forward(Cd, want_w=False), T0)
    for r in results:
        assert (r - T0).abs().max().item() <= 4e-6 * T0.abs().max().item()


# ------------------------------------------------------------------------------ two devices, one process
def test_plans_on_two_devices(rt):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from raht_3dgs_codec_amd import synth, _lib
    V, keys, C = synth.scene(50000, 10, 59, seed=41)
    out = []
    plans = []
    for d in (0, 1):
        dev = torch.device("cuda", d)
        with torch.cuda.device(dev):
            kd = torch.from_numpy(keys.view(np.int64)).to(dev)
            p = rt.RahtPlan.from_keys(kd, 30)
            plans.append(p)
            Cd = torch.from_numpy(C).to(dev)
            T = p.forward(Cd, want_w=False)
            Q = p.forward_quant(Cd, 0.01)
            out.append((T.cpu(), Q.cpu(), p.inverse(T).cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    # a plan used while ANOTHER device is current is refused, not run on the wrong GPU's memory
    import ctypes as Cc
    with torch.cuda.device(1):
        x = torch.zeros((plans[0].N, 59), device="cuda:1")
        rc = _lib.lib().raht_fwd(plans[0]._h, Cc.c_void_p(x.data_ptr()), 59, 59, Cc.c_void_p(x.data_ptr()), 59, None, None)
        assert rc == -1 and b"device" in _lib.lib().raht_last_error()
    # freeing the device-0 plan and building a new device-1 plan must not hand device-0 blocks to device 1
    del plans[0]
    with torch.cuda.device(1):
        p1 = rt.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).to("cuda:1"), 30)
        T = p1.forward(torch.from_numpy(C).to("cuda:1"), want_w=False)
        assert torch.equal(T.cpu(), out[0][0])


# ----------------------------------------------------------- float64 quantizer (reference precision)
@pytest.mark.parametrize("name", [n for n in golden_names(exclude_prefix="vox_") if any(k.startswith("q_step") for k in load_golden(n))])
def test_float64_quantizer_reproduces_the_reference_integers(rt, name):
    """encode_3dgs.py:204,210,215 on float64 coefficients, as the reference computes them (CPU torch: true
    division): the float64 kernels give the reference's integers, differing only where the reference's quotient
    sits on a rounding tie (the transforms agree to 1 ulp, not bitwise)."""
    import torch
    g = load_golden(name)
    J = int(g["J"])
    plan = rt.RahtPlan.from_coords(_dev(g["V"].astype(np.float64)), torch.zeros(3, dtype=torch.float64), 2 ** J, J)
    C64 = _dev(g["C"].astype(np.float64))
    T64, _ = plan.forward(C64)
    order = g["order"]
    for k in sorted(k for k in g if k.startswith("q_step")):
        step = float(k[len("q_step"):])
        Qref = g[k].astype(np.int64)
        for Q in (plan.quant_reorder(T64, step), plan.forward_quant(C64, step)):
            Qg = Q.cpu().numpy().astype(np.int64)
            bad = np.nonzero(Qg != Qref)
            if bad[0].size:
                q = g["T"][order[bad[0]], bad[1]] / step
                assert np.all(np.abs(Qg[bad] - Qref[bad]) == 1)
                assert np.all(np.abs(q + 0.5 - np.round(q + 0.5)) <= 1e-9 * np.maximum(1.0, np.abs(q)))
        Td = plan.dequant_unreorder(_dev(g[k].astype(np.int32)), step, dtype=torch.float64)
        want = np.empty_like(g["T"])
        want[order] = g[k].astype(np.float64) * step                          # encode_3dgs.py:261,267-268
        np.testing.assert_array_equal(Td.cpu().numpy(), want)
        Crec = plan.dequant_inverse(_dev(g[k].astype(np.int32)), step, dtype=torch.float64)
        assert Crec.dtype == torch.float64
        assert (Crec - plan.inverse(Td)).abs().max().item() == 0.0
        ref_rec = g["crec_step" + k[len("q_step"):]]                         # the reference's reconstruction
        np.testing.assert_allclose(Crec.cpu().numpy(), ref_rec, rtol=1e-12, atol=1e-12 * max(1.0, float(np.abs(ref_rec).max())))


def test_forward_chaining_of_later_stages_is_bit_identical(tmp_path):
    """RAHT_CHAIN=1 (measured slower, off by default: DESIGN.md 10): the later tile stages of the forward direction as ONE launch in
    which the workgroup that completes a parent tile's inputs runs it. Same tiles, same code: bit-identical outputs. Run in a
    child process (the knob is read once)."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
V, keys, C = synth.scene(700000, 11, 59, seed=3)
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 33)
st = p.stage_stats(4, 59)
assert len(st["rows_per_stage"]) >= 4, st
Cd = torch.from_numpy(C).cuda()
T = p.forward(Cd, want_w=False); Q = p.forward_quant(Cd, 0.01)
torch.save((T.cpu(), Q.cpu()), sys.argv[1])
'''
    outs = []
    for chain in ("0", "1"):
        f = tmp_path / f"out{chain}.pt"
        env = dict(os.environ, RAHT_CHAIN=chain)
        r = subprocess.run([sys.executable, "-c", code % ROOT, str(f)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        import torch
        outs.append(torch.load(f))
    import torch
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
