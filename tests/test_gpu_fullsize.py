"""Full-size parity: the HIP path on WHOLE BASELINE.json scenes against the CPU oracle (not only through
size-independent properties). The oracle does a 3 M x 59 forward pass in ~2.5 s on one core; here it runs
one thread per channel block (oracle/threaded.py).

  cfg2      993 262 x 14, J = 10      BASELINE configs[1]
  cfg3    2 999 072 x 59, J = 12      BASELINE configs[2] (the headline scene)
  cfg4    one scene of configs[3]'s largest size class: 6 M draws x 59, J = 12, seed 11
  cfg5    50 M x 59 on ONE GPU, generated on the device: properties only (the oracle would need ~50 GB
          of float64 on the host and minutes) -- round trip, Parseval, DC, fused == two-call

Bars as in test_gpu_parity.py: integers bit-exact; float64 1e-12; float32 per column 2e-6 * max / 1e-6 * rms;
quantized integers: |Q32 - Q_oracle| <= 1 + |T32 - T64| / step elementwise (Q32 IS floor(T32 / step + 0.5):
the fused == two-call and exact-division tests pin that), float64 Q equal to the oracle's except next to an
exact rounding tie.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def _cases():
    from raht_3dgs_codec_amd import synth
    return {"cfg2": synth.CONFIGS["cfg2"], "cfg3": synth.CONFIGS["cfg3"], "cfg4_6M": (6_000_000, 12, 59, 11)}



def _col_stats(T32, T64):
    """per-column max / rms error of a float32 result against the float64 oracle, blockwise (bounded temporaries)"""
    N, D = T64.shape
    err = np.zeros(D); se = np.zeros(D); colmax = np.zeros(D); sq = np.zeros(D)
    for r0 in range(0, N, 1 << 19):
        a = T32[r0:r0 + (1 << 19)].astype(np.float64)
        b = T64[r0:r0 + (1 << 19)]
        d = a - b
        err = np.maximum(err, np.abs(d).max(axis=0))
        se += (d * d).sum(axis=0)
        colmax = np.maximum(colmax, np.abs(b).max(axis=0))
        sq += (b * b).sum(axis=0)
    return err, np.sqrt(se / N), colmax, np.sqrt(sq / N)


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4_6M"])
def test_full_scene_matches_oracle(rt, oracle, name):
    import torch
    from oracle import threaded
    from raht_3dgs_codec_amd import synth
    n, J, D, seed = _cases()[name]
    V, keys, C = synth.scene(n, J, D, seed)
    N = V.shape[0]
    C64 = C.astype(np.float64)
    po = oracle.raht_param(V.astype(np.float64), np.zeros(3), 2 ** J, J)
    To, wo = threaded.forward(oracle, C64, po)                                   # RAHT.py:252-336, float64

    # the plan: from coordinates for cfg2 (the drivers' call), from sorted keys for the larger scenes
    if name == "cfg2":
        plan = rt.RahtPlan.from_coords(_dev(V.astype(np.float64)), [0, 0, 0], 2 ** J, J)
    else:
        plan = rt.RahtPlan.from_keys(_dev(keys.view(np.int64)), 3 * J)
    assert plan.levels == po.nlevels
    assert np.array_equal(plan.order_RAGFT.cpu().numpy(), po.order)              # RAHT_param.py:251-274, bit-exact
    st = plan.stage_stats(4, D)
    assert st["valid"] and st["rows_per_stage"][0] == N

    # ---- float32 forward (the fast path) ----
    Cd = _dev(C)
    T32d, w = plan.forward(Cd)
    T32 = T32d.cpu().numpy()
    err, rms, colmax, colrms = _col_stats(T32, To)
    assert np.all(err <= 2e-6 * np.maximum(colmax, 1e-30)), (name, float((err / np.maximum(colmax, 1e-30)).max()))
    assert np.all(rms <= 1e-6 * np.maximum(colrms, 1e-30)), (name, float((rms / np.maximum(colrms, 1e-30)).max()))
    assert np.array_equal(w.cpu().numpy().reshape(-1).astype(np.float64), wo.reshape(-1))       # RAHT.py:325-328
    # ---- float32 inverse of the float32 coefficients ----
    Crec = plan.inverse(T32d)
    assert (Crec - Cd).abs().max().item() <= 1e-5 * Cd.abs().max().item()       # encode_3dgs.py:186-195

    # ---- float64 (the reference's precision): forward, and inverse of the ORACLE's coefficients ----
    C64d = _dev(C64)
    T64d, _ = plan.forward(C64d)
    scale = torch.from_numpy(np.maximum(colmax, 1.0)).cuda()
    Tod = _dev(To)
    assert bool(((T64d - Tod).abs() <= 1e-12 * scale + 1e-12 * Tod.abs()).all())
    Cinv = plan.inverse(Tod)
    cscale = C64d.abs().amax(dim=0).clamp_min(1.0)
    # derived bar: a butterfly rounds each output by <= 2.5 eps64 of the node's low-pass magnitude, which is <= sqrt(w) max|C|
    # for a node of w leaves (sqrt(N) at the root, shrinking by sqrt(2) per level: the sum over a root-to-leaf path is
    # <= 3.5 sqrt(N) max|C|); times 4 for the forward's own error in the oracle's coefficients and slack
    inv_bar = 4 * 2.5 * 2.2e-16 * 3.5 * np.sqrt(N)                                 # 1.3e-11 at 3 M rows, 1.9e-11 at 6 M
    assert bool(((Cinv - C64d).abs() <= inv_bar * cscale).all())
    del Cinv, C64d

    # ---- quantize + reorder (encode_3dgs.py:204,210,215) at two steps ----
    order = po.order
    for step in (0.01, 1.0):
        Qo = oracle.quant_reorder(To, step, order)                               # float64, true division
        # float64 kernels: equal except where the oracle's quotient sits on a rounding tie (1-ulp transform noise)
        Q64 = plan.quant_reorder(T64d, step).cpu().numpy()
        bad = np.nonzero(Q64 != Qo)
        if bad[0].size:
            q = To[order[bad[0]], bad[1]] / step
            assert np.all(np.abs(Q64[bad] - Qo[bad]) == 1)
            assert np.all(np.abs(q + 0.5 - np.round(q + 0.5)) <= 1e-9 * np.maximum(1.0, np.abs(q))), (name, step, bad[0].size)
        # (the integer-valued xyz columns are full of EXACT ties at step 1: the high-pass of two equal-weight
        # two-point nodes is ((x2 + x3) - (x0 + x1)) / 2; which way 1-ulp noise tips them is implementation-defined,
        # also between the reference's own CPU and GPU runs.) The attribute columns have next to none:
        a0 = 3 if D in (14, 59) else 0
        assert int((bad[1] >= a0).sum()) <= 1e-6 * Qo.size + 2, (name, step, int((bad[1] >= a0).sum()))
        del Q64
        # float32 fused kernel against the float64 oracle. SURVEY 8c: "mismatch rate <= 1e-6 and each mismatch = +-1" was
        # sized on step-1 data; what float32 can deliver at ANY step is derived from its coefficient error:
        #   (a) per element |dQ| <= 1 + (|T32 - T64| + 1.2e-7 |T64|) / step  (coefficient error + the float32 quotient's
        #       rounding, both in units of the step; on the xyz columns |T| reaches 1e6, so |T| / step > 2^24 at step 0.01
        #       and float32 integers cannot be exact there: include/raht.h tells callers to use raht_fwd_quant_f64 then);
        #   (b) the NUMBER of differing integers: an integer differs when a rounding boundary falls between the two
        #       quotients, which for a quotient spread over many integers happens with probability |T32 - T64| / step --
        #       so the count must stay near sum(|dT|) / step. A rounding-mode or +0.5 mistake flips ~every integer.
        Q32 = plan.forward_quant(Cd, step).cpu().numpy()
        dT = np.abs(T32.astype(np.float64) - To)[order]
        lim = 1.0 + (dT + 1.2e-7 * np.abs(To)[order]) / step
        dq = np.abs(Q32.astype(np.int64) - Qo.astype(np.int64))
        assert np.all(dq <= lim), (name, step, int((dq > lim).sum()))
        # attribute channels (not the xyz columns): mismatches are +-1, and as many as the coefficient error predicts
        nz = dq[:, a0:] != 0
        n_bad = int(nz.sum())
        expected = float(dT[:, a0:].sum() / step)
        rate = n_bad / nz.size
        print(f"[fullsize] {name} step {step}: attribute-channel integer mismatches {n_bad} of {nz.size} (rate {rate:.3g}); "
              f"sum|dT|/step predicts {expected:.1f}; xyz columns max |dQ| {int(dq[:, :a0].max()) if a0 else 0}")
        assert np.all(dq[:, a0:] <= 1), (name, step, int(dq[:, a0:].max()))
        # (the prediction treats error and distance-to-boundary as independent; they are mildly correlated -- large
        # coefficients carry the large errors -- hence the factor)
        assert n_bad <= 4.0 * expected + 8.0 * np.sqrt(expected) + 10, (name, step, n_bad, expected)
        # (no separate rate cap: the count bound above IS the float32 contract -- DESIGN.md 3 -- and a cap set from a
        # measurement can only catch regressions. Measured rates at step 0.01: cfg3 7.9e-7, cfg4-6M 6.7e-7, cfg2 2.2e-6.)
        # xyz columns in float32: reported above on their own; every element is inside (a). Their integer-valued inputs put ~1 % of
        # the coefficients on EXACT rounding ties at step 1 (see the float64 comparison above), which float32 noise tips
        # either way, so no rate is asserted for them.
        del dT, nz
        # ---- mixed precision: the xyz columns carried in float64 inside the float32 launches (raht_fwd_quant_mixed) ----
        # those columns: the oracle's integers except on exact rounding ties (the float64 bar); the others: the float32 kernels'
        # integers bit for bit
        if a0:
            Qmx = plan.forward_quant_mixed(Cd, step, a0).cpu().numpy()
            assert np.array_equal(Qmx[:, a0:], Q32[:, a0:]), (name, step)
            badx = np.nonzero(Qmx[:, :a0] != Qo[:, :a0])
            if badx[0].size:
                q = To[order[badx[0]], badx[1]] / step
                assert np.all(np.abs(Qmx[:, :a0][badx].astype(np.int64) - Qo[:, :a0][badx]) == 1)
                assert np.all(np.abs(q + 0.5 - np.round(q + 0.5)) <= 1e-9 * np.maximum(1.0, np.abs(q))), (name, step, badx[0].size)
            print(f"[fullsize] {name} step {step}: mixed precision: xyz integers off the oracle's {badx[0].size} (all on exact rounding ties), "
                  f"float32 alone: max |dQ| {int(dq[:, :a0].max())}")
            del Qmx
        del lim, dq, Q32
        # ... and back, from the ORACLE's integers: dequantize + un-reorder + inverse (encode_3dgs.py:261,267-268,274)
        Cq = plan.dequant_inverse(_dev(Qo), step)
        ref = torch.from_numpy(threaded.inverse(oracle, oracle.dequant_unreorder(Qo, step, order), po)).cuda()
        cs = ref.abs().amax(dim=0).clamp_min(1.0)
        assert bool(((Cq.double() - ref).abs() <= 1e-5 * cs).all()), (name, step)
        if a0:
            # mixed decode: the xyz columns = the float64 inverse rounded once to float32; the others bit-identical to the float32 decode
            Cmx = plan.dequant_inverse_mixed(_dev(Qo), step, a0)
            assert torch.equal(Cmx[:, a0:], Cq[:, a0:]), (name, step)
            ex = (Cmx[:, :a0].double() - ref[:, :a0]).abs()
            assert bool((ex <= 6e-8 * ref[:, :a0].abs() + 1e-10 * cs[:a0]).all()), (name, step, float(ex.max()))
            del Cmx
        del Qo, Cq, ref


def test_cfg5_single_gpu_properties(rt):
    """50 M Gaussians x 59 channels on one GPU (BASELINE configs[4] before sharding): 11.8 GB per matrix."""
    import torch
    from raht_3dgs_codec_amd import synth
    n, J, D, seed = synth.CONFIGS["cfg5"]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(seed)
    kraw = torch.randint(0, 1 << (3 * J), (int(n * 1.002),), device=dev, dtype=torch.int64, generator=g)
    keys = torch.unique(kraw)[:n].contiguous()
    del kraw
    N = int(keys.shape[0])
    Cd = torch.empty((N, D), dtype=torch.float32, device=dev)
    for c0 in range(0, D, 8):
        Cd[:, c0:c0 + 8] = torch.randn((N, min(8, D - c0)), device=dev, generator=g)
    plan = rt.RahtPlan.from_keys(keys, 3 * J)
    st = plan.stage_stats(4, D)
    assert st["valid"] and st["rows_per_stage"][0] == N
    order = plan.order_RAGFT
    chk = torch.zeros(N, dtype=torch.bool, device=dev)
    chk[order] = True
    assert bool(chk.all())                                                       # a permutation
    del chk
    T, w = plan.forward(Cd)
    assert w[0].item() == float(N)
    e_in = torch.stack([(Cd[:, c].double() ** 2).sum() for c in range(D)])
    e_out = torch.stack([(T[:, c].double() ** 2).sum() for c in range(D)])
    assert torch.allclose(e_in, e_out, rtol=1e-5)                                # Parseval (encode_3dgs.py:183-184)
    dc = torch.stack([Cd[:, c].double().sum() for c in range(D)]) / np.sqrt(N)
    assert torch.allclose(T[0].double(), dc, rtol=1e-4, atol=1e-4 * dc.abs().max().item() + 1e-3)   # utils.py:46-57
    Crec = plan.inverse(T)
    cmax = Cd.abs().max().item()
    assert (Crec - Cd).abs().max().item() <= 1e-5 * cmax                         # encode_3dgs.py:186-195
    del Crec
    Q = plan.quant_reorder(T, 0.01)
    assert torch.equal(plan.forward_quant(Cd, 0.01), Q)                          # fused == two-call, bit for bit
    Td = plan.dequant_unreorder(Q, 0.01)
    assert (Td - T).abs().max().item() <= 0.005 * 1.0001 + 1e-6 * T.abs().max().item()
    del T
    Cq = plan.dequant_inverse(Q, 0.01)
    assert torch.equal(Cq, plan.inverse(Td))
