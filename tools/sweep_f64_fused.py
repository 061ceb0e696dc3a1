#!/usr/bin/env python3
"""float64 fused kernels (raht_fwd_quant_f64 / raht_dequant_inv_f64) on cfg3 over the rows per tile: LDS per workgroup decides how
many 512-thread workgroups share a CU (160 KiB in 128 granules of 1280 bytes).   python tools/sweep_f64_fused.py [R ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import synth  # noqa: E402

rows = [int(x) for x in sys.argv[1:]] or [0, 48, 56, 64, 72, 80, 88, 96, 104, 120, 128]
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
N = int(kd.shape[0])
C = torch.from_numpy(Ch).to(torch.float64).cuda()
alg = 2 * (N * D * 8 + N * D * 4 + 8 * N)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps


ref = None
for r in rows:
    p = R.RahtPlan.from_keys(kd, 3 * J)
    if r:
        p.set_engine("tile", r, r, 0, 0)
    Q = p.forward_quant(C, 0.01)
    if ref is None:
        ref = Q.clone()
    assert torch.equal(Q, ref)
    tf = timed(lambda: p.forward_quant(C, 0.01))
    ti = timed(lambda: p.dequant_inverse(Q, 0.01, dtype=torch.float64))
    st = p.stage_stats(8, D)
    print(json.dumps({"tile_rows": r or "default", "fwd_ms": round(tf, 4), "inv_ms": round(ti, 4), "step_ms": round(tf + ti, 4),
                      "frac_of_peak": round(alg / ((tf + ti) * 1e-3) / 8e12, 4), "stages": st.get("stage_entries", st)}))
    del p
