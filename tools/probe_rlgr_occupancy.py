#!/usr/bin/env python3
"""How the segmented RLGR kernels scale with the number of independent streams in ONE launch: the same 3 M x 56 frame of quantized
coefficients stacked 1, 2, 3 times along the rows (2048 symbols per segment: 1.25, 2.5, 3.75 waves per SIMD). Prints ms per
168 M symbols -- the case for coding the quantization steps of a frame together."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import rlgr, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
V, keys, Ch = synth.scene(3_000_000, 12, 56, 2)
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 36)
Q1 = p.forward_quant(torch.from_numpy(Ch).cuda(), 0.04)                    # (N, 56) row-major
out = {}
for f in (1, 2, 3):
    Q = Q1.repeat(f, 1).contiguous()
    N, D = Q.shape
    sc = rlgr.SegmentedCoder(N, D, 2048)
    sc.encode(Q); sc.decode()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        sc.encode(Q)
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        back = sc.decode()
    torch.cuda.synchronize(); td = (time.perf_counter() - t) / reps
    out[f"x{f}"] = {"encode_ms_per_frame": round(te * 1e3 / f, 3), "decode_ms_per_frame": round(td * 1e3 / f, 3), "lanes": sc.G}
    del sc, Q, back
print(json.dumps(out))
