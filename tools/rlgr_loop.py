#!/usr/bin/env python3
"""The segmented RLGR coder on a 3 M x 56 frame of quantized RAHT coefficients, a few encode / decode passes: for rocprofv3 (kernel
trace or --pmc passes, tools/pmc_rlgr.sh). Prints the pass times."""
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
Q = p.forward_quant(torch.from_numpy(Ch).cuda(), 0.04)                    # (N, 56) row-major
N, D = Q.shape
out = {}
for S in (2048, 1024):
    sc = rlgr.SegmentedCoder(N, D, S)
    sc.encode(Q); sc.decode()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        sc.encode(Q)
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        back = sc.decode()
    torch.cuda.synchronize(); td = (time.perf_counter() - t) / reps
    assert torch.equal(back.t(), Q)
    out[f"seg_{S}"] = {"encode_row_major_ms": round(te * 1e3, 3), "decode_channel_major_ms": round(td * 1e3, 3), "bytes": sc.size_bytes,
                       "symbols": N * D, "lanes": sc.G}
print(json.dumps(out))
