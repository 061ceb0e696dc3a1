#!/usr/bin/env python3
"""The nine quantization steps of one frame (python/encode_3dgs.py:28,199-217): nine fused forward passes against ONE pass that
quantizes nine times (raht_fwd_quant_multi), on a 3 M x 56 frame."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import synth  # noqa: E402

V, keys, Ch = synth.scene(3_000_000, 12, 56, 2)
Cd = torch.from_numpy(Ch).cuda()
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 36)
steps = [0.01 * s for s in (1, 4, 8, 12, 16, 20, 24, 32, 64)]


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


single = timed(lambda: [p.forward_quant(Cd, s) for s in steps])
multi = timed(lambda: p.forward_quant_multi(Cd, steps))
N, D = Cd.shape
print(json.dumps({"rows": N, "channels": D, "steps": len(steps), "nine_forward_passes_ms": round(single, 4), "one_pass_nine_quantizations_ms": round(multi, 4),
                  "bytes_moved_multi": 4 * N * D * (1 + len(steps)), "frac_of_peak_multi": round(4 * N * D * (1 + len(steps)) / (multi * 1e-3) / 8e12, 4)}))
