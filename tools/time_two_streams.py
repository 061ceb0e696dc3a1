#!/usr/bin/env python3
"""The drivers' loop over quantization steps (python/encode_3dgs.py:199-275) with the two directions on two streams: forward +
quantize of step s + 1 next to dequantize + inverse of step s (raht_plan_set_concurrent_directions), against the same calls
back to back on one stream -- bench.py's two_stream_loop (which probes its side stream: two streams that share a hardware queue
do not overlap). Scenes: cfg3 (3 M x 59), the reference's own shape (J = 10, ~1 M x 56), cfg2 (1 M x 14)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402

L = _lib.lib()
out = {}
for name, (n, J, D, seed) in (("cfg3", synth.CONFIGS["cfg3"]), ("reference_shape_d56", (1_000_000, 10, 56, 7)), ("cfg2", synth.CONFIGS["cfg2"])):
    V, keys, Ch = synth.scene(n, J, D, seed)
    kd = torch.from_numpy(keys.view(np.int64)).cuda()
    out[name] = bench.two_stream_loop(R, L, _lib, kd, torch.from_numpy(Ch).cuda(), 3 * J, 0.01)
print(json.dumps(out))
