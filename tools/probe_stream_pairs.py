#!/usr/bin/env python3
"""Does the two-stream step loop's gain depend on WHICH two streams carry it? The reference's shape (J = 10, 1 M x 56), one main
stream against each of seven side streams, twice. (HIP multiplexes streams onto a few hardware queues: two streams that share
one cannot overlap.)   python tools/probe_stream_pairs.py"""
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
V, keys, Ch = synth.scene(1_000_000, 10, 56, 7)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
Cd = torch.from_numpy(Ch).cuda()
pool = [torch.cuda.Stream() for _ in range(8)]
out = {"env_GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
for rnd in range(2):
    for i in range(1, 8):
        r = bench.two_stream_loop(R, L, _lib, kd, Cd, 30, 0.01, reps=100, streams=(pool[0], pool[i]))
        out[f"round{rnd}_main0_side{i}"] = [r["one_stream_ms_per_step"], r["two_streams_ms_per_step"]]
print(json.dumps(out))
