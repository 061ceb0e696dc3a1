#!/usr/bin/env python3
"""For rocprofv3 --pmc passes (tools/pmc_rlgr.sh <tag> tools/rlgr_loop_batch_rm.py): the nine steps of a 3 M x 56 frame through the
batched encoder and the symbol-synchronous (row-major) batched decoder only, a few passes each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import rlgr, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = [0.01, 0.04, 0.08, 0.12, 0.16, 0.20, 0.24, 0.32, 0.64]
V, keys, Ch = synth.scene(3_000_000, 12, 56, 2)
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 36)
Qs = p.forward_quant_multi(torch.from_numpy(Ch).cuda(), steps)
N, D = Qs[0].shape
coders = [rlgr.SegmentedCoder(N, D, 2048) for _ in steps]
outs = [torch.empty((N, D), dtype=torch.int32, device="cuda") for _ in steps]
for _ in range(reps):
    rlgr.SegmentedCoder.encode_batch(coders, Qs)
    rlgr.SegmentedCoder.decode_batch(coders, outs=outs)
torch.cuda.synchronize()
assert all(torch.equal(o, q) for o, q in zip(outs, Qs))
print("ok")
