#!/usr/bin/env python3
"""The segmented RLGR coder on the nine quantization steps of a 3 M x 56 frame: one call per step against ONE set of launches for
all steps (raht_rlgr_seg_encode_batch / _decode_batch). Prints ms per step (= per 168 M symbols).
   python tools/time_rlgr_batch.py [reps] [seg_len]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import rlgr, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = [0.01, 0.04, 0.08, 0.12, 0.16, 0.20, 0.24, 0.32, 0.64]
V, keys, Ch = synth.scene(3_000_000, 12, 56, 2)
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 36)
Qs = p.forward_quant_multi(torch.from_numpy(Ch).cuda(), steps)            # (N, 56) row-major each
N, D = Qs[0].shape
k = len(steps)
coders = [rlgr.SegmentedCoder(N, D, S) for _ in steps]
SC = rlgr.SegmentedCoder


def wall(fn):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


outs = [torch.empty((D, N), dtype=torch.int32, device="cuda") for _ in steps]
e1 = wall(lambda: [c.encode(q) for c, q in zip(coders, Qs)])
ref = [c.container() for c in coders[:2]]
d1 = wall(lambda: [c.decode(out=o) for c, o in zip(coders, outs)])
eb = wall(lambda: SC.encode_batch(coders, Qs))
assert [c.container() for c in coders[:2]] == ref
db = wall(lambda: SC.decode_batch(coders, outs=outs))
for o, q in zip(outs, Qs):
    assert torch.equal(o.t(), q)
del outs
outs_rm = [torch.empty((N, D), dtype=torch.int32, device="cuda") for _ in steps]
d1_rm = wall(lambda: [c.decode(out=o) for c, o in zip(coders, outs_rm)])
db_rm = wall(lambda: SC.decode_batch(coders, outs=outs_rm))
for o, q in zip(outs_rm, Qs):
    assert torch.equal(o, q)
print(json.dumps({"symbols_per_step": N * D, "steps": k, "seg_len": S, "lanes_per_step": coders[0].G,
                  "one_call_per_step": {"encode_ms_per_step": round(e1 / k, 3), "decode_ms_per_step": round(d1 / k, 3), "decode_row_major_ms_per_step": round(d1_rm / k, 3)},
                  "all_steps_one_launch": {"encode_ms_per_step": round(eb / k, 3), "decode_ms_per_step": round(db / k, 3), "decode_row_major_ms_per_step": round(db_rm / k, 3)},
                  "bytes": [c.size_bytes for c in coders]}))
