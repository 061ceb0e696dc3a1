#!/usr/bin/env python3
"""The fused forward (float32 and mixed) on cfg3 against the distance between its input C and its output Q, both carved from one
slab: Q behind C's end by 0 ... 2 GiB, and in front of C's start. (buffer placement.)"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402

L = _lib.lib()
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
N = int(kd.shape[0])
nb = N * D * 4
MiB = 1 << 20
slab = torch.empty(9 * 1024 * MiB, dtype=torch.uint8, device="cuda")
base = (-slab.data_ptr()) % (2 * MiB)
c_off = base + 4 * 1024 * MiB
Cd = slab[c_off: c_off + nb].view(torch.float32).view(N, D)
Cd.copy_(torch.from_numpy(Ch).cuda())
p = R.RahtPlan.from_keys(kd, 3 * J)
vp = C.c_void_p
st32, st64 = (C.c_float * 1)(0.01), (C.c_double * 1)(0.01)
s = vp(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=30):
    for _ in range(8):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


span = (nb + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
rows = []
gaps = [0, 2, 8, 32, 64, 128, 256, 512, 1024, 2048]
for side in (+1, -1):
    for g in gaps:
        q_off = c_off + span + g * MiB if side > 0 else c_off - span - g * MiB
        Q = slab[q_off: q_off + nb].view(torch.int32).view(N, D)
        f = timed(lambda: _lib.check(L.raht_fwd_quant(p._h, vp(Cd.data_ptr()), D, D, st32, 1, vp(Q.data_ptr()), D, s)))
        m = timed(lambda: _lib.check(L.raht_fwd_quant_mixed(p._h, vp(Cd.data_ptr()), D, D, st64, 1, 3, vp(Q.data_ptr()), D, s)))
        i = timed(lambda: _lib.check(L.raht_dequant_inv_mixed(p._h, vp(Q.data_ptr()), D, D, st64, 1, 3, vp(Cd.data_ptr() + 0), D, s))) if False else None
        rows.append({"q_after_c_end_MiB" if side > 0 else "q_end_before_c_MiB": g, "f32_fwd_ms": f, "mixed_fwd_ms": m})
print(json.dumps(rows))
