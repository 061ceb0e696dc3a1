#!/usr/bin/env python3
"""Does re-allocating the output buffer move the fused forward between its two speeds (buffer placement)? cfg3, the
float32 and the mixed fused forward, twelve freshly allocated Q buffers (the earlier ones are kept, so every one is new memory)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import synth  # noqa: E402

n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
Cd = torch.from_numpy(Ch).cuda()
p = R.RahtPlan.from_keys(kd, 3 * J)


def timed(fn, reps=40):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


keep, out = [], {"f32_fwd_ms": [], "mixed_fwd_ms": [], "f32_inv_ms": [], "q_minus_c_MiB": []}
import ctypes as C
from raht_3dgs_codec_amd import _lib
L = _lib.lib()
vp = C.c_void_p
st32 = (C.c_float * 1)(0.01)
st64 = (C.c_double * 1)(0.01)
for t in range(12):
    Q = torch.empty((Cd.shape[0], D), dtype=torch.int32, device="cuda")
    Cr = torch.empty_like(Cd)
    keep += [Q, Cr]
    s = vp(torch.cuda.current_stream().cuda_stream)
    out["q_minus_c_MiB"].append(round((Q.data_ptr() - Cd.data_ptr()) / 2 ** 20, 1))
    out["f32_fwd_ms"].append(timed(lambda: _lib.check(L.raht_fwd_quant(p._h, vp(Cd.data_ptr()), D, D, st32, 1, vp(Q.data_ptr()), D, s))))
    out["mixed_fwd_ms"].append(timed(lambda: _lib.check(L.raht_fwd_quant_mixed(p._h, vp(Cd.data_ptr()), D, D, st64, 1, 3, vp(Q.data_ptr()), D, s))))
    out["f32_inv_ms"].append(timed(lambda: _lib.check(L.raht_dequant_inv(p._h, vp(Q.data_ptr()), D, D, st32, 1, vp(Cr.data_ptr()), D, s))))
    out.setdefault("mixed_inv_ms", []).append(timed(lambda: _lib.check(L.raht_dequant_inv_mixed(p._h, vp(Q.data_ptr()), D, D, st64, 1, 3, vp(Cr.data_ptr()), D, s))))
    out.setdefault("mixed_step_ms", []).append(timed(lambda: (_lib.check(L.raht_fwd_quant_mixed(p._h, vp(Cd.data_ptr()), D, D, st64, 1, 3, vp(Q.data_ptr()), D, s)),
                                                              _lib.check(L.raht_dequant_inv_mixed(p._h, vp(Q.data_ptr()), D, D, st64, 1, 3, vp(Cr.data_ptr()), D, s))), 100))
    out.setdefault("cr_minus_q_MiB", []).append(round((Cr.data_ptr() - Q.data_ptr()) / 2 ** 20, 1))
print(json.dumps(out))
