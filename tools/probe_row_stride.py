#!/usr/bin/env python3
"""Does the row stride matter? The fused cfg3 step with the matrices' rows at 59 (contiguous: 236-byte rows, 16-byte chunks at
4-byte alignment), 60 (240 bytes: every chunk 16-byte aligned) and 64 floats (256 bytes: rows = two whole 128-byte lines),
for C / C_rec (ldc) and for Q (ldq) separately. The C ABI takes any stride >= D (include/raht.h)."""
import ctypes as C
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
dev = torch.device("cuda")
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
N = int(kd.shape[0])
plan = R.RahtPlan.from_keys(kd, 3 * J)
C0 = torch.from_numpy(Ch).to(dev)
vp = C.c_void_p
step = (C.c_float * 1)(0.01)


def timed(fn, reps=200):
    for _ in range(64):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


ref = None
for ldc, ldq in ((59, 59), (60, 60), (64, 64), (59, 60), (59, 64), (60, 59), (64, 59)):
    Cp = torch.zeros((N, ldc), dtype=torch.float32, device=dev); Cp[:, :D] = C0
    Q = torch.zeros((N, ldq), dtype=torch.int32, device=dev)
    Rr = torch.zeros((N, ldc), dtype=torch.float32, device=dev)
    s = vp(torch.cuda.current_stream().cuda_stream)

    def fq():
        _lib.check(L.raht_fwd_quant(plan._h, vp(Cp.data_ptr()), ldc, D, step, 1, vp(Q.data_ptr()), ldq, s))

    def di():
        _lib.check(L.raht_dequant_inv(plan._h, vp(Q.data_ptr()), ldq, D, step, 1, vp(Rr.data_ptr()), ldc, s))
    fq(); di(); torch.cuda.synchronize()
    if ref is None:
        ref = (Q[:, :D].clone(), Rr[:, :D].clone())
    else:
        assert torch.equal(Q[:, :D], ref[0]) and torch.equal(Rr[:, :D], ref[1])
    tf, ti = timed(fq), timed(di)
    print(f"ldc {ldc} ldq {ldq}: fwd_quant {tf:.4f} ms  dequant_inv {ti:.4f} ms  step {tf + ti:.4f} ms")
    del Cp, Q, Rr
