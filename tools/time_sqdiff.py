#!/usr/bin/env python3
"""The decode side of a codec step on a 3 M x 56 frame: dequantize + inverse, then the PSNR sums as a pass of their own
(raht_dequant_inv + raht_sqdiff_columns) against the fused raht_dequant_inv_sqdiff with and without writing C_rec."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402


def timed(fn, reps=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    L = _lib.lib()
    V, keys, Ch = synth.scene(3_000_000, 12, 56, 2)
    Cd = torch.from_numpy(Ch).cuda()
    p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 36)
    N, D = Cd.shape
    Q = p.forward_quant(Cd, 0.04)
    rec = torch.empty_like(Cd)
    ssd = torch.empty(D, dtype=torch.float64, device="cuda")
    st = (C.c_float * 1)(0.04)
    vp = C.c_void_p

    def s_():
        return vp(torch.cuda.current_stream().cuda_stream)

    def two():
        _lib.check(L.raht_dequant_inv(p._h, vp(Q.data_ptr()), D, D, st, 1, vp(rec.data_ptr()), D, s_()))
        _lib.check(L.raht_sqdiff_columns(vp(Cd.data_ptr()), D, vp(rec.data_ptr()), D, N, D, _lib.RAHT_F32, vp(ssd.data_ptr()), s_()))

    def inv_only():
        _lib.check(L.raht_dequant_inv(p._h, vp(Q.data_ptr()), D, D, st, 1, vp(rec.data_ptr()), D, s_()))

    def fused(keep):
        _lib.check(L.raht_dequant_inv_sqdiff(p._h, vp(Q.data_ptr()), D, D, st, 1, vp(Cd.data_ptr()), D, vp(rec.data_ptr()) if keep else None, D, vp(ssd.data_ptr()), s_()))
    out = {"rows": N, "channels": D, "dequant_inv_ms": round(timed(inv_only), 4), "dequant_inv_then_sqdiff_columns_ms": round(timed(two), 4),
           "dequant_inv_sqdiff_keep_rec_ms": round(timed(lambda: fused(True)), 4), "dequant_inv_sqdiff_no_rec_ms": round(timed(lambda: fused(False)), 4)}
    # algorithmic bytes: Q in + C in (+ C_rec out) + 8 bytes of plan per row; against the 8 TB/s peak
    for key, nmat in (("dequant_inv_sqdiff_keep_rec", 3), ("dequant_inv_sqdiff_no_rec", 2), ("dequant_inv", 2)):
        alg = nmat * 4.0 * N * D + 8.0 * N
        out[key + "_roofline"] = {"bound": "hbm", "alg_bytes": alg, "achieved_GBs": round(alg / (out[key + "_ms"] * 1e-3) / 1e9, 1),
                                  "frac_of_peak": round(alg / (out[key + "_ms"] * 1e-3) / 1e9 / 8000.0, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
