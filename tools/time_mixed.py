#!/usr/bin/env python3
"""Fused step on the cfg3 scene: float32 kernels, mixed-precision kernels (xyz columns in float64), float64 kernels.
   python tools/time_mixed.py [--reps 200] [--workload cfg3]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(max(10, reps // 4)):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--step", type=float, default=0.01)
    ap.add_argument("--wide", type=int, default=3)
    ap.add_argument("--tile-rows", type=int, default=0)
    a = ap.parse_args()
    L = _lib.lib()
    n, J, D, seed = synth.CONFIGS[a.workload]
    V, keys, Ch = synth.scene(n, J, D, seed)
    dev = torch.device("cuda")
    Cd = torch.from_numpy(Ch).to(dev)
    kd = torch.from_numpy(keys.view(np.int64)).to(dev)
    N = int(kd.shape[0])
    p = R.RahtPlan.from_keys(kd, 3 * J)
    if a.tile_rows:
        p.set_engine("tile", a.tile_rows, a.tile_rows, 0, 0)
    vp = C.c_void_p
    Q = torch.empty((N, D), dtype=torch.int32, device=dev)
    Cr = torch.empty_like(Cd)
    s32 = (C.c_float * 1)(a.step)
    s64 = (C.c_double * 1)(a.step)

    def st():
        return vp(torch.cuda.current_stream().cuda_stream)

    def f32_f():
        _lib.check(L.raht_fwd_quant(p._h, vp(Cd.data_ptr()), D, D, s32, 1, vp(Q.data_ptr()), D, st()))

    def f32_i():
        _lib.check(L.raht_dequant_inv(p._h, vp(Q.data_ptr()), D, D, s32, 1, vp(Cr.data_ptr()), D, st()))

    def mx_f():
        _lib.check(L.raht_fwd_quant_mixed(p._h, vp(Cd.data_ptr()), D, D, s64, 1, a.wide, vp(Q.data_ptr()), D, st()))

    def mx_i():
        _lib.check(L.raht_dequant_inv_mixed(p._h, vp(Q.data_ptr()), D, D, s64, 1, a.wide, vp(Cr.data_ptr()), D, st()))

    out = {"workload": a.workload, "N": N, "D": D, "mixed_stats": p.mixed_stats(D, a.wide), "f32_stats": p.stage_stats(4, D)}
    for rnd in range(2):
        for name, f, i in (("f32", f32_f, f32_i), ("mixed", mx_f, mx_i)):
            tf, ti = timed(f, a.reps), timed(i, a.reps)
            both = timed(lambda: (f(), i()), a.reps)
            out[f"{name}_{rnd}"] = {"fwd_ms": round(tf, 4), "inv_ms": round(ti, 4), "step_ms": round(both, 4), "MGs": round(N / both / 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
