#!/usr/bin/env python3
"""The drivers' loop over quantization steps (python/encode_3dgs.py:199-275) with the two directions on two streams: forward +
quantize of step s + 1 next to dequantize + inverse of step s (raht_plan_set_concurrent_directions), against the same calls
back to back on one stream. Scenes: cfg3 (3 M x 59) and the reference's own shape (J = 10, ~1 M x 56)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402


def run(n, J, D, seed, reps=200):
    L = _lib.lib()
    V, keys, Ch = synth.scene(n, J, D, seed)
    Cd = torch.from_numpy(Ch).cuda()
    p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 3 * J)
    N = Cd.shape[0]
    Q = [torch.empty((N, D), dtype=torch.int32, device="cuda") for _ in range(2)]
    Cr = torch.empty_like(Cd)
    st = (C.c_float * 1)(0.01)
    vp = C.c_void_p
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def fwd(q, s):
        _lib.check(L.raht_fwd_quant(p._h, vp(Cd.data_ptr()), D, D, st, 1, vp(q.data_ptr()), D, vp(s.cuda_stream)))

    def inv(q, s):
        _lib.check(L.raht_dequant_inv(p._h, vp(q.data_ptr()), D, D, st, 1, vp(Cr.data_ptr()), D, vp(s.cuda_stream)))

    def timed(body, reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sa)
        body(reps)
        sa.wait_stream(sb)
        e1.record(sa); e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def serial(reps):
        for i in range(reps):
            fwd(Q[0], sa); inv(Q[0], sa)

    def overlapped(reps):
        # step i: forward into Q[i % 2] on stream A; inverse of Q[(i - 1) % 2] on stream B once that forward is done
        ev_f = [torch.cuda.Event() for _ in range(2)]
        ev_i = [torch.cuda.Event() for _ in range(2)]
        fwd(Q[0], sa); ev_f[0].record(sa)
        for i in range(1, reps + 1):
            if i >= 2:
                sa.wait_event(ev_i[i % 2])                 # the inverse that read this Q buffer two steps ago
            if i < reps:
                fwd(Q[i % 2], sa); ev_f[i % 2].record(sa)
            sb.wait_event(ev_f[(i - 1) % 2])
            inv(Q[(i - 1) % 2], sb); ev_i[(i - 1) % 2].record(sb)
    p.set_concurrent_directions(False)
    for _ in range(2):
        serial(20)
    t_serial = timed(serial, reps)
    ref = Cr.clone()
    p.set_concurrent_directions(True)
    overlapped(20)
    trials = sorted(timed(overlapped, reps) for _ in range(5))       # (two modes on small scenes: see bench.py two_stream_loop)
    t_two = trials[2]
    torch.cuda.synchronize()
    assert torch.equal(Cr, ref), "two-stream loop reconstructs differently"
    alg = 2 * (8.0 * N * D + 8.0 * N)
    return {"rows": N, "channels": D, "J": J, "one_stream_ms_per_step": round(t_serial, 4), "two_streams_ms_per_step": round(t_two, 4), "two_streams_trials_ms": [round(t, 4) for t in trials],
            "one_stream_frac_of_peak": round(alg / (t_serial * 1e-3) / 8e12, 4), "two_streams_frac_of_peak": round(alg / (t_two * 1e-3) / 8e12, 4),
            "bit_identical": True}


if __name__ == "__main__":
    out = {"cfg3": run(*synth.CONFIGS["cfg3"]), "reference_shape_d56": run(1_000_000, 10, 56, 1), "cfg2": run(*synth.CONFIGS["cfg2"])}
    print(json.dumps(out))
