"""Duration of every single fused step after the process has been idle: is there a start-up transient?"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
Cd = torch.from_numpy(Ch).to(dev)
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
Q = torch.empty((N, D), dtype=torch.int32, device=dev)
Crec = torch.empty_like(Cd)
steps = (C.c_float * 1)(0.01)
vp = C.c_void_p
h = plan._h


def step():
    s = vp(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.raht_fwd_quant(h, vp(Cd.data_ptr()), D, D, steps, 1, vp(Q.data_ptr()), D, s))
    _lib.check(L.raht_dequant_inv(h, vp(Q.data_ptr()), D, D, steps, 1, vp(Crec.data_ptr()), D, s))


step(); torch.cuda.synchronize()
for idle in (0.0, 0.5):
    time.sleep(idle)
    K = 120
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for k in range(K):
        step(); ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)]
    print(f"after {idle} s idle: " + " ".join(f"{x:.3f}" for x in ms[:12]) + " ... steps 20-29 mean %.4f, 50-59 mean %.4f, 100-119 mean %.4f" % (np.mean(ms[20:30]), np.mean(ms[50:60]), np.mean(ms[100:120])))
