"""Does replaying the fused codec step as a hipGraph beat launching its 8 kernels one by one?
(cfg3; the plan is prepared, so the entry points only enqueue kernels and can be captured.)"""
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
n, J, D, seed = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
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
_lib.check(L.raht_plan_prepare(h, 4, D, None))


def step():
    s = vp(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.raht_fwd_quant(h, vp(Cd.data_ptr()), D, D, steps, 1, vp(Q.data_ptr()), D, s))
    _lib.check(L.raht_dequant_inv(h, vp(Q.data_ptr()), D, D, steps, 1, vp(Crec.data_ptr()), D, s))


def wall(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step(); step()
torch.cuda.synchronize()
direct = wall(step, 50)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    step()
torch.cuda.synchronize()
ref = Crec.clone()
Crec.zero_()
g.replay(); torch.cuda.synchronize()
assert torch.equal(ref, Crec), "graph replay differs"
graph = wall(g.replay, 50)
direct2 = wall(step, 50)
print(f"direct {direct:.4f} ms  graph {graph:.4f} ms  direct again {direct2:.4f} ms  ({N} rows)")
