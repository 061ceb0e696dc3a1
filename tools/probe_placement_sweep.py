# GPU probe: stage-0 kernel duration against the distance between its input and output buffers
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import _lib, synth
L = _lib.lib()
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
dev = torch.device("cuda")
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
plan.prepare(D)
vp = C.c_void_p
nbytes = N * D * 4
span = (nbytes + (1 << 21) - 1) >> 21 << 21
pool = torch.empty(2 * span + (96 << 20), dtype=torch.uint8, device=dev)
base = pool.data_ptr()
base_al = (base + (1 << 21) - 1) >> 21 << 21
def view(off, dtype):
    o = base_al - base + off
    return pool[o:o + nbytes].view(dtype).view(N, D)
Cd = view(0, torch.float32); Cd.copy_(torch.from_numpy(Ch).to(dev))
s = vp(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=8):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("span", span, "pool %#x" % base)
deltas = [k * 65536 for k in range(0, 64)] + [k << 21 for k in range(2, 40)]
for delta in deltas:
    T = view(span + delta, torch.float32)
    Q = view(span + delta, torch.int32)
    tp = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, D, C.c_float(0.01), 0, s)))
    tf = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(Cd.data_ptr()), D, D, None, 0, vp(Q.data_ptr()), D, C.c_float(0.01), 0, s)))
    print("delta %9d (%6.2f MiB): plain fwd %.1f us   fused fwd %.1f us" % (delta, delta / 2**20, tp, tf))
