# GPU probe: does the stage-0 kernel's duration depend on where its input / output buffers sit?
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
pool = torch.empty(3 * span + (64 << 20), dtype=torch.uint8, device=dev)
base = pool.data_ptr()
base_al = (base + (1 << 21) - 1) >> 21 << 21
print("N=%d bytes=%d span=%d pool=%#x" % (N, nbytes, span, base))
def view(off, dtype):
    o = base_al - base + off
    return pool[o:o + nbytes].view(dtype).view(N, D)
Cd = view(0, torch.float32); Cd.copy_(torch.from_numpy(Ch).to(dev))
s = vp(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for delta in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096 + 256, 5 << 20, (13 << 20) + 12288 + 768):
    T = view(span + delta, torch.float32)
    Q = view(span + delta, torch.int32)
    tp = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, D, C.c_float(0.01), 0, s)))
    tf = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(Cd.data_ptr()), D, D, None, 0, vp(Q.data_ptr()), D, C.c_float(0.01), 0, s)))
    ti = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 1, 0, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, D, C.c_float(0.01), 0, s)))
    tc = timeit(lambda: T.copy_(Cd))
    print("T - C = span + %9d : fwd plain %.1f us  fwd fused %.1f us  inv plain %.1f us  torch copy %.1f us" % (delta, tp, tf, ti, tc))
# separate torch allocations, as bench.py does
C3 = torch.from_numpy(Ch).to(dev); T3 = torch.empty_like(C3)
print("separate allocs C=%#x T=%#x diff=%d" % (C3.data_ptr(), T3.data_ptr(), T3.data_ptr() - C3.data_ptr()))
print("  fwd plain %.1f us" % timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(C3.data_ptr()), D, D, vp(T3.data_ptr()), D, None, D, C.c_float(0.01), 0, s))))
