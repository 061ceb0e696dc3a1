# GPU probe: which input->output distances (multiples of 2 MiB) slow the stage-0 kernel down?
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import _lib, synth
L = _lib.lib()
n, J, D, seed = synth.CONFIGS["cfg3"]
if len(sys.argv) > 1: n = int(sys.argv[1])
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
dev = torch.device("cuda")
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
plan.prepare(D)
vp = C.c_void_p
nbytes = N * D * 4
span = (nbytes + (1 << 21) - 1) >> 21 << 21
K = 60
pool = torch.empty(span + (K + 2) * (2 << 20) + nbytes, dtype=torch.uint8, device=dev)
base = pool.data_ptr()
base_al = (base + (1 << 21) - 1) >> 21 << 21
def view(off, dtype):
    o = base_al - base + off
    return pool[o:o + nbytes].view(dtype).view(N, D)
Cd = view(0, torch.float32); Cd.copy_(torch.from_numpy(Ch).to(dev))
s = vp(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("span = %d x 2 MiB, base %#x" % (span >> 21, base_al))
bad = []
for k in list(range((span >> 21) - 12, (span >> 21) + K)):
    if (k << 21) < nbytes: continue
    T = view(k << 21, torch.float32)
    tp = timeit(lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, 0, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, D, C.c_float(0.01), 0, s)))
    bad.append((k - (span >> 21), round(tp, 1)))
print("(distance - span) / 2 MiB : us ->", bad)
