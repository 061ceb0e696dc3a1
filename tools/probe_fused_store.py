# GPU probe: where does the fused stage-0 kernel lose time against the plain one?
#   plain bulk      : T rows written as one 16-byte-vector span per tile
#   plain row-wise  : same destinations, but row-granular dword stores (ld = D + 1 disables the span path)
#   fused           : quantize + scatter to Q[inv_order[row]]
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
Cd = torch.from_numpy(Ch).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
plan.prepare(D)
vp = C.c_void_p
T1 = torch.empty((N, D), dtype=torch.float32, device=dev)
T2 = torch.empty((N, D + 1), dtype=torch.float32, device=dev)
C2 = torch.empty((N, D + 1), dtype=torch.float32, device=dev); C2[:, :D] = Cd
Q = torch.empty((N, D), dtype=torch.int32, device=dev)
s = vp(torch.cuda.current_stream().cuda_stream)
def stage(inv, mat, ldm, mat2, ld2, q, ablate=0):
    _lib.check(L.raht_debug_run_stage(plan._h, inv, 0, vp(mat.data_ptr()) if mat is not None else None, ldm, D,
                                      vp(mat2.data_ptr()) if mat2 is not None else None, ld2,
                                      vp(q.data_ptr()) if q is not None else None, D, C.c_float(0.01), ablate, s))
def timeit(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("fwd plain  bulk     : %.1f us" % timeit(lambda: stage(0, Cd, D, T1, D, None)))
print("fwd plain  row-wise : %.1f us" % timeit(lambda: stage(0, Cd, D, T2, D + 1, None)))
print("fwd plain  src ld+1 : %.1f us" % timeit(lambda: stage(0, C2, D + 1, T1, D, None)))
print("fwd fused           : %.1f us" % timeit(lambda: stage(0, Cd, D, None, 0, Q)))
for ab in (1, 2, 3):
    print("fwd fused ablate=%d  : %.1f us" % (ab, timeit(lambda: stage(0, Cd, D, None, 0, Q, ab))))
print("inv plain  bulk     : %.1f us" % timeit(lambda: stage(1, T1, D, Cd, D, None)))
print("inv plain  row-wise : %.1f us" % timeit(lambda: stage(1, T2, D + 1, Cd, D, None)))
print("inv fused           : %.1f us" % timeit(lambda: stage(1, None, 0, Cd, D, Q)))
for ab in (1, 2, 3):
    print("inv fused ablate=%d  : %.1f us" % (ab, timeit(lambda: stage(1, None, 0, Cd, D, Q, ab))))
