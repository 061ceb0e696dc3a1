# GPU probe: do the tail stages of one half-scene overlap with the big kernel of the other half when
# the two halves run on two streams?  (run on the GPU box)
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import _lib, synth
L = _lib.lib()
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
dev = torch.device("cuda")
h = N // 2
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
Cd = torch.from_numpy(Ch).to(dev)
full = R.RahtPlan.from_keys(kd, 3 * J)
pa = R.RahtPlan.from_keys(kd[:h].contiguous(), 3 * J)
pb = R.RahtPlan.from_keys(kd[h:].contiguous(), 3 * J)
Ca, Cb = Cd[:h].contiguous(), Cd[h:].contiguous()
Q = torch.empty((N, D), dtype=torch.int32, device=dev)
Qa, Qb = torch.empty((h, D), dtype=torch.int32, device=dev), torch.empty((N - h, D), dtype=torch.int32, device=dev)
Ra, Rb, Rf = torch.empty_like(Ca), torch.empty_like(Cb), torch.empty_like(Cd)
st = (C.c_float * 1)(0.01)
vp = C.c_void_p
for p in (full, pa, pb):
    p.prepare(D)
def run(p, Cx, Qx, Rx, stream):
    s = vp(stream.cuda_stream)
    _lib.check(L.raht_fwd_quant(p._h, vp(Cx.data_ptr()), D, D, st, 1, vp(Qx.data_ptr()), D, s))
    _lib.check(L.raht_dequant_inv(p._h, vp(Qx.data_ptr()), D, D, st, 1, vp(Rx.data_ptr()), D, s))
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
print("full scene, one stream          : %.4f ms" % timeit(lambda: run(full, Cd, Q, Rf, s0)))
print("two halves, one stream (serial) : %.4f ms" % timeit(lambda: (run(pa, Ca, Qa, Ra, s0), run(pb, Cb, Qb, Rb, s0))))
def conc():
    run(pa, Ca, Qa, Ra, s1); run(pb, Cb, Qb, Rb, s2)
print("two halves, two streams         : %.4f ms" % timeit(conc))
def conc_staggered():
    # half B starts its forward while half A is already in its tails: forward A, then B, inverse in opposite order
    sa, sb = vp(s1.cuda_stream), vp(s2.cuda_stream)
    _lib.check(L.raht_fwd_quant(pa._h, vp(Ca.data_ptr()), D, D, st, 1, vp(Qa.data_ptr()), D, sa))
    _lib.check(L.raht_fwd_quant(pb._h, vp(Cb.data_ptr()), D, D, st, 1, vp(Qb.data_ptr()), D, sb))
    _lib.check(L.raht_dequant_inv(pa._h, vp(Qa.data_ptr()), D, D, st, 1, vp(Ra.data_ptr()), D, sa))
    _lib.check(L.raht_dequant_inv(pb._h, vp(Qb.data_ptr()), D, D, st, 1, vp(Rb.data_ptr()), D, sb))
print("two halves, two streams (interleaved issue): %.4f ms" % timeit(conc_staggered))
