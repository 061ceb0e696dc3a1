# GPU probe: cost of row-granular access patterns (run on the GPU box)
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
Cd = torch.from_numpy(Ch).to(dev)
p = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).to(dev), 3 * J)
h = p._h; vp = C.c_void_p
s_ = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
T = torch.empty_like(Cd); Crec = torch.empty_like(Cd)
Tp = torch.zeros((N, 64), dtype=torch.float32, device=dev); Cp = torch.zeros((N, 64), dtype=torch.float32, device=dev)
Q = torch.empty((N, D), dtype=torch.int32, device=dev); Qp = torch.zeros((N, 64), dtype=torch.int32, device=dev)
st = (C.c_float * 1)(0.01)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
chk = _lib.check
r = {}
r["fwd contiguous"] = timed(lambda: chk(L.raht_fwd(h, vp(Cd.data_ptr()), D, D, vp(T.data_ptr()), D, None, s_())))
r["inv contiguous"] = timed(lambda: chk(L.raht_inv(h, vp(T.data_ptr()), D, D, vp(Crec.data_ptr()), D, s_())))
Cp[:, :D] = Cd
r["fwd in ld=64"] = timed(lambda: chk(L.raht_fwd(h, vp(Cp.data_ptr()), 64, D, vp(T.data_ptr()), D, None, s_())))
r["fwd out ld=64"] = timed(lambda: chk(L.raht_fwd(h, vp(Cd.data_ptr()), D, D, vp(Tp.data_ptr()), 64, None, s_())))
r["inv in ld=64 (row-granular identity gather)"] = timed(lambda: chk(L.raht_inv(h, vp(Tp.data_ptr()), 64, D, vp(Crec.data_ptr()), D, s_())))
r["inv out ld=64"] = timed(lambda: chk(L.raht_inv(h, vp(T.data_ptr()), D, D, vp(Cp.data_ptr()), 64, s_())))
r["fwd_quant ldq=59"] = timed(lambda: chk(L.raht_fwd_quant(h, vp(Cd.data_ptr()), D, D, st, 1, vp(Q.data_ptr()), D, s_())))
r["fwd_quant ldq=64"] = timed(lambda: chk(L.raht_fwd_quant(h, vp(Cd.data_ptr()), D, D, st, 1, vp(Qp.data_ptr()), 64, s_())))
r["dequant_inv ldq=59"] = timed(lambda: chk(L.raht_dequant_inv(h, vp(Q.data_ptr()), D, D, st, 1, vp(Crec.data_ptr()), D, s_())))
r["dequant_inv ldq=64"] = timed(lambda: chk(L.raht_dequant_inv(h, vp(Qp.data_ptr()), 64, D, st, 1, vp(Crec.data_ptr()), D, s_())))
x = torch.empty_like(Cd)
r["torch copy N*D f32 (r+w)"] = timed(lambda: x.copy_(Cd))
idx = p.order_RAGFT
r["torch index_select rows by order"] = timed(lambda: torch.index_select(Cd, 0, idx, out=x))
for k, v in r.items(): print(f"{k:50s} {v:.4f} ms   {8.0*N*D/v/1e6:.0f} GB/s-equivalent")
