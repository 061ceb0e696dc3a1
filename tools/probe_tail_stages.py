# GPU probe: duration of every stage of the fused schedule in isolation, with the profiling ablations
# (1 = no butterflies, 2 = no merge resolution either) -- where do the latency-bound tail stages spend it?
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import _lib, synth
L = _lib.lib()
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
kd = torch.from_numpy(keys.view(np.int64)).cuda()
Cd = torch.from_numpy(Ch).cuda()
plan = R.RahtPlan.from_keys(kd, 3 * J)
plan.prepare(D)
Q = plan.forward_quant(Cd, 0.01)          # fills the workspaces with real survivors
Rc = plan.dequant_inverse(Q, 0.01)
vp = C.c_void_p
s = vp(torch.cuda.current_stream().cuda_stream)
st = plan.stage_stats(4, D)
print(st)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for k in range(len(st["rows_per_stage"])):
    for inv in (0, 1):
        row = []
        for ab in (0, 1, 2):
            if inv == 0:
                f = lambda: _lib.check(L.raht_debug_run_stage(plan._h, 0, k, vp(Cd.data_ptr()), D, D, None, 0, vp(Q.data_ptr()), D, C.c_float(0.01), ab, s))
            else:
                f = lambda: _lib.check(L.raht_debug_run_stage(plan._h, 1, k, None, 0, D, vp(Rc.data_ptr()), D, vp(Q.data_ptr()), D, C.c_float(0.01), ab, s))
            row.append(timeit(f))
        print("stage %d (%8d rows) %s : full %.1f us   no rounds %.1f us   no resolution %.1f us" % (k, st["rows_per_stage"][k], "inv" if inv else "fwd", *row))
