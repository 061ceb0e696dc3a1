# GPU timing: the sharded driver's per-step overhead without communication (world = 1): truncated
# local plan + root buffer + replicated top tree + root quantization vs the plain fused step
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth, sharded
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
Cd = torch.from_numpy(Ch).cuda()
plan = R.RahtPlan.from_keys(kd, 3 * J)
sh = sharded.ShardedRaht(kd, 3 * J, prefix_bits=9)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("plain fused step    %.4f ms" % timeit(lambda: plan.dequant_inverse(plan.forward_quant(Cd, 0.01), 0.01)))
print("sharded step (w=1)  %.4f ms  (%d roots)" % (timeit(lambda: sh.step(Cd, 0.01)), sh.n_roots))
