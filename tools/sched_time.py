# GPU timing: plan creation (arrays + default schedule), prepare and destroy on cfg3
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
for i in range(4):
    torch.cuda.synchronize(); t=time.perf_counter()
    p = R.RahtPlan.from_keys(kd, 3*J)
    torch.cuda.synchronize(); t1=time.perf_counter()
    p.prepare(D)
    torch.cuda.synchronize(); t2=time.perf_counter()
    st = p.stage_stats(4, D)
    del p
    torch.cuda.synchronize(); t3=time.perf_counter()
    print("create %.3f ms  prepare %.3f ms  destroy %.3f ms" % ((t1-t)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), st["rows_per_stage"])
