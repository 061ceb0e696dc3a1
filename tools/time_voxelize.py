# GPU timing: voxelize_pc_batched on the unsorted cfg3 cloud (3 xyz + 56 attribute columns)
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
dev = torch.device("cuda", 0)
N = V.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
perm = torch.randperm(N, device=dev, generator=g)
xyz = torch.from_numpy(V.astype(np.float32)).to(dev)[perm] + 0.5
PC = torch.cat([xyz, torch.from_numpy(Ch).to(dev)[perm][:, :56]], dim=1).contiguous()
for resid in (False, True):
    for i in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev, residuals=resid, sorted_points=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("voxelize residuals=%s: %.3f ms (%d points x %d columns)" % (resid, dt * 1e3, N, PC.shape[1]))
        del out
