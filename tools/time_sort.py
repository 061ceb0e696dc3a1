# GPU timing of raht_sort_keys on the cfg3 key set (36-bit and 60-bit), for rocprofv3 --kernel-trace --stats
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
n, J, D, seed = synth.CONFIGS["cfg3"]
keys = synth.sorted_unique_keys(n, J, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
ku = kd[torch.randperm(kd.shape[0], device="cuda", generator=g)].contiguous()
for nb in (36, 36, 36, 60):
    kk = ku if nb == 36 else ((ku << 24) | (ku & ((1 << 24) - 1)))
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        ko, idx = R.sort_keys(kk, nbits=nb)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    assert bool((ko[1:] >= ko[:-1]).all())
    print("sort %d bit: %.3f ms" % (nb, best * 1e3))
