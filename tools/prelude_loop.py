#!/usr/bin/env python3
"""The prelude stages in loops, for `rocprofv3 --kernel-trace --stats` (tools/profile_prelude.sh): plan build from sorted keys
(x30), 36-bit key sort (x30), voxelizer without / with residuals (x10) on the cfg3 scene. Prints wall times per call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import synth  # noqa: E402

what = set(sys.argv[1:]) or {"plan", "sort", "vox"}
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
dev = torch.device("cuda")
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
N = int(kd.shape[0])


def wall(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


if "plan" in what:
    print("plan_from_sorted_keys ms", round(wall(lambda: R.RahtPlan.from_keys(kd, 3 * J), 30), 4))
    print("plan_from_sorted_keys_borrowed ms", round(wall(lambda: R.RahtPlan.from_keys(kd, 3 * J, borrow=True), 30), 4))
g = torch.Generator(device=dev); g.manual_seed(1)
perm = torch.randperm(N, device=dev, generator=g)
ku = kd[perm].contiguous()
if "sort" in what:
    print("radix_sort_36bit ms", round(wall(lambda: R.sort_keys(ku, nbits=3 * J), 30), 4))
if "vox" in what:
    xyz = torch.from_numpy(V.astype(np.float32)).to(dev)[perm] + 0.5
    PC = torch.cat([xyz, torch.from_numpy(Ch).to(dev)[perm][:, :56]], dim=1).contiguous()
    print("voxelize ms", round(wall(lambda: R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev, residuals=False, sorted_points=False), 10), 4))
    print("voxelize_with_residuals ms", round(wall(lambda: R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev), 10), 4))
    print("voxelize_plan ms", round(wall(lambda: R.voxelize_plan(PC, [0.0, 0.0, 0.0], float(2 ** J), J, device=dev), 20), 4))
