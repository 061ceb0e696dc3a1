#!/usr/bin/env python3
"""3 M Gaussians x (11 + 48) columns -> voxelized frame: voxelizer, then the merge kernel on five split arrays (the reference's two
steps, python/test_voxelize_3dgs.py:203-257) against raht_voxelize_merge (one call, one pass over the rows)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raht_3dgs_codec_amd import merge, ops, synth  # noqa: E402

N, J, cd = 3_000_000, 10, 48
P = torch.from_numpy(synth.blob_positions(N, 3).astype(np.float32)).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
A = torch.randn((N, 8 + cd), device="cuda", generator=g)
A[:, 7] = torch.sigmoid(A[:, 7])
G = torch.cat([P, A], dim=1).contiguous()
means, q, sc, op, col = G[:, :3].contiguous(), G[:, 3:7].contiguous(), G[:, 7:10].contiguous(), G[:, 10].contiguous(), G[:, 11:].contiguous()


def wall(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def two():
    PCvox, _, vidx, _, info = ops.voxelize_pc_batched(means, [0.0, 0.0, 0.0], 1.0, J, residuals=False, sorted_points=False)
    co = torch.cat([vidx, torch.tensor([N], dtype=torch.int64, device="cuda")]).int()
    return merge.merge_gaussian_clusters_with_indices(means, q, sc, op, col, info["sort_idx"].int(), co, True)


def one():
    return ops.voxelize_merge(G, [0.0, 0.0, 0.0], 1.0, J)


Gvox, info = one()
nv = info["Nvox"]
t2, t1 = wall(two), wall(one)
alg = 4.0 * N * (11 + cd) + 4.0 * nv * (11 + cd) + 24.0 * N * 4      # rows read once, merged rows written, 30-bit key sort
print(json.dumps({"gaussians": N, "voxels": nv, "columns": 11 + cd, "voxelize_then_merge_ms": round(t2, 4), "voxelize_merge_ms": round(t1, 4),
                  "alg_bytes": alg, "frac_of_peak": round(alg / (t1 * 1e-3) / 8e12, 4)}))
