# GPU timing of the per-voxel Gaussian merge kernel (csrc/merge.hip) at codec sizes
import sys
import torch
sys.path.insert(0, '.')
from raht_3dgs_codec_amd import merge
for N, K, cd in ((3_000_000, 1_000_000, 48), (3_000_000, 2_700_000, 48), (1_000_000, 300_000, 3)):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    labels = torch.randint(0, K, (N,), device="cuda", generator=g)
    ci, co = merge.prepare_cluster_data(labels)
    means = torch.randn(N, 3, device="cuda"); quats = torch.randn(N, 4, device="cuda"); scales = torch.rand(N, 3, device="cuda")
    op = torch.rand(N, device="cuda"); colors = torch.randn(N, cd, device="cuda")
    f = lambda: merge.merge_gaussian_clusters_with_indices(means, quats, scales, op, colors, ci, co)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    Kr = co.shape[0] - 1
    byts = N * (11 + cd) * 4 + N * 4 + Kr * (11 + cd) * 4 + Kr * 4
    print("N=%d clusters=%d color_dim=%d : %.3f ms  %.0f M-Gaussians/s  %.2f TB/s (members read once + outputs)" % (N, Kr, cd, ms, N / ms / 1e3, byts / ms / 1e9))
