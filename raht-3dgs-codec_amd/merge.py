"""Per-voxel Gaussian merge: drop-in for the reference's ``merge_cluster_cuda`` Python API
(reference cuda/merge_cluster_cuda/__init__.py:30-204), backed by the HIP kernel in csrc/merge.hip."""
import ctypes as C

import torch

from . import _lib
from ._lib import check


def prepare_cluster_data(cluster_labels):
    """cluster labels [N] -> (cluster_indices int32 [N] sorted by cluster, cluster_offsets int32 [K+1]).
    Mirrors reference __init__.py:30-75 (a stable argsort here, so members keep their input order)."""
    device = cluster_labels.device
    unique_clusters, inverse = torch.unique(cluster_labels, return_inverse=True)
    sorted_indices = torch.argsort(inverse, stable=True)
    ids = inverse[sorted_indices]
    boundaries = torch.cat([torch.tensor([0], device=device, dtype=torch.int64),
                            torch.where(ids[1:] != ids[:-1])[0] + 1,
                            torch.tensor([len(sorted_indices)], device=device, dtype=torch.int64)])
    return sorted_indices.to(torch.int32), boundaries.to(torch.int32)


def merge_gaussian_clusters_with_indices(means, quats, scales, opacities, colors, cluster_indices, cluster_offsets,
                                         weight_by_opacity=True):
    """Mirrors reference __init__.py:149-204. Returns (means, quats, scales, opacities, colors) per cluster."""
    for t in (means, quats, scales, opacities, colors, cluster_indices, cluster_offsets):
        if not t.is_cuda:
            raise RuntimeError("raht-3dgs-codec_amd: all inputs must be CUDA (HIP) tensors; no CPU path")
    means, quats, scales = means.contiguous().float(), quats.contiguous().float(), scales.contiguous().float()
    opacities, colors = opacities.contiguous().float(), colors.contiguous().float()
    ci, co = cluster_indices.contiguous().int(), cluster_offsets.contiguous().int()
    N = means.shape[0]
    if not (means.dim() == 2 and means.shape[1] == 3 and quats.shape == (N, 4) and scales.shape == (N, 3)
            and opacities.shape == (N,) and colors.dim() == 2 and colors.shape[0] == N):
        raise ValueError("expected means [N,3], quats [N,4], scales [N,3], opacities [N], colors [N,C]")
    K, cd, dev = co.shape[0] - 1, colors.shape[1], means.device
    out = [torch.zeros((K, 3), device=dev), torch.zeros((K, 4), device=dev), torch.zeros((K, 3), device=dev),
           torch.zeros((K,), device=dev), torch.zeros((K, cd), device=dev)]
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    with torch.cuda.device(dev):
        check(_lib.lib().raht_merge_clusters(p(ci), p(co), K, p(means), p(quats), p(scales), p(opacities), p(colors), cd,
                                             1 if weight_by_opacity else 0, p(out[0]), p(out[1]), p(out[2]), p(out[3]),
                                             p(out[4]), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return tuple(out)


def merge_gaussian_clusters(means, quats, scales, opacities, colors, cluster_labels, weight_by_opacity=True):
    """Mirrors reference __init__.py:78-147."""
    ci, co = prepare_cluster_data(cluster_labels.contiguous().long())
    return merge_gaussian_clusters_with_indices(means, quats, scales, opacities, colors, ci, co, weight_by_opacity)
