"""Gaussian merge kernel against the C restatement of cuda/merge_cluster.cu.
The restatement is PARITY UNPINNED (the reference extension is CUDA-only and cannot run in the
build container); HIP kernel and restatement use the same float32 operation order, so they must
agree bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(rng, N, K, cd, frac_empty=0.05):
    labels = rng.integers(0, K, size=N)
    means = rng.normal(size=(N, 3)).astype(np.float32)
    q = rng.normal(size=(N, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    scales = np.exp(rng.normal(-3, 1, size=(N, 3))).astype(np.float32)
    op = (1 / (1 + np.exp(-rng.normal(0, 2, size=N)))).astype(np.float32)
    colors = rng.normal(0, 0.5, size=(N, cd)).astype(np.float32)
    return labels, means, q, scales, op, colors


@pytest.mark.parametrize("N,K,cd,wbo", [(20000, 15000, 48, True), (5000, 300, 3, True), (3000, 2500, 60, False), (64, 1, 48, True)])
def test_merge_matches_restatement_bitwise(oracle, N, K, cd, wbo):
    from raht_3dgs_codec_amd import merge
    rng = np.random.default_rng(N + cd)
    labels, means, q, scales, op, colors = _scene(rng, N, K, cd)
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
    ci, co = merge.prepare_cluster_data(t(labels))
    got = merge.merge_gaussian_clusters_with_indices(t(means), t(q), t(scales), t(op), t(colors), ci, co, wbo)
    ref = oracle.merge_clusters(ci.cpu().numpy(), co.cpu().numpy(), means, q, scales, op, colors, wbo)
    for g, r, name in zip(got, ref, ("means", "quats", "scales", "opacities", "colors")):
        assert np.array_equal(g.cpu().numpy(), r), name
    # high-level entry, same thing from labels
    got2 = merge.merge_gaussian_clusters(t(means), t(q), t(scales), t(op), t(colors), t(labels), wbo)
    assert all(torch.equal(a, b) for a, b in zip(got, got2))
    # invariants of the reference kernel (merge_cluster.cu:76-96)
    qn = got[1].norm(dim=1)
    assert torch.allclose(qn, torch.ones_like(qn), atol=1e-5)
    assert float(got[3].max()) <= 1.0


def test_merge_edge_cases(oracle):
    """Empty clusters (zeros), zero total weight (means: divide by 1, colours: 0, quat: identity)."""
    from raht_3dgs_codec_amd import merge
    dev = "cuda"
    means = torch.tensor([[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]], device=dev)
    quats = torch.tensor([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 0., 0.]], device=dev)
    scales = torch.ones((3, 3), device=dev)
    op = torch.tensor([0.0, 0.0, 0.5], device=dev)
    colors = torch.arange(6, dtype=torch.float32, device=dev).reshape(3, 2)
    ci = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
    co = torch.tensor([0, 2, 2, 3], dtype=torch.int32, device=dev)          # cluster 1 is empty
    got = merge.merge_gaussian_clusters_with_indices(means, quats, scales, op, colors, ci, co, True)
    ref = oracle.merge_clusters(ci.cpu().numpy(), co.cpu().numpy(), means.cpu().numpy(), quats.cpu().numpy(),
                                scales.cpu().numpy(), op.cpu().numpy(), colors.cpu().numpy(), True)
    for g, r in zip(got, ref):
        assert np.array_equal(g.cpu().numpy(), r)
    assert got[1][0].tolist() == [0.0, 0.0, 0.0, 1.0]                       # zero weights -> identity quaternion
    assert got[0][1].tolist() == [0.0, 0.0, 0.0] and got[4][0].tolist() == [0.0, 0.0]
    with pytest.raises(RuntimeError):
        merge.merge_gaussian_clusters_with_indices(means.cpu(), quats, scales, op, colors, ci, co)


@pytest.mark.parametrize("N,J,cd,wbo", [(60000, 6, 48, True), (30000, 5, 3, True), (8000, 4, 60, False), (5000, 10, 48, True), (300, 2, 0, True)])
def test_voxelize_merge_equals_voxelizer_then_merge_kernel(oracle, N, J, cd, wbo):
    """raht_voxelize_merge (one call, one pass over the rows) against the reference's two steps -- voxelizer, then the merge kernel
    on the sort permutation / voxel starts (python/test_voxelize_3dgs.py:203-257) -- bit for bit, and against the C restatement of
    cuda/merge_cluster.cu (PARITY UNPINNED: the reference extension is CUDA-only; see the module docstring)."""
    from raht_3dgs_codec_amd import merge, ops
    rng = np.random.default_rng(N + cd + J)
    _, means, q, scales, op, colors = _scene(rng, N, 1, cd)
    if wbo:
        op[::97] = 0.0                                              # voxels whose whole weight may be zero
    means = (means * 0.7).astype(np.float32)
    G = np.concatenate([means, q, scales, op[:, None], colors], axis=1).astype(np.float32)
    Gd = torch.from_numpy(G).cuda()
    Gvox, info = ops.voxelize_merge(Gd, J=J, weight_by_opacity=wbo)
    # the two-step sequence
    PCvox, _, vidx, _, vinfo = ops.voxelize_pc_batched(Gd[:, :3].contiguous(), J=J, residuals=False, sorted_points=False)
    assert info["Nvox"] == vinfo["Nvox"] and torch.equal(info["sort_idx"], vinfo["sort_idx"]) and torch.equal(info["voxel_indices"], vidx)
    assert info["Nvox"] < N or J >= 10                              # (coarse grids: several Gaussians per voxel, something to merge)
    ci = vinfo["sort_idx"].int()
    co = torch.cat([vidx, torch.tensor([N], dtype=torch.int64, device="cuda")]).int()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    mm, mq, ms, mo, mc = merge.merge_gaussian_clusters_with_indices(t(means), t(q), t(scales), t(op), t(colors), ci, co, wbo)
    assert torch.equal(Gvox[:, :3], PCvox[:, :3])                   # integer voxel coordinates
    assert torch.equal(info["merged_means"], mm)
    assert torch.equal(Gvox[:, 3:7], mq) and torch.equal(Gvox[:, 7:10], ms) and torch.equal(Gvox[:, 10], mo)
    assert torch.equal(Gvox[:, 11:], mc)
    ref = oracle.merge_clusters(ci.cpu().numpy(), co.cpu().numpy(), means, q, scales, op, colors, wbo)
    assert np.array_equal(Gvox[:, 3:7].cpu().numpy(), ref[1]) and np.array_equal(Gvox[:, 11:].cpu().numpy(), ref[4])


def test_compress_to_nvox_fused_equals_two_steps():
    from raht_3dgs_codec_amd import pipeline
    rng = np.random.default_rng(5)
    N, cd = 40000, 48
    _, means, q, scales, op, colors = _scene(rng, N, 1, cd)
    t = lambda a: torch.from_numpy(a).cuda()   # noqa: E731
    a = pipeline.compress_to_nvox(t(means), t(q), t(scales), t(op), t(colors), J=6, fused=True)
    b = pipeline.compress_to_nvox(t(means), t(q), t(scales), t(op), t(colors), J=6, fused=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2]["merged_means"], b[2]["merged_means"]) and a[2]["Nvox"] == b[2]["Nvox"]
