"""Seeded synthetic 3DGS scenes for tests and bench.py.

No datasets or checkpoints are reachable from the build or GPU boxes, so inputs are generated:
positions = mixture of Gaussian blobs clipped to [0,1)^3, voxelized to a 2^J grid, deduplicated and
Morton-sorted (what the reference's encode drivers expect, reference python/encode_3dgs.py:129-142);
attributes follow the 3DGS layout of reference python/data_util.py:336-368: quats(4), scales(3),
opacity(1), SH (3 DC + 45 rest), optionally preceded by xyz (the 59-column PCvox of
reference python/voxelize_pc.py:155).
"""
import numpy as np


def morton_keys(V, J):
    """digit_k = z_k + 2 y_k + 4 x_k at bits [3k, 3k+2] (reference python/voxelize_pc.py:50-57)."""
    V = V.astype(np.uint64)
    mc = np.zeros(V.shape[0], dtype=np.uint64)
    for i in range(J):
        s = np.uint64(i)
        mc |= (((V[:, 2] >> s) & np.uint64(1)) | (((V[:, 1] >> s) & np.uint64(1)) << np.uint64(1))
               | (((V[:, 0] >> s) & np.uint64(1)) << np.uint64(2))) << np.uint64(3 * i)
    return mc


def keys_to_coords(keys, J):
    keys = keys.astype(np.uint64)
    V = np.zeros((keys.shape[0], 3), dtype=np.int64)
    for i in range(J):
        dg = (keys >> np.uint64(3 * i)) & np.uint64(7)
        V[:, 2] |= ((dg & np.uint64(1)).astype(np.int64)) << i
        V[:, 1] |= (((dg >> np.uint64(1)) & np.uint64(1)).astype(np.int64)) << i
        V[:, 0] |= (((dg >> np.uint64(2)) & np.uint64(1)).astype(np.int64)) << i
    return V


def blob_positions(n, seed, nblobs=64, sigma=0.03, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(0.1, 0.9, size=(nblobs, 3))
    p = ctr[rng.integers(0, nblobs, size=n)] + rng.normal(0, sigma, size=(n, 3))
    p = np.clip(p, 0.0, 1.0 - 1e-9)
    return lo + p * (hi - lo)


def sorted_unique_keys(n_draws, J, seed, prefix_range=None):
    """Sorted, unique Morton keys of a blob scene. prefix_range=(a, b, nbits_prefix) restricts the
    scene to Morton prefixes [a, b) of the top nbits_prefix bits (a shard of a larger scene)."""
    P = blob_positions(n_draws, seed)
    V = np.floor(P * (1 << J)).astype(np.int64)
    k = morton_keys(V, J)
    if prefix_range is not None:
        a, b, pb = prefix_range
        span = np.uint64(b - a)
        sh = np.uint64(3 * J - pb)
        # fold every key into the shard's prefix range, keep the low bits
        low = k & ((np.uint64(1) << sh) - np.uint64(1))
        pre = (k >> sh) % span + np.uint64(a)
        k = (pre << sh) | low
    return np.unique(k)


def gaussian_attributes(n, D, seed, with_xyz_from=None):
    """float32 (n, D) attribute matrix. D in {11, 14, 56, 59, ...}; 14 / 59 prepend xyz."""
    rng = np.random.default_rng(seed + 7919)
    cols = []
    if with_xyz_from is not None:
        cols.append(with_xyz_from.astype(np.float32))
    q = rng.standard_normal((n, 4), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    cols.append(q)
    cols.append(np.exp(rng.standard_normal((n, 3), dtype=np.float32) - 4.0))
    cols.append(1.0 / (1.0 + np.exp(-2.0 * rng.standard_normal((n, 1), dtype=np.float32))))
    cols.append(0.5 * rng.standard_normal((n, 3), dtype=np.float32))
    have = sum(c.shape[1] for c in cols)
    if D > have:
        cols.append(0.1 * rng.standard_normal((n, D - have), dtype=np.float32))
    A = np.concatenate(cols, axis=1)[:, :D]
    return np.ascontiguousarray(A, dtype=np.float32)


def scene(n_draws, J, D, seed, prefix_range=None):
    """-> (V int64 (N,3) sorted unique, keys uint64 (N,), C float32 (N,D))."""
    keys = sorted_unique_keys(n_draws, J, seed, prefix_range)
    V = keys_to_coords(keys, J)
    xyz = V if D in (14, 59) else None
    C = gaussian_attributes(keys.shape[0], D, seed, with_xyz_from=xyz)
    return V, keys, C


CONFIGS = {
    # name: (n_draws, J, D, seed)         BASELINE.json configs / SURVEY 8d
    "cfg2": (1_000_000, 10, 14, 1),       # ~1M Gaussians, SH deg 0 (~14 attribute channels)
    "cfg3": (3_000_000, 12, 59, 2),       # ~3M Gaussians, SH deg 3 (59 channels) -- headline
    "cfg4": (3_000_000, 12, 59, 10),      # BASELINE configs[3]: one scene per GPU, sizes CFG4_DRAWS[rank], seed 10 + rank
    "cfg5": (50_000_000, 14, 59, 3),      # ~50M Gaussians (BASELINE configs[4] on ONE GPU; bench.py builds it on device)
}
# cfg4: 8 scenes of Mip-NeRF360-like sizes, one per GPU
CFG4_DRAWS = (1_000_000, 6_000_000, 2_000_000, 5_000_000, 3_000_000, 4_000_000, 1_500_000, 3_500_000)
