"""One-off stress of the key sort and the voxelizer on the GPU box: 300 random (size, key width, ordering) sort cases against
torch.sort(stable=True), 60 random clouds (points, columns, depth, duplicates) against a torch restatement of keys / voxel starts /
means / residuals. SEED picks the sequence; the sort's knobs (RAHT_SORT_TICKET, RAHT_SORT_ONESWEEP, RAHT_SORT_ROUNDS) apply."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import raht_3dgs_codec_amd as R
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
g = torch.Generator(device="cuda"); g.manual_seed(5)
bad = 0
for it in range(300):
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 3000000)]))
    nbits = int(rng.integers(1, 64))
    mode = rng.integers(0, 4)
    hi = 1 << min(nbits, 62)
    k = torch.randint(0, hi, (n,), device="cuda", dtype=torch.int64, generator=g)
    if mode == 1: k = k >> int(rng.integers(0, max(1, nbits)))          # few distinct values
    if mode == 2: k, _ = torch.sort(k)                                    # already sorted
    if mode == 3: k = torch.flip(torch.sort(k)[0], [0])                   # reversed
    ks, idx = R.sort_keys(k, nbits=nbits)
    rs, ri = torch.sort(k, stable=True)
    if not (torch.equal(ks, rs) and torch.equal(idx, ri)):
        bad += 1; print("MISMATCH", it, n, nbits, mode)
print("fuzz sort: %d mismatches of 300" % bad)
# voxelizer fuzz vs a torch reference of keys/starts/means
bad2 = 0
for it in range(60):
    n = int(rng.integers(1, 200000)); d = int(rng.choice([0, 2, 5, 8, 11, 31, 56, 64])); J = int(rng.integers(1, 13))
    P = torch.rand((n, 3), device="cuda", generator=g) * float(rng.uniform(0.5, 9.0))
    if rng.random() < 0.3 and n > 3:
        m3 = P[1::3].shape[0]
        P[::3][:m3] = P[1::3]
    PC = torch.cat([P, torch.randn((n, d), device="cuda", generator=g)], dim=1).contiguous()
    try:
        PCvox, PCs, vidx, Dl, info = R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], 9.0, J, device="cuda")
    except Exception as e:
        print("ERR", n, d, J, e); bad2 += 1; continue
    k = info["keys_sorted"]; si = info["sort_idx"]
    ok = bool((k[1:] >= k[:-1]).all()) and torch.equal(PCs, PC[si])
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), (torch.nonzero(k[1:] != k[:-1]).flatten() + 1)])
    ok = ok and torch.equal(vidx, starts) and info["Nvox"] == starts.shape[0]
    if d > 0 and ok:
        cnt = torch.diff(torch.cat([starts, torch.tensor([n], device="cuda")]))
        seg = torch.repeat_interleave(torch.arange(starts.shape[0], device="cuda"), cnt)
        m = torch.zeros((starts.shape[0], d), dtype=torch.float64, device="cuda").index_add_(0, seg, PCs[:, 3:].double()) / cnt[:, None]
        ok = ok and bool(((PCvox[:, 3:].double() - m).abs() <= 1e-5 * (1 + m.abs())).all())
        ok = ok and bool(torch.equal(Dl[:, 3:], PCs[:, 3:] - PCvox[seg][:, 3:]))
    if not ok:
        bad2 += 1; print("VOX MISMATCH", it, n, d, J)
print("fuzz voxelizer: %d mismatches of 60" % bad2)
sys.exit(1 if (bad or bad2) else 0)
