"""Large inputs through the key sort and the voxelizer on the GPU box (many rounds of tiles, hundreds of tile groups): 50 M keys x 42
bits and 120 M keys x 60 bits (sorted, a permutation of the input, stable), 20 M points x 14 columns at J = 14 against a torch
restatement of keys / voxel starts / residuals."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import raht_3dgs_codec_amd as R
g = torch.Generator(device="cuda"); g.manual_seed(11)
for n, nb in ((50_000_000, 42), (120_000_000, 60)):
    k = torch.randint(0, 1 << nb, (n,), device="cuda", dtype=torch.int64, generator=g)
    k[::5] = k[1::5][: k[::5].shape[0]] if k[1::5].shape[0] >= k[::5].shape[0] else k[::5]
    torch.cuda.synchronize(); t = time.perf_counter()
    ks, idx = R.sort_keys(k, nbits=nb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = bool((ks[1:] >= ks[:-1]).all()) and bool(torch.equal(k[idx], ks))
    eq = ks[1:] == ks[:-1]
    ok = ok and bool((idx[1:][eq] > idx[:-1][eq]).all())          # stable
    print("sort n=%d bits=%d ok=%s %.2f ms" % (n, nb, ok, dt * 1e3), flush=True)
    del k, ks, idx, eq
n, d, J = 20_000_000, 11, 14
P = torch.rand((n, 3), device="cuda", generator=g) * 7.0
PC = torch.cat([P, torch.randn((n, d), device="cuda", generator=g)], dim=1).contiguous(); del P
torch.cuda.synchronize(); t = time.perf_counter()
PCvox, PCs, vidx, Dl, info = R.voxelize_pc_batched(PC, [0.0, 0.0, 0.0], 7.0, J, device="cuda")
torch.cuda.synchronize(); dt = time.perf_counter() - t
k = info["keys_sorted"]; si = info["sort_idx"]
ok = bool((k[1:] >= k[:-1]).all()) and bool(torch.equal(PCs, PC[si]))
starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), (torch.nonzero(k[1:] != k[:-1]).flatten() + 1)])
ok = ok and bool(torch.equal(vidx, starts))
cnt = torch.diff(torch.cat([starts, torch.tensor([n], device="cuda")]))
seg = torch.repeat_interleave(torch.arange(starts.shape[0], device="cuda"), cnt)
ok = ok and bool(torch.equal(Dl[:, 3:], PCs[:, 3:] - PCvox[seg][:, 3:]))
print("voxelize n=%d d=%d J=%d nvox=%d ok=%s %.2f ms" % (n, d, J, info["Nvox"], ok, dt * 1e3))
from raht_3dgs_codec_amd import _lib as _l  # noqa: E402
nfb = int(_l.lib().raht_sort_fallbacks())
print("sort fallbacks: %d" % nfb, flush=True)
assert nfb == 0, "a one-sweep sort gave up and fell back to the pass-by-pass form"
