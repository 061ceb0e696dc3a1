# raht_sort_keys against torch's stable sort: sizes around the tile edges, every digit-pass count, heavy duplicates
# (stability: equal keys keep their input order), then timings on the cfg3 key set. RAHT_SORT_ONESWEEP=0 selects the
# pass-by-pass form, RAHT_SORT_ROUNDS = 8 / 12 / 16 the one-sweep tile (2048 / 3072 / 4096 items).
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(7)
bad = 0
for n in (1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 12289, 100003, 1 << 20, 3000017):
    for nbits in (1, 3, 8, 9, 16, 17, 24, 30, 36, 42, 60, 63):
        if n > 200000 and nbits not in (3, 36, 60):
            continue
        hi = (1 << nbits)
        k = torch.randint(0, hi, (n,), device=dev, dtype=torch.int64, generator=g) if nbits < 63 else torch.randint(0, (1 << 62), (n,), device=dev, dtype=torch.int64, generator=g) * 2 + torch.randint(0, 2, (n,), device=dev, dtype=torch.int64, generator=g)
        if nbits >= 16 and n > 1000:          # force duplicates
            k[::3] = k[1::3][: k[::3].shape[0]] if k[1::3].shape[0] >= k[::3].shape[0] else k[::3]
        ks, idx = R.sort_keys(k, nbits=nbits)
        rs, ri = torch.sort(k, stable=True)
        ok = bool((ks == rs).all()) and bool((idx == ri).all())
        if not ok:
            bad += 1
            print("MISMATCH n=%d nbits=%d keys_equal=%s idx_equal=%s" % (n, nbits, bool((ks == rs).all()), bool((idx == ri).all())))
# the voxelizer on top of the same sort (its mean kernel runs behind the sort without a host round trip in between)
rng = np.random.default_rng(3)
for n, d, J in ((5000, 11, 6), (200000, 56, 8), (9000, 2, 5)):
    P = (rng.random((n, 3)) * 2.0).astype(np.float32)
    PC = torch.from_numpy(np.concatenate([P, rng.standard_normal((n, d)).astype(np.float32)], axis=1)).to(dev)
    PCvox, PCsorted, vidx, DeltaPC, info = R.voxelize_pc_batched(PC, None, None, J, device=dev)
    k = info["keys_sorted"]
    si = info["sort_idx"]
    ok = bool((k[1:] >= k[:-1]).all()) and bool(torch.equal(PCsorted, PC[si])) and int(vidx.shape[0]) == int((k[1:] != k[:-1]).sum()) + 1
    cnt = torch.diff(torch.cat([vidx, torch.tensor([n], device=dev)]))
    mean0 = torch.zeros(vidx.shape[0], device=dev, dtype=torch.float64).index_add_(0, torch.repeat_interleave(torch.arange(vidx.shape[0], device=dev), cnt), PCsorted[:, 3].double()) / cnt
    ok = ok and bool((PCvox[:, 3].double() - mean0).abs().max() < 1e-5)
    if not ok:
        bad += 1
        print("VOXELIZER MISMATCH n=%d d=%d J=%d" % (n, d, J))
print("correctness: %d mismatches" % bad)
# how often a one-sweep sort gave up and the call repeated the sort pass by pass (include/raht.h: raht_sort_fallbacks): 0 in
# every form but the ones that make a tile give up on purpose
from raht_3dgs_codec_amd import _lib as _l
nfb = int(_l.lib().raht_sort_fallbacks())
print("sort fallbacks: %d" % nfb)
if os.environ.get("RAHT_SORT_DEBUG_FAIL_TILE") is None:
    if nfb != 0:
        bad += 1
        print("UNEXPECTED SORT FALLBACKS: %d" % nfb)
elif os.environ.get("RAHT_SORT_ONESWEEP") != "0" and nfb == 0:
    bad += 1
    print("the forced give-up did not register as a fallback")
if "--time" in sys.argv:
    n, J, D, seed = synth.CONFIGS["cfg3"]
    keys = synth.sorted_unique_keys(n, J, seed)
    kd = torch.from_numpy(keys.view(np.int64)).cuda()
    ku = kd[torch.randperm(kd.shape[0], device="cuda", generator=g)].contiguous()
    for m in (250000, 1000000, 6000000, 50000000):
        kk = torch.randint(0, 1 << 36, (m,), device=dev, dtype=torch.int64, generator=g)
        for _ in range(5):
            R.sort_keys(kk, nbits=36)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20):
            R.sort_keys(kk, nbits=36)
        torch.cuda.synchronize()
        print("sort 36 bit, %d random keys: %.4f ms" % (m, (time.perf_counter() - t) / 20 * 1e3))
        del kk
    for nb in (36, 60):
        kk = ku if nb == 36 else ((ku << 24) | (ku & ((1 << 24) - 1)))
        for _ in range(10):
            R.sort_keys(kk, nbits=nb)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50):
            ko, idx = R.sort_keys(kk, nbits=nb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        assert bool((ko[1:] >= ko[:-1]).all())
        print("sort %d bit: %.4f ms" % (nb, dt * 1e3))
sys.exit(1 if bad else 0)
