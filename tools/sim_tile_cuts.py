# CPU simulation (numpy): survivors of stage 0 on the cfg3 key set for tile boundaries placed at high-level node starts
# instead of every R rows -- greedy (next cut = row of the highest level in the last `win` rows a tile may still hold) and
# non-greedy (cut k inside a fixed window). DESIGN.md 10.1. Runs in ~1 min, no GPU.
import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
from raht_3dgs_codec_amd import synth
n, J, D, seed = synth.CONFIGS["cfg3"]
keys = synth.sorted_unique_keys(n, J, seed).astype(np.uint64)
N = keys.shape[0]
def msb(x):
    m = np.zeros(x.shape, dtype=np.int64); t = x.copy()
    for s in (32,16,8,4,2,1):
        big = t >= (np.uint64(1) << np.uint64(s)); m[big] += s; t[big] >>= np.uint64(s)
    return m
lvl = np.full(N, 255, dtype=np.int64); lvl[1:] = msb(keys[1:] ^ keys[:-1])
l = lvl[1:].astype(np.uint64); i = np.arange(1, N)
wl = np.zeros(N, dtype=np.int64); wr = np.zeros(N, dtype=np.int64)
wl[1:] = i - np.searchsorted(keys, (keys[:-1] >> l) << l, side="left")
wr[1:] = np.searchsorted(keys, ((keys[1:] >> l) + np.uint64(1)) << l, side="left") - i
rows = np.arange(N)
def survivors(cuts):
    # cuts: sorted tile start rows (first = 0); returns number of survivors
    cuts = np.asarray(cuts); ends = np.concatenate([cuts[1:], [N]])
    t = np.searchsorted(cuts, rows, side="right") - 1
    start = cuts[t]; end = ends[t]
    merged = (rows > 0) & (rows - wl >= start) & (rows + wr <= end)
    return int((~merged).sum()), len(cuts)
R = 184
fixed = np.arange(0, N, R)
print("fixed R=184:", survivors(fixed))
for win in (8, 16, 32, 48):
    # greedy: next cut = row with max lvl in (pos + R - win, pos + R]
    cuts = [0]; pos = 0
    lv = lvl.copy(); lv[0] = 0
    while pos + R < N:
        lo, hi = pos + R - win + 1, min(pos + R, N - 1)
        j = lo + int(np.argmax(lv[lo:hi + 1]))
        cuts.append(j); pos = j
    s, nt = survivors(cuts)
    print(f"adaptive win={win}: survivors {s} tiles {nt} avg rows {N/nt:.1f}")
print("non-greedy: cut k = row of max lvl in (k*S, k*S + win], S = R - win")
lv = lvl.copy(); lv[0] = 0
for win in (4, 8, 16, 32):
    S = R - win
    ks = np.arange(1, (N - 1) // S + 1)
    cuts = [0]
    for k in ks:
        lo, hi = k * S + 1, min(k * S + win, N - 1)
        if lo > hi: break
        cuts.append(lo + int(np.argmax(lv[lo:hi + 1])))
    cuts = np.unique(np.array(cuts))
    s, nt = survivors(cuts)
    print(f"win={win}: survivors {s} tiles {nt} avg rows {N/nt:.1f} max tile {int(np.diff(np.concatenate([cuts,[N]])).max())}")
# second stage of the greedy win=8 schedule
