"""Where the plan build's kernels spend their time: shader-clock stamps at the phase boundaries of level_extent_kernel (per
1024-row block), tile_heights_kernel (per tile) and sched_tail_kernel (one workgroup). Library built with -DRAHT_PHASE_CLOCKS
(`make -C raht-3dgs-codec_amd/csrc EXTRA=-DRAHT_PHASE_CLOCKS OUT=../lib_variant_clk.bin BUILD=build_clk`, swapped in as
libraht_hip.so for the run). cfg3 keys; medians over the stamped workgroups."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402

L = _lib.lib()
raw = C.CDLL(_lib.SO_PATH)
if not hasattr(raw, "raht_debug_read_phase_clocks_plan"):
    sys.exit("this libraht_hip.so was not built with -DRAHT_PHASE_CLOCKS")
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
for _ in range(3):
    plan = R.RahtPlan.from_keys(kd, 3 * J)
torch.cuda.synchronize()
NB = 4096


def read(which):
    buf = np.zeros((NB, 12), dtype=np.uint64)
    assert raw.raht_debug_read_phase_clocks_plan(buf.ctypes.data_as(C.c_void_p), which, NB) == 12
    return buf


def show(tag, buf, names):
    last = len(names)
    d = np.diff(buf[:, :last + 1].astype(np.int64), axis=1)
    tot = (buf[:, last] - buf[:, 0]).astype(np.int64)
    print(f"{tag}: {buf.shape[0]} workgroups, median {np.median(tot):.0f} clocks (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}); "
          f"first start -> last end {int(buf[:, last].max() - buf[:, 0].min())} clocks")
    for k, nm in enumerate(names):
        print(f"    {nm:52s} {np.median(d[:, k]):8.0f}  ({100 * np.median(d[:, k]) / max(np.median(tot), 1):4.1f} %)")


ext = read(0)[: (kd.shape[0] + 1023) // 1024]
show("level_extent_kernel (1024 rows per block)", ext,
     ["keys loaded, levels, lvl / bucket stores issued", "barrier", "one word per (level present, wave)", "barrier", "neighbours from LDS, wl / wr stores issued",
      "queue + histograms", "barrier", "queue to global (+ overflow searches)"])
ht = read(1)
ht = ht[ht[:, 4] > ht[:, 0]]
print("tile_heights: levels walked per tile, median", np.median(ht[:, 11]), "max", ht[:, 11].max())
show("tile_heights_kernel (one wave per tile)", ht, ["metadata loaded", "partners resolved, level mask", "levels walked", "heights stored"])
tl = read(2)[:1]
print("sched_tail_kernel stamps (clocks from start):", [int(x - tl[0, 0]) for x in tl[0, 1:8] if x > 0], "top body:", [int(x - tl[0, 0]) for x in tl[0, 8:12]])
