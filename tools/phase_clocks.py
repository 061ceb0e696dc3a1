"""Where a stage-0 tile's time goes: shader-clock stamps at the phase boundaries of tile_kernel, from a library
built with -DRAHT_PHASE_CLOCKS (tools/README.md has the build line; swap it in as libraht_hip.so for this run).
Prints, per kernel variant, the median cycles between consecutive stamps over the first 4096 tiles."""
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
if not hasattr(raw, "raht_debug_read_phase_clocks"):
    sys.exit("this libraht_hip.so was not built with -DRAHT_PHASE_CLOCKS")
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
dev = torch.device("cuda", 0)
N = V.shape[0]
Cd = torch.from_numpy(Ch).to(dev)
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
T = torch.empty_like(Cd); Q = torch.empty((N, D), dtype=torch.int32, device=dev); Crec = torch.empty_like(Cd)
vp = C.c_void_p
h = plan._h
NT, NS = 4096, 10
names = ["P0b loads issued+landed", "sync1", "P1 merge flags/hist", "P2 offsets/ranks", "P3 resolve", "P4 butterflies", "P5 write-back issued", "final sync"]


def run(inverse, fused):
    qp, qs = (vp(Q.data_ptr()), 0.01) if fused else (None, 0.0)
    src, dst = (T, Crec) if inverse else (Cd, T)
    for _ in range(3):
        _lib.check(L.raht_debug_run_stage(h, inverse, 0, vp(src.data_ptr()), D, D, vp(dst.data_ptr()), D, qp, D, qs, 0, None))
    buf = np.zeros((NT, NS), dtype=np.uint64)
    assert raw.raht_debug_read_phase_clocks(buf.ctypes.data_as(vp), NT) == NS
    d = np.diff(buf[:, :9].astype(np.int64), axis=1)
    tot = (buf[:, 8] - buf[:, 0]).astype(np.int64)
    print(f"{'inverse' if inverse else 'forward'} stage 0, {'fused' if fused else 'plain'}: median tile {np.median(tot):.0f} clocks "
          f"(p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
    for k, nm in enumerate(names):
        print(f"    {nm:28s} {np.median(d[:, k]):8.0f}  ({100 * np.median(d[:, k]) / np.median(tot):4.1f} %)")


plan.forward(Cd)            # fills T; warms the schedule
for inverse in (0, 1):
    for fused in (1, 0):
        run(inverse, fused)
