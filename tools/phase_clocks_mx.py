"""Where a stage-0 tile's time goes in the mixed-precision kernels next to the float32 ones: shader-clock stamps at the phase
boundaries (library built with -DRAHT_PHASE_CLOCKS: `make -C raht-3dgs-codec_amd/csrc EXTRA=-DRAHT_PHASE_CLOCKS OUT=../lib_variant_clk.bin
BUILD=build_clk`, swapped in as libraht_hip.so for the run). Medians over the first 4096 tiles of the cfg3 scene."""
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
if not hasattr(raw, "raht_debug_read_phase_clocks_mx"):
    sys.exit("this libraht_hip.so was not built with -DRAHT_PHASE_CLOCKS")
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
dev = torch.device("cuda", 0)
N = V.shape[0]
Cd = torch.from_numpy(Ch).to(dev)
kd = torch.from_numpy(keys.view(np.int64)).to(dev)
plan = R.RahtPlan.from_keys(kd, 3 * J)
vp = C.c_void_p
NT = 4096
names_f32 = ["P0b loads issued", "sync1", "P1 merge flags/hist", "P2 offsets/ranks (+inv: rows landed)", "P3 resolve (+fwd: rows landed)", "P4 butterflies", "P5 write-back issued", "final sync"]
names_mx = ["P0b loads issued", "sync1", "P1 merge flags/hist", "P2 offsets/ranks (+inv: landed, widen)", "P3 resolve (+fwd: landed, widen)", "P4 butterflies", "P5a survivors + f64 quantize + sync", "P5b row stores issued"]


def show(tag, buf, names, last):
    buf = buf[1024:]                        # (the later, smaller stages of a whole transform re-stamp the first tiles)
    d = np.diff(buf[:, :last + 1].astype(np.int64), axis=1)
    tot = (buf[:, last] - buf[:, 0]).astype(np.int64)
    print(f"{tag}: median tile {np.median(tot):.0f} clocks (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
    for k, nm in enumerate(names):
        print(f"    {nm:42s} {np.median(d[:, k]):8.0f}  ({100 * np.median(d[:, k]) / np.median(tot):4.1f} %)")


Q = plan.forward_quant(Cd, 0.01)
for inverse in (0, 1):
    for _ in range(3):
        if inverse:
            plan.dequant_inverse(Q, 0.01)
        else:
            plan.forward_quant(Cd, 0.01)
    buf = np.zeros((NT, 10), dtype=np.uint64)
    assert raw.raht_debug_read_phase_clocks(buf.ctypes.data_as(vp), NT) == 10
    show(f"float32 fused {'inverse' if inverse else 'forward'} stage 0", buf, names_f32, 8)
    for _ in range(3):
        if inverse:
            plan.dequant_inverse_mixed(Q, 0.01, 3)
        else:
            plan.forward_quant_mixed(Cd, 0.01, 3)
    buf = np.zeros((NT, 12), dtype=np.uint64)
    assert raw.raht_debug_read_phase_clocks_mx(buf.ctypes.data_as(vp), NT) == 12
    if inverse:
        buf[:, 7] = buf[:, 6]
    show(f"mixed fused {'inverse' if inverse else 'forward'} stage 0", buf, names_mx, 8)
