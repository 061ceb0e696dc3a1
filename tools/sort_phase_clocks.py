"""Where a one-sweep sort tile's time goes (scan_sort.hip, os_pass_kernel): 100 MHz wall-clock stamps at the phase
boundaries of every tile of one digit pass, from a library built with -DRAHT_OS_CLOCKS:
  make -C raht-3dgs-codec_amd/csrc EXTRA=-DRAHT_OS_CLOCKS OUT=../lib_variant_osclk.bin BUILD=/tmp/b_osclk
and swapped in for this run (RAHT_LIB_SWAP=raht-3dgs-codec_amd/lib_variant_osclk.bin copies it over libraht_hip.so in the
box's scratch copy of the repo)."""
import ctypes as C
import os
import shutil
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sw = os.environ.get("RAHT_LIB_SWAP")
if sw:
    shutil.copyfile(os.path.join(root, sw), os.path.join(root, "raht-3dgs-codec_amd", "libraht_hip.so"))
sys.path.insert(0, root)
import raht_3dgs_codec_amd as R  # noqa: E402
from raht_3dgs_codec_amd import _lib, synth  # noqa: E402

_lib.lib()
raw = C.CDLL(_lib.SO_PATH)
if not hasattr(raw, "raht_debug_sort_clocks"):
    sys.exit("this libraht_hip.so was not built with -DRAHT_OS_CLOCKS")
n, J, D, seed = synth.CONFIGS["cfg3"]
keys = synth.sorted_unique_keys(n, J, seed)
kd = torch.from_numpy(keys.view(np.int64)).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
ku = kd[torch.randperm(kd.shape[0], device="cuda", generator=g)].contiguous()
rounds = int(os.environ.get("RAHT_SORT_ROUNDS", "16"))
nt = -(-ku.shape[0] // (256 * rounds))
names = ["ticket", "loads + wave histograms", "publish + slot scan", "ranks -> LDS", "tiles before, in group", "groups before", "write-out"]
for shift in (8, 29):
    raw.raht_debug_sort_clocks(None, 0, shift)
    for _ in range(3):
        R.sort_keys(ku, nbits=36)
    torch.cuda.synchronize()
    buf = np.zeros((nt, 8), dtype=np.uint64)
    assert raw.raht_debug_sort_clocks(buf.ctypes.data_as(C.c_void_p), nt, -1) == 0
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1) * 10.0 / 1e3           # us
    print(f"pass at shift {shift}: {nt} tiles; kernel span {(t[:, 7].max() - t0) * 10 / 1e3:.1f} us; first stamp spread {(t[:, 0].max() - t0) * 10 / 1e3:.1f} us")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} median {np.median(d[:, k]):6.2f} us  p90 {np.percentile(d[:, k], 90):6.2f}  max {d[:, k].max():6.2f}")
    print(f"  tile lifetime               median {np.median((t[:, 7] - t[:, 0]) * 10 / 1e3):6.2f} us  max {((t[:, 7] - t[:, 0]) * 10 / 1e3).max():6.2f}")
    for k in range(8):
        print(f"  stamp {k}: relative to kernel start, median {np.median((t[:, k] - t0) * 10 / 1e3):6.2f} us  max {((t[:, k] - t0) * 10 / 1e3).max():6.2f}")
