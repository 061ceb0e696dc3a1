# GPU timing of the float64 kernels (the reference's default precision) on cfg3
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raht_3dgs_codec_amd as R
from raht_3dgs_codec_amd import synth
n, J, D, seed = synth.CONFIGS["cfg3"]
V, keys, Ch = synth.scene(n, J, D, seed)
N = V.shape[0]
kd = torch.from_numpy(keys.view(np.int64)).cuda()
p = R.RahtPlan.from_keys(kd, 3 * J)
for dt in (torch.float32, torch.float64):
    C = torch.from_numpy(Ch).to(dt).cuda()
    for _ in range(3):                 # warm: the allocator's blocks for T / Rc exist before the timed loop
        T = p.forward(C, want_w=False); Rc = p.inverse(T)
    torch.cuda.synchronize()
    es = 8 if dt == torch.float64 else 4
    st = p.stage_stats(es, D)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        T = p.forward(C, want_w=False); Rc = p.inverse(T)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(dt, "fwd+inv %.3f ms  %.0f M-Gaussians/s  %.2f TB/s algorithmic" % (ms, N / ms / 1e3, 4 * N * D * es / ms / 1e9), st, "err", (Rc - C).abs().max().item())
