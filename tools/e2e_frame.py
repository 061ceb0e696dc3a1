#!/usr/bin/env python3
"""One whole frame through the codec on the MI355X at the headline size: the repo's counterpart of the reference's
python/encode_3dgs.py main loop (:126-411) on a synthetic voxelized 3DGS frame of ~3 M Gaussians x 56 attribute channels
(J = 12), the reference's nine quantization steps (:33, scaled by 1e-2 for unit-range attributes: SURVEY.md 8d). Writes the
reference's 20-column CSV (:70-76) and a JSON with what the CSV has no column for (device <-> host copies, transposes).

    python tools/e2e_frame.py [--rows 3000000] [--out gpurun_out/e2e] [--overlap 0|1] [--threads 0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raht_3dgs_codec_amd import pipeline, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=3_000_000)
ap.add_argument("--J", type=int, default=12)
ap.add_argument("--D", type=int, default=56)
ap.add_argument("--out", default="gpurun_out/e2e")
ap.add_argument("--overlap", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--entropy", default="host", choices=["host", "gpu"])
ap.add_argument("--seg-len", type=int, default=2048)
ap.add_argument("--keep-rec", type=int, default=1, help="0: the decode side never writes C_rec (raht_dequant_inv_sqdiff)")
ap.add_argument("--batch-steps", type=int, default=0, help="1 (with --entropy gpu): all steps through every stage at once")
ap.add_argument("--steps", default="0.01,0.04,0.08,0.12,0.16,0.20,0.24,0.32,0.64")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
V, keys, C = synth.scene(a.rows, a.J, a.D, seed=2)
steps = [float(x) for x in a.steps.split(",")]
Vt, Ct = torch.from_numpy(V), torch.from_numpy(C)
kw = dict(nthreads=a.threads, overlap=bool(a.overlap), entropy=a.entropy, seg_len=a.seg_len, keep_rec=bool(a.keep_rec), batch_steps=bool(a.batch_steps))
pipeline.encode_frame(Vt, Ct, a.J, steps if a.batch_steps else steps[:1], frame=0, **kw)         # warm-up (encode_3dgs.py:88-118)
t0 = time.time()
rows = pipeline.encode_frame(Vt, Ct, a.J, steps, frame=1, **kw)
wall = time.time() - t0
tag = ("gpu_entropy_batched" if a.batch_steps else "gpu_entropy") if a.entropy == "gpu" else ("overlap" if a.overlap else "sequential")
with open(os.path.join(a.out, f"runtime_3dgs_{tag}.csv"), "w") as f:
    f.write(pipeline.CSV_HEADER + "\n" + "\n".join(pipeline.format_row(r) for r in rows) + "\n")
extra = []
for r in rows:
    e = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items() if k not in ("C_rec",)}
    extra.append(e)
summary = {"rows": int(V.shape[0]), "channels": a.D, "J": a.J, "steps": steps, "mode": tag, "host_threads": a.threads or os.cpu_count(),
           "wall_s_all_steps": round(wall, 4), "wall_s_per_step": round(wall / len(steps), 4), "per_step": extra}
json.dump(summary, open(os.path.join(a.out, f"e2e_{tag}.json"), "w"), indent=1)
keys_ = ["RAHT_transform_time", "Transpose_time", "D2H_time", "Entropy_enc_time", "Entropy_dec_time", "Roundtrip_check_time", "H2D_time", "iRAHT_time", "PSNR_time", "Step_wall_time"]
print(tag, "rows", V.shape[0], "x", a.D, "wall per step %.4f s" % (wall / len(steps)))
for k in keys_:
    vals = [r.get(k, 0.0) for r in rows]
    print(f"  {k:22s} mean {np.mean(vals) * 1e3:9.3f} ms   (first step {vals[0] * 1e3:9.3f} ms)")
print("  symbols/s enc: %.1f M   dec: %.1f M" % (V.shape[0] * a.D / np.mean([r["Entropy_enc_time"] for r in rows]) / 1e6,
                                                 V.shape[0] * a.D / np.mean([r["Entropy_dec_time"] for r in rows]) / 1e6))
