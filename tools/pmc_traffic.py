#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
bench.py into per-launch HBM traffic, with the gfx950 corrections of MI355X_MICROARCH.md (HBM):
WRITE_SIZE is in KiB and exact for wide stores; FETCH_SIZE (KiB) reports exactly 1/2 of the bytes
of a wide coalesced read -> doubled. Both facts are re-checked on torch's elementwise kernels of
the same run (known byte counts). Writes profiles/traffic.json and profiles/<tag>_pmc_summary.csv.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write r01 cfg3 2999072 59
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """the same hash bench.py computes: traffic.json is only quoted next to live timings of THIS build"""
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "raht-3dgs-codec_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "raht-3dgs-codec_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "raht.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    dfetch, dwrite, tag, workload, N, D = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    F, W = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    mat = N * D * 4
    # calibration on torch's abs kernel: reads one N x D float matrix, writes one
    cal = [k for k in F if "AbsFunctor" in k]
    cal_f = F[cal[0]][0] * 1024 / mat if cal else None
    cal_w = W[[k for k in W if "AbsFunctor" in k][0]][0] * 1024 / mat if cal else None
    rows = []
    for k in sorted(set(F) | set(W), key=lambda k: -(F.get(k, (0, 0))[0] + W.get(k, (0, 0))[0])):
        f, nf = F.get(k, (0.0, 0))
        w, nw = W.get(k, (0.0, 0))
        rows.append(dict(kernel=k, launches=max(nf, nw), FETCH_SIZE_KiB=round(f, 1), WRITE_SIZE_KiB=round(w, 1),
                         read_bytes=int(2 * f * 1024), write_bytes=int(w * 1024), hbm_bytes=int(2 * f * 1024 + w * 1024)))
    os.makedirs("profiles", exist_ok=True)
    with open(f"profiles/{tag}_pmc_summary.csv", "w", newline="") as fo:
        wr = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows[:24])

    def pick(sub):
        for r in rows:
            if sub in r["kernel"]:
                return r["hbm_bytes"]
        return None
    alg = 8.0 * N * D + 8.0 * N
    out = {}
    tp = "profiles/traffic.json"
    if os.path.exists(tp):
        out = json.load(open(tp))
    out[workload] = {
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                  "(gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads)",
        "calibration": {"torch_abs_kernel_read_over_known": None if cal_f is None else round(2 * cal_f, 4),
                        "torch_abs_kernel_write_over_known": None if cal_w is None else round(cal_w, 4)},
        "alg_bytes_per_launch": alg,
        "fused": {"fwd_stage0_bytes": pick("tile_kernel<float, false, true, true"),
                  "inv_stage0_bytes": pick("tile_kernel<float, true, true, true")},
        "plain": {"fwd_stage0_bytes": pick("tile_kernel<float, false, true, false"),
                  "inv_stage0_bytes": pick("tile_kernel<float, true, true, false")},
        "mixed": {"fwd_stage0_bytes": pick("tile_kernel_mx<false, true"),
                  "inv_stage0_bytes": pick("tile_kernel_mx<true, true")},
        "source": f"profiles/{tag}_pmc_summary.csv",
        "source_hash": kernel_source_hash(),
    }
    json.dump(out, open(tp, "w"), indent=1)
    print(json.dumps(out[workload], indent=1))


if __name__ == "__main__":
    main()
