#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path at the benchmark sizes.

    python tools/time_reference_cpu.py [--configs cfg2 cfg3] [--threads 8] [--repeats 5] [--out profiles/reference_cpu.json]

Runs ONLY in the build container: it imports the reference's operators unchanged from /root/reference/python
(RAHT_param_reorder_fast: RAHT_param.py:190-279, RAHT2_optimized: RAHT.py:252-336, inverse_RAHT_optimized:
iRAHT.py:40-114) and feeds them the SAME seeded scenes bench.py times on the MI355X (raht_3dgs_codec_amd.synth). float64
like the reference's drivers (encode_3dgs.py:82-83), torch CPU, torch.set_num_threads(--threads), one warm-up then the
median of --repeats runs per stage. Nothing of the reference travels to the GPU box: the numbers are written to
profiles/reference_cpu.json (and into BASELINE.md by hand), and bench.py quotes them as
cpu_baseline.reference_torch_cpu -- a constant with this provenance -- next to its live C-oracle timing.
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("RAHT_REFERENCE", "/root/reference/python")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["cfg2", "cfg3"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu.json"))
    a = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: this tool runs in the build container only")
    sys.path.insert(0, REF)
    import numpy as np
    import torch
    from RAHT import RAHT2_optimized
    from RAHT_param import RAHT_param_reorder_fast
    from iRAHT import inverse_RAHT_optimized
    from raht_3dgs_codec_amd import synth
    torch.set_num_threads(a.threads)
    out = {"what": "the reference's CPU path, imported unchanged, on bench.py's seeded scenes", "dtype": "float64", "torch": torch.__version__,
           "threads": a.threads, "cpu": cpu_model(), "host_cpus": os.cpu_count(), "repeats": a.repeats, "date": time.strftime("%Y-%m-%d"),
           "reference": {"RAHT_param": "python/RAHT_param.py:190-279", "RAHT": "python/RAHT.py:252-336", "iRAHT": "python/iRAHT.py:40-114"},
           "configs": {}}

    def med(fn):
        fn()                                                # warm-up
        ts = []
        for _ in range(a.repeats):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        return statistics.median(ts), r

    for name in a.configs:
        n, J, D, seed = synth.CONFIGS[name]
        V, keys, C = synth.scene(n, J, D, seed)
        N = int(V.shape[0])
        Vt = torch.from_numpy(V.astype(np.float64))
        Ct = torch.from_numpy(C.astype(np.float64))
        origin = torch.zeros(3, dtype=torch.float64)
        t_par, (ListC, FlagsC, weightsC, order) = med(lambda: RAHT_param_reorder_fast(Vt, origin, 2 ** J, J))
        t_fwd, (T, w) = med(lambda: RAHT2_optimized(Ct, ListC, FlagsC, weightsC))
        t_inv, Crec = med(lambda: inverse_RAHT_optimized(T, ListC, FlagsC, weightsC))
        err = float((Crec - Ct).abs().max())
        step = 0.01
        t_q, _ = med(lambda: torch.floor(T / step + 0.5).index_select(0, order))      # encode_3dgs.py:204,210
        row = {"rows": N, "channels": D, "depth_J": J, "levels": len(FlagsC), "RAHT_param_s": round(t_par, 4), "RAHT_s": round(t_fwd, 4), "iRAHT_s": round(t_inv, 4),
               "fwd_inv_s": round(t_fwd + t_inv, 4), "M_Gaussians_per_s": round(N / (t_fwd + t_inv) / 1e6, 4),
               "quant_reorder_s": round(t_q, 4), "roundtrip_abs_err": err}
        out["configs"][name] = row
        print(name, json.dumps(row), flush=True)
        del T, w, Crec, ListC, FlagsC, weightsC, Vt, Ct
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
