#!/bin/bash
# SQ counters of the segmented RLGR kernels (judge item: occupancy evidence): separate rocprofv3 --pmc passes over tools/rlgr_loop.py.
# usage (GPU box): bash tools/pmc_rlgr.sh <tag> [script and its arguments, default "tools/rlgr_loop.py 3"]  -> gpurun_out/<tag>_rlgr_pmc/, summary on stdout
#   bash tools/pmc_rlgr.sh r04b_batch tools/time_rlgr_batch.py 2     (the nine steps of a frame by one set of launches)
tag=$1
shift
cmd=${*:-tools/rlgr_loop.py 3}
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_rlgr_pmc/p$i --output-format csv -- python3 $cmd > gpurun_out/${tag}_rlgr_pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import collections, csv, glob
agg = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/${tag}_rlgr_pmc/p*/*/*counter_collection.csv")):
    tmp = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "seg_" in k:
            tmp[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in tmp.items():
        for c, x in v.items():
            agg[k][c] = sum(x) / len(x)
for f in sorted(glob.glob("gpurun_out/${tag}_rlgr_pmc/p1/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "seg_" in k:
            g = r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1))
            dur[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0)):
    d = sorted(dur.get(k, [0]))
    print(k[0], "grid", k[1], "median %.1f us" % d[len(d) // 2])
    for c, v in agg[k].items():
        print("    %-24s %16.0f" % (c, v) + ("   (KiB; FETCH_SIZE counts half of wide coalesced reads on gfx950)" if c in ("WRITE_SIZE", "FETCH_SIZE") else ""))
PY
