#!/usr/bin/env python3
"""Per (kernel, grid size) launch statistics from a rocprofv3 kernel trace CSV. bench.py's default run also times
a cfg2 leg with the SAME kernel instantiations on a smaller scene, so rocprofv3's own per-name --stats average mixes
two workloads; split by grid size the cfg3 launches (stage 0: ceil(N / 184) workgroups of 512 threads) stand alone.

    python tools/stats_by_grid.py gpurun_out/<tag>_prof/run_kernel_trace.csv profiles/<tag>_kernel_stats_by_grid.csv
"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[(r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for (name, grid, wg), d in rows.items():
    out.append(dict(Name=name, GridThreads=grid, WorkgroupSize=wg, Calls=len(d), TotalDurationNs=sum(d), AverageNs=round(sum(d) / len(d), 1), MinNs=min(d), MaxNs=max(d)))
out.sort(key=lambda r: -r["TotalDurationNs"])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(out[0].keys()))
    w.writeheader()
    w.writerows(out[:60])
for r in out[:8]:
    print(r["AverageNs"], r["Calls"], r["GridThreads"], r["Name"][:90])
