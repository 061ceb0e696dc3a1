#!/bin/bash
# per-kernel durations of the float32 and the mixed fused step on cfg3 (rocprofv3 --kernel-trace)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$1_trace -- python tools/time_mixed.py --reps 50 > gpurun_out/$1_trace.log 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob('gpurun_out/$1_trace/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void raht::','')
    if 'tile_kernel' in k or 'top_kernel' in k:
        d[(k, r.get('Grid_Size_X') or r.get('Grid_Size'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v); print(f"{k[0]:50s} grid {k[1]:>9s} n={len(v):4d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
