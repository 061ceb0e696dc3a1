#!/bin/bash
# per-kernel times of the prelude (plan build / sort / voxelizer) on the GPU box: bash tools/profile_prelude.sh <tag> [plan|sort|vox ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/${tag}_prelude
python3 tools/prelude_loop.py "$@" > gpurun_out/${tag}_prelude/wall.txt 2>&1; cat gpurun_out/${tag}_prelude/wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prelude/prof -o run -- python3 tools/prelude_loop.py "$@" > gpurun_out/${tag}_prelude/under_rocprof.txt 2> gpurun_out/${tag}_prelude/prof.log || { tail -3 gpurun_out/${tag}_prelude/prof.log; exit 1; }
cp gpurun_out/${tag}_prelude/prof/run_kernel_stats.csv gpurun_out/${tag}_prelude_kernel_stats.csv
python3 - gpurun_out/${tag}_prelude_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):5d}  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
