#!/bin/bash
# One round's profile set, on the GPU box: bash tools/profile_round.sh <tag>     (e.g. r02a)
#   gpurun_out/<tag>_bench.json                 python bench.py (the driver's command, default flags)
#   gpurun_out/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_bench_under_rocprof.json   its JSON line under the profiler
#   profiles/traffic.json, profiles/<tag>_pmc_summary.csv   HBM bytes per launch: separate --pmc FETCH_SIZE / WRITE_SIZE passes
# Copy what should be judged from gpurun_out/ into profiles/ afterwards (gpurun_out/ is scratch).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python3 bench.py --steps 200 --warmup 50 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; tail -3 gpurun_out/${tag}_bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o run -- python3 bench.py --steps 200 --warmup 50 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.log || { echo "rocprof run failed"; tail -3 gpurun_out/${tag}_prof.log; exit 1; }
cp gpurun_out/${tag}_prof/run_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
# the default command also times a cfg2 leg with the same kernel instantiations: per-(kernel, grid) averages keep cfg3 apart
python3 tools/stats_by_grid.py gpurun_out/${tag}_prof/run_kernel_trace.csv gpurun_out/${tag}_kernel_stats_by_grid.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$c -- python3 bench.py --steps 5 --warmup 2 --skip-legs --skip-prelude --skip-oracle-gate > gpurun_out/${tag}_pmc_$c.json 2> gpurun_out/${tag}_pmc_$c.log || { echo "pmc pass $c failed"; tail -3 gpurun_out/${tag}_pmc_$c.log; exit 1; }
done
read N D < <(python3 -c "import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['config']['rows_per_gpu'], d['config']['channels'])")
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE $tag cfg3 $N $D > gpurun_out/${tag}_traffic.log 2>&1 || { echo "pmc_traffic failed"; tail -5 gpurun_out/${tag}_traffic.log; }
cp profiles/traffic.json gpurun_out/${tag}_traffic.json; cp profiles/${tag}_pmc_summary.csv gpurun_out/ 2>/dev/null
# once more, now that profiles/traffic.json belongs to this build: the line that quotes the measured HBM bytes
timeout -k 10 500 python3 bench.py --steps 200 --warmup 50 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for w in cfg2 cfg5; do timeout -k 10 500 python3 bench.py --workload $w --steps 100 --warmup 20 --skip-legs --skip-prelude > gpurun_out/${tag}_bench_$w.json 2>> gpurun_out/${tag}_bench.err; done
timeout -k 10 300 python3 bench.py --no-quant --steps 200 --warmup 50 --skip-legs --skip-prelude > gpurun_out/${tag}_bench_plain.json 2>> gpurun_out/${tag}_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline_inv']['frac'], 'whole-step frac', d['path_hbm']['whole_step_frac_of_peak'])
print('prelude', {k:(v['ms'] if isinstance(v,dict) else v) for k,v in d['prelude'].items()})
print('f64', d.get('f64',{}).get('fwd_inv_ms'), 'cfg2', d.get('cfg2',{}).get('ms_per_step'), 'cpu', d.get('cpu_baseline',{}).get('value'))
"
head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
