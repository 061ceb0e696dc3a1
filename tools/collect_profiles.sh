#!/bin/bash
# copy what a round's profile runs (tools/profile_round.sh, tools/profile_extras.sh) left under gpurun_out/ into profiles/ (tracked)
tag=$1
cd "$(dirname "$0")/.." || exit 1
for f in bench.json bench_cfg2.json bench_cfg5.json bench_plain.json bench_under_rocprof.json kernel_stats.csv kernel_stats_by_grid.csv pmc_summary.csv \
         mixed_vs_f32.json two_streams.json dequant_inv_sqdiff.json fwd_quant_multi.json voxelize_merge.json e2e_frame_gpu_entropy.txt \
         mixed_kernel_durations.txt mixed_sq_counters.txt rlgr_sq_counters.txt rehearsal_4rank_cfg5_gloo_one_gpu.json \
         e2e_frame_gpu_entropy_batched.txt rlgr_batch.json rlgr_batch_sq_counters.txt prelude_wall.txt; do
  [ -s gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
[ -s gpurun_out/${tag}_traffic.json ] && cp gpurun_out/${tag}_traffic.json profiles/traffic.json
[ -d gpurun_out/${tag}_e2e ] && cp gpurun_out/${tag}_e2e/runtime_3dgs_gpu_entropy.csv profiles/${tag}_runtime_3dgs_gpu_entropy.csv && cp gpurun_out/${tag}_e2e/e2e_gpu_entropy.json profiles/${tag}_e2e_frame_gpu_entropy.json
[ -s gpurun_out/${tag}_e2e/runtime_3dgs_gpu_entropy_batched.csv ] && cp gpurun_out/${tag}_e2e/runtime_3dgs_gpu_entropy_batched.csv profiles/${tag}_runtime_3dgs_gpu_entropy_batched.csv && cp gpurun_out/${tag}_e2e/e2e_gpu_entropy_batched.json profiles/${tag}_e2e_frame_gpu_entropy_batched.json
ls profiles | grep "^${tag}_"
