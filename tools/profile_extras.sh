#!/bin/bash
# The round-4 additions to a round's profile set, on the GPU box: bash tools/profile_extras.sh <tag>   (after tools/profile_round.sh <tag>)
# Copy what should be judged from gpurun_out/ into profiles/ afterwards.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# ---- round 4: the new entry points, each with its own small tool (JSON lines under gpurun_out/<tag>_*.json) ----
timeout -k 10 200 python3 tools/time_mixed.py --reps 200 2>/dev/null | tail -1 > gpurun_out/${tag}_mixed_vs_f32.json
timeout -k 10 200 python3 tools/time_two_streams.py 2>/dev/null | tail -1 > gpurun_out/${tag}_two_streams.json
timeout -k 10 200 python3 tools/time_sqdiff.py 2>/dev/null | tail -1 > gpurun_out/${tag}_dequant_inv_sqdiff.json
timeout -k 10 200 python3 tools/time_multi.py 2>/dev/null | tail -1 > gpurun_out/${tag}_fwd_quant_multi.json
timeout -k 10 200 python3 tools/time_voxmerge.py 2>/dev/null | tail -1 > gpurun_out/${tag}_voxelize_merge.json
timeout -k 10 300 python3 tools/e2e_frame.py --entropy gpu --keep-rec 0 --out gpurun_out/${tag}_e2e > gpurun_out/${tag}_e2e_frame_gpu_entropy.txt 2>&1
timeout -k 10 300 python3 tools/e2e_frame.py --entropy gpu --keep-rec 0 --batch-steps 1 --out gpurun_out/${tag}_e2e > gpurun_out/${tag}_e2e_frame_gpu_entropy_batched.txt 2>&1
timeout -k 10 200 python3 tools/time_rlgr_batch.py 3 2048 2>/dev/null | tail -1 > gpurun_out/${tag}_rlgr_batch.json
timeout -k 10 200 python3 tools/prelude_loop.py > gpurun_out/${tag}_prelude_wall.txt 2>/dev/null
timeout -k 10 300 bash tools/trace_mx.sh ${tag} > gpurun_out/${tag}_mixed_kernel_durations.txt 2>&1
timeout -k 10 400 bash tools/pmc_rlgr.sh ${tag} > gpurun_out/${tag}_rlgr_sq_counters.txt 2>&1
timeout -k 10 400 bash tools/pmc_rlgr.sh ${tag}_batch tools/time_rlgr_batch.py 2 > gpurun_out/${tag}_rlgr_batch_sq_counters.txt 2>&1
timeout -k 10 300 bash tools/pmc_mx.sh ${tag}mx > gpurun_out/${tag}_mixed_sq_counters.txt 2>&1
# N > 1 rehearsal on the one GPU (4 ranks: the process guard allows 6 processes on the card, and the launcher and the parent count), so that the multi_gpu fields of such a line can be reviewed
timeout -k 10 500 python3 bench.py --gpus 4 --backend gloo --workload cfg5 --rows 6000000 --steps 20 --warmup 5 --settle-steps 0 --skip-legs --skip-prelude > gpurun_out/${tag}_rehearsal_4rank_cfg5_gloo_one_gpu.json 2>> gpurun_out/${tag}_bench.err
for f in mixed_vs_f32 two_streams dequant_inv_sqdiff fwd_quant_multi voxelize_merge; do echo "$f: $(cat gpurun_out/${tag}_$f.json | cut -c1-400)"; done
