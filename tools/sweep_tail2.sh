#!/bin/bash
# later-stage tile geometry sweep on one box (round 3, height-keyed rounds): "rows:channels:top" triples, 0 = automatic
CFG=("$@"); [ ${#CFG[@]} -eq 0 ] && CFG=(0:0:0 0:0:8192 256:0:8192 368:32:4096 384:32:8192 512:32:8192 512:20:8192 768:16:8192 1024:12:8192 368:0:4096 0:0:0)
mkdir -p gpurun_out/sweep_tail2
for c in "${CFG[@]}"; do IFS=: read tr tc tp <<< "$c"
timeout -k 10 200 python bench.py --steps 100 --warmup 30 --skip-legs --skip-prelude --skip-oracle-gate --skip-cpu-baseline --tail-rows $tr --tail-ch $tc --top-rows $tp 2>gpurun_out/sweep_tail2/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('tail=$c', d['ms_per_step'], 'fwdq', b['fwd_quant_fused_ms'], 'invq', b['dequant_inv_fused_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['active_rows_per_stage'])" || tail -3 gpurun_out/sweep_tail2/err.txt
done
