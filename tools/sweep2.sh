#!/bin/bash
# fused step: tile geometry sweep (threads, stage-0 rows)
for cfg in "512 184" "512 160" "512 128" "256 96" "256 88" "256 128" "512 256" "256 184"; do set -- $cfg; thr=$1; R=$2
  echo -n "threads=$thr R=$R : "
  RAHT_TILE_THREADS=$thr timeout -k 10 200 python bench.py --steps 20 --warmup 5 --skip-oracle-gate --skip-legs --skip-prelude --tile-rows $R 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('MG/s',d['value'],'ms',d['ms_per_step'],'k_fwd',d['roofline']['avg_launch_ms'],'k_inv',d['roofline_inv']['avg_launch_ms'], d['breakdown_ms']['fwd_quant_fused_ms'], d['breakdown_ms']['dequant_inv_fused_ms'], d['config'].get('active_rows_per_stage'))"
done
