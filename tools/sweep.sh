#!/bin/bash
for cfg in "256 128" "256 192" "512 192" "512 256" "256 256"; do set -- $cfg; thr=$1; R=$2
  echo -n "threads=$thr R=$R : "
  RAHT_TILE_THREADS=$thr timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-quant --skip-cpu-baseline --tile-rows $R 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('MG/s',d['value'],d['breakdown_ms'],'k_fwd',d['roofline']['avg_launch_ms'],'k_inv',d['roofline_inv']['avg_launch_ms'], 'GB/s fwd', d['roofline']['achieved'])"
done
