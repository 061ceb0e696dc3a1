# cfg2 (1 M x 14): tile geometry sweep of the fused step
for th in 512 256; do for tr in 0 448 384 320 256 192; do
RAHT_TILE_THREADS=$th timeout -k 10 200 python bench.py --workload cfg2 --steps 200 --warmup 50 --skip-legs --skip-prelude --skip-oracle-gate --tile-rows $tr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('threads=$th tile_rows=$tr', d['ms_per_step'], 'fwdq', b['fwd_quant_fused_ms'], 'invq', b['dequant_inv_fused_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['active_rows_per_stage'], d['config']['tile_rows'])"
done; done
