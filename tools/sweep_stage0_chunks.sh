run() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 30 --skip-legs --skip-prelude --skip-oracle-gate $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('$1 $2', d['ms_per_step'], 'fwdq', b.get('fwd_quant_fused_ms'), 'invq', b.get('dequant_inv_fused_ms'), 'fwd', b['fwd_ms'], 'inv', b['inv_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['active_rows_per_stage'], d['config']['roundtrip_rel_err'])"; }
for i in 1 2; do
run "X=0" ""
run "RAHT_STAGE0_CH=32" "--tile-rows 336"
run "RAHT_STAGE0_CH=32" "--tile-rows 336 --tail-rows 184"
run "RAHT_STAGE0_CH=32" "--tile-rows 256"
run "RAHT_STAGE0_CH=20" "--tile-rows 480"
done
