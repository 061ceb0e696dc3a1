for tr in 0 64 96 128; do for top in 0 8192; do
timeout -k 10 200 python bench.py --steps 100 --warmup 30 --skip-legs --skip-prelude --skip-oracle-gate --tail-rows $tr --top-rows $top 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('tail_rows=$tr top=$top', d['ms_per_step'], 'fwdq', b['fwd_quant_fused_ms'], 'invq', b['dequant_inv_fused_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['active_rows_per_stage'])"
done; done
