# later-stage tile geometry sweep on one box: tools/sweep_tail.sh [rows...]   (0 = automatic)
ROWS=("$@"); [ ${#ROWS[@]} -eq 0 ] && ROWS=(0 64 96 128 256 384 512)
for i in 1 2; do for tr in "${ROWS[@]}"; do
timeout -k 10 200 python bench.py --steps 100 --warmup 30 --skip-legs --skip-prelude --skip-oracle-gate --tail-rows $tr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('tail_rows=$tr', d['ms_per_step'], 'fwdq', b['fwd_quant_fused_ms'], 'invq', b['dequant_inv_fused_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['active_rows_per_stage'])"
done; done
