for i in 1 2 3 4; do for pb in 0 1; do
timeout -k 10 200 python bench.py --steps 200 --warmup 50 --skip-legs --skip-prelude --skip-oracle-gate --pooled-buffers $pb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('pooled=$pb', d['ms_per_step'], 'fwdq', b['fwd_quant_fused_ms'], 'invq', b['dequant_inv_fused_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"
done; done
