# A/B on one box: usage bash tools/ab.sh ENVVAR [bench args]
v=$1; shift
for i in 1 2 3; do for x in 0 1; do env $v=$x timeout -k 10 200 python bench.py --skip-oracle-gate --skip-legs --skip-prelude "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('$v=$x', d['value'], d['ms_per_step'], 'fwd', b.get('fwd_quant_fused_ms', b['fwd_ms']), 'inv', b.get('dequant_inv_fused_ms', b['inv_ms']), 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done; done
