# A/B by environment knob inside ONE box, interleaved: tools/ab_env.sh VAR val1 val2 ... [-- extra bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2 3; do for v in "${VALS[@]}"; do env $VAR=$v timeout -k 10 200 python bench.py --steps 100 --warmup 30 --skip-legs --skip-prelude --skip-oracle-gate "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('$VAR=$v', d['ms_per_step'], 'fwdq', b.get('fwd_quant_fused_ms'), 'invq', b.get('dequant_inv_fused_ms'), 'fwd', b['fwd_ms'], 'inv', b['inv_ms'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done; done
