libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2 3; do for l in "${libs[@]}"; do RAHT_HIP_LIB_EXPERIMENT=$PWD/raht-3dgs-codec_amd/$l timeout -k 10 200 python bench.py --skip-cpu-baseline --skip-prelude "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('$l', d['value'], d['ms_per_step'], 'fwd', b.get('fwd_quant_fused_ms', b['fwd_ms']), 'inv', b.get('dequant_inv_fused_ms', b['inv_ms']), 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], d['config']['roundtrip_rel_err'])"; done; done
