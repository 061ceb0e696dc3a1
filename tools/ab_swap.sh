# A/B of two library builds on one box: swaps the in-tree .so between runs (variants: raht-3dgs-codec_amd/lib_variant_<name>.bin)
# tools/ab_swap.sh name1 name2 ... [-- extra bench args]
P=raht-3dgs-codec_amd
VARS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done; [ "$1" = "--" ] && shift
cp $P/libraht_hip.so /tmp/lib_keep.so
for i in 1 2 3; do for v in "${VARS[@]}"; do cp $P/lib_variant_$v.bin $P/libraht_hip.so; timeout -k 10 200 python bench.py --skip-oracle-gate --skip-legs --skip-prelude "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_ms']
print('$v', d['value'], d['ms_per_step'], 'fwd', b.get('fwd_quant_fused_ms'), 'inv', b.get('dequant_inv_fused_ms'), 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'], 'R', d['config'].get('tile_rows'), d['config']['roundtrip_rel_err'])"; done; done
cp /tmp/lib_keep.so $P/libraht_hip.so
