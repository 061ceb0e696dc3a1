# like ab_swap.sh, with 200 timed steps per run (step-time differences below 1 %)
P=raht-3dgs-codec_amd
cp $P/libraht_hip.so /tmp/lib_keep.so
for i in 1 2 3 4; do for v in "$@"; do cp $P/lib_variant_$v.bin $P/libraht_hip.so; timeout -k 10 200 python bench.py --steps 200 --warmup 20 --skip-oracle-gate --skip-legs --skip-prelude 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', d['value'], d['ms_per_step'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done; done
cp /tmp/lib_keep.so $P/libraht_hip.so
