P=raht-3dgs-codec_amd
cp $P/libraht_hip.so /tmp/lib_keep.so
for i in 1 2 3; do for v in base new; do cp $P/lib_variant_$v.bin $P/libraht_hip.so; timeout -k 10 200 python bench.py --no-quant --skip-oracle-gate --skip-legs --skip-prelude 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('plain $v', d['value'], d['ms_per_step'], 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done; done
cp /tmp/lib_keep.so $P/libraht_hip.so
