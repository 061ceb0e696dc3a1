# bench.py --ablate 0/1/2/3 on fused and plain stage-0 kernels (1 = no butterflies, 2 = no merge resolution either)
for i in 1 2; do for a in 0 1 2 3; do timeout -k 10 200 python bench.py --skip-oracle-gate --skip-legs --skip-prelude --ablate $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate=$a', 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done; done
for a in 0 1 2 3; do timeout -k 10 200 python bench.py --no-quant --skip-oracle-gate --skip-legs --skip-prelude --ablate $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('plain ablate=$a', 'k_fwd', d['roofline']['avg_launch_ms'], 'k_inv', d['roofline_inv']['avg_launch_ms'])"; done
