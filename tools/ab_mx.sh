#!/bin/bash
# mixed-precision fused step on cfg3 next to the float32 one, default geometry and a few tile heights
for r in 0 160 176; do echo "rows=$r"; timeout -k 10 120 python tools/time_mixed.py --reps 100 --tile-rows $r 2>/dev/null | tail -1 | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print({k:o[k] for k in o if k[-1] in '01'})"; done
