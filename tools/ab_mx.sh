#!/bin/bash
for d in 0 1 2 4 7; do echo "dbg=$d"; RAHT_MX_DBG=$d timeout -k 10 120 python tools/time_mixed.py --reps 100 2>/dev/null | tail -1 | python3 -c "import json,sys; o=json.loads(sys.stdin.read()); print({k:o[k] for k in o if k[-1] in '1'})"; done
