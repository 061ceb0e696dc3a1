#!/bin/bash
# A/B of one environment knob inside ONE gpurun call (same box): bench.py alternately without / with `$1` (NAME=VALUE),
# $2 rounds (default 3), extra bench arguments after that. Prints ms_per_step, stage-0 launch times and the cfg2 leg.
knob="$1"; rounds="${2:-3}"; shift 2
out=gpurun_out/ab_env2; mkdir -p $out
for r in $(seq 1 $rounds); do
  python bench.py --skip-oracle-gate --skip-cpu-baseline --skip-prelude "$@" > $out/a_$r.json 2> $out/a_$r.err || { tail -5 $out/a_$r.err; exit 1; }
  env "$knob" python bench.py --skip-oracle-gate --skip-cpu-baseline --skip-prelude "$@" > $out/b_$r.json 2> $out/b_$r.err || { tail -5 $out/b_$r.err; exit 1; }
done
python - "$knob" $rounds <<'PY'
import json,sys
knob,rounds=sys.argv[1],int(sys.argv[2])
for tag,name in (("a","default"),("b",knob)):
    for r in range(1,rounds+1):
        d=json.load(open(f"gpurun_out/ab_env2/{tag}_{r}.json"))
        br=d.get("breakdown_ms",{})
        print(f"{name:28s} step {d['ms_per_step']:.4f} ms  stage0 fwd {d['roofline']['avg_launch_ms']:.4f} inv {d['roofline_inv']['avg_launch_ms']:.4f}  fused fwd {br.get('fwd_quant_fused_ms')} inv {br.get('dequant_inv_fused_ms')}  plain fwd {br.get('fwd_ms')} inv {br.get('inv_ms')}  cfg2 {d.get('cfg2',{}).get('ms_per_step')}")
PY
