#!/bin/bash
# VGPRs / spills / occupancy of every tile and top kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
# Any spill is a regression: a kernel that touches scratch at all lost 30 %.
cd "$(dirname "$0")/../raht-3dgs-codec_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -fno-fast-math -ffp-contract=on ${EXTRA} \
  -Rpass-analysis=kernel-resource-usage -c ${1:-transform.hip} -o /tmp/reg_report.o 2>&1 | python3 -c "
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)', line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('spill',r'VGPRs Spill: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[k]=int(m.group(1))
import subprocess
for r in rows:
    if 'tile_kernel' in r['name'] or 'top_kernel' in r['name']:
        d=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
        d=re.sub(r'\(.*','',d).replace('void raht::','')
        print(f\"{d:48s} vgpr {r.get('vgpr')} spill {r.get('spill')} scratch {r.get('scratch')} occ {r.get('occ')}\")
"
