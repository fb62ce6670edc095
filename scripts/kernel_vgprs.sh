#!/bin/bash
# VGPRs / scratch / occupancy of every k_linearize instantiation (hipcc remarks):  bash scripts/kernel_vgprs.sh [extra -D flags]
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -Rpass-analysis=kernel-resource-usage "$@" \
  tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o /tmp/_vg.so 2>&1 | python3 -c "
import re,sys
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}
    for k in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',l)
        if m: d[k[:5]]=m.group(1)
    if 'LDS Size' in l and cur and 'k_linearize' in cur: print(cur[13:70], d)
"
