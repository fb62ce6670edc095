#!/bin/bash
# A/B of k_pack's pixel tile: 64x4 (production) against 32x8 and 16x16 (-DTC_PACK_TW=32 / 16), ON THE GPU BOX:  bash scripts/experiments/pack_tile_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p tightly_coupled_sfm_amd/variants
for v in 64 32 16; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -DTC_PACK_TW=$v tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o tightly_coupled_sfm_amd/variants/pk$v.so
done
cp tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep1.so
for r in $(seq 1 ${1:-3}); do
  for v in 64 32 16; do
    cp tightly_coupled_sfm_amd/variants/pk$v.so tightly_coupled_sfm_amd/libtcsfm_hip.so
    python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sat-windows 0 --modes-budget 0 --shim-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('pack tile $v x', 256//$v, 'pack_us', r['other_kernels_avg_us_hip_events']['pack'], 'value', d['value'], 'single', d['single_stream']['value'], 'merged', d['launch_mode']['merged']['value'])"
  done
done
cp /tmp/lib_keep1.so tightly_coupled_sfm_amd/libtcsfm_hip.so
