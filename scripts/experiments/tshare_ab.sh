#!/bin/bash
# A/B of the shared-pack form of the pose window calls (TCSFM_TSHARE=0 / 1, same library), ON THE GPU BOX: bash scripts/experiments/tshare_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for r in $(seq 1 ${1:-3}); do
  for v in 0 1; do
    TCSFM_TSHARE=$v python bench.py --steps 20 --warmup 5 --cpu-sample 0 --sat-windows 32 --modes-budget 0 --shim-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r, s = d['roofline'], d['roofline_saturated']
print('tshare $v', 'lin_us', r['avg_launch_us'], 'sat_us', s['avg_launch_us'], 'value', d['value'], 'single', d['single_stream']['value'], 'pack_us', r['other_kernels_avg_us_hip_events']['pack'], 'chip', r['timed_mode']['chip_level']['frac'], 'traffic', r.get('traffic'))"
  done
done
