#!/bin/bash
# A/B of library builds on ONE GPU box (run-to-run and box-to-box drift exceed the differences of interest):
#   bash scripts/experiments/ab_bench.sh <rounds> <variant.so> [<variant.so> ...]     (variants under tightly_coupled_sfm_amd/variants/, git-ignored)
# alternates the variants, prints k_linearize's in-kernel duration alone / with the chip full and the bench value.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    python $ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --sat-windows 32 --modes-budget 0 --shim-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r, s = d['roofline'], d['roofline_saturated']
print('$V', 'lin_us', r['avg_launch_us'], 'in_flight_us', r['in_flight']['avg_launch_us'], 'sat_us', s['avg_launch_us'], 'value', d['value'], 'single', d['single_stream']['value'], 'pack_us', r['other_kernels_avg_us_hip_events']['pack'])"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
