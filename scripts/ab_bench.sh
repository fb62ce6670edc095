#!/bin/bash
# A/B of library builds on ONE GPU box (run-to-run and box-to-box drift exceed the differences of interest):
#   bash scripts/ab_bench.sh <rounds> <variant.so> [<variant.so> ...]     (variants under tightly_coupled_sfm_amd/variants/, git-ignored)
# alternates the variants, prints k_linearize's in-kernel duration alone / with the chip full and the bench value.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    python $ROOT/bench.py --cpu-sample 0 --sat-windows 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$V', 'lin_us', d['roofline']['live']['avg_launch_us_in_kernel'], 'sat_us', d['roofline_saturated']['live']['avg_launch_us_in_kernel'], 'value', d['value'], 'single', d['single_stream']['value'])"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
