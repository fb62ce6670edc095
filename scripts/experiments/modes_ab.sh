#!/bin/bash
# A/B of library builds over every mode of scripts/bench_modes.py (regression check of a kernel change across the BASELINE configs) on ONE box:
#   bash scripts/experiments/modes_ab.sh <rounds> <a.so> <b.so> ...      (variants under tightly_coupled_sfm_amd/variants/, git-ignored)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    python $ROOT/scripts/bench_modes.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    us = d.get('us_per_call', d.get('us_per_window'))
    print('%-11s %-100s us %8.1f lin_us %s' % ('$V', d['config'][:100], us, d.get('linearize_us')))"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
