#!/bin/bash
# A/B of dense-mode library builds on ONE GPU box: variants under tightly_coupled_sfm_amd/variants/ (git-ignored), alternated; per variant the
# timings of scripts/dense_timing.py (BASELINE config 5: B=1 call and 32 windows per call, k_dense_linearize's in-kernel bracket).
#   bash scripts/experiments/dense_ab.sh <rounds> <a.so> <b.so> ...
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    python $ROOT/scripts/dense_timing.py basic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    if 'pairs' in d: print('%-10s %-8s pairs %3d us/call %8.1f k_dense_linearize_us %7.2f' % ('$V', d['HxW'], d['pairs'], d['us_per_call'], d['k_dense_linearize_us']))
    else: print('%-10s %-8s lanes %d %s us/window %7.1f' % ('$V', d['HxW'], d['calls_in_flight'], d['launches'], d['us_per_window']))"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
