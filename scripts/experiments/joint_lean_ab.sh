#!/bin/bash
# A/B of k_dense_joint builds on ONE GPU box: variants under tightly_coupled_sfm_amd/variants/ (git-ignored), alternated; per variant the
# reference-loss timings of scripts/dense_ref_timing.py (one call in flight / merged queued calls / the reference's minibatch of 6).
#   bash scripts/experiments/joint_lean_ab.sh <rounds> <filter> <a.so> <b.so> ...
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; F=$2; shift; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    python $ROOT/scripts/dense_ref_timing.py $F 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('%-9s %-8s S=%d %-44s %-10s us/window %7.1f joint_us %s' % ('$V', d['HxW'], d['S'], d['launch'][:44], d['tag'], d['us_per_window'], d.get('joint_kernel_us')))"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
