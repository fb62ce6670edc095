#!/bin/bash
# TCSFM_JOINT_SPLIT sweep (workgroups per target that share the joint solve's record sum; 0 = the library's own policy):
#   bash scripts/experiments/joint_split_sweep.sh [mode tag of scripts/dense_ref_timing.py, default ref_fi] [splits...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ref_fi}; shift
for s in ${@:-0 2 3 4}; do
  if [ $s = 0 ]; then unset TCSFM_JOINT_SPLIT; else export TCSFM_JOINT_SPLIT=$s; fi
  python scripts/dense_ref_timing.py $TAG 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('split $s %-8s S=%d %-30s %-10s us/window %7.1f' % (d['HxW'], d['S'], d['launch'][:30], d['tag'], d['us_per_window']))"
done
