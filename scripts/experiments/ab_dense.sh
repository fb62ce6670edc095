#!/bin/bash
# bash scripts/experiments/ab_dense.sh <rounds> <variant.so> [...]   (see ab_modes.sh)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cp $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for V in "$@"; do
    cp $ROOT/tightly_coupled_sfm_amd/variants/$V $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
    echo "$V $(python $ROOT/scripts/experiments/ab_dense.py 2>/dev/null | tail -1)"
  done
done
cp /tmp/lib_keep.so $ROOT/tightly_coupled_sfm_amd/libtcsfm_hip.so
