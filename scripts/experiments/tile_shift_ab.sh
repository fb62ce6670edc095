#!/bin/bash
# A/B of the tile-constant colour shift in k_linearize (TC_TILE_SHIFT), ON THE GPU BOX: builds both variants, runs the parity tests of the pose
# modes on the shifted build, then alternates the variants under bench.py (in-kernel brackets):  bash scripts/experiments/tile_shift_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p tightly_coupled_sfm_amd/variants
for v in 0 1; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -DTC_TILE_SHIFT=$v tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o tightly_coupled_sfm_amd/variants/ts$v.so
done
cp tightly_coupled_sfm_amd/variants/ts1.so tightly_coupled_sfm_amd/libtcsfm_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_window_rule.py tests/test_gpu_truth.py tests/test_gpu_dense_reference.py -x -q 2>&1 | tail -4
bash scripts/experiments/ab_bench.sh ${1:-3} ts0.so ts1.so
