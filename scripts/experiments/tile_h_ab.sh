#!/bin/bash
# A/B of k_linearize's tile: 32x16 / 512 threads (production) against 32x8 / 256 threads (-DTC_TILE_H=8), ON THE GPU BOX: builds both variants,
# runs the pose-mode parity tests on the 32x8 build, then alternates the variants under bench.py:  bash scripts/experiments/tile_h_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p tightly_coupled_sfm_amd/variants
ALT=${TH_ALT:-8}      # TH_ALT=32: 32x32 tiles of 1024 threads (one workgroup per CU, 133 KB of LDS)
for v in 16 $ALT; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -DTC_TILE_H=$v tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o tightly_coupled_sfm_amd/variants/th$v.so
done
cp tightly_coupled_sfm_amd/libtcsfm_hip.so /tmp/lib_keep0.so
cp tightly_coupled_sfm_amd/variants/th$ALT.so tightly_coupled_sfm_amd/libtcsfm_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_window_rule.py tests/test_gpu_coalesce.py -x -q 2>&1 | tail -3
cp /tmp/lib_keep0.so tightly_coupled_sfm_amd/libtcsfm_hip.so
bash scripts/experiments/ab_bench.sh ${1:-3} th16.so th$ALT.so
