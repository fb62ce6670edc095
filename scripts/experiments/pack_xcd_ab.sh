#!/bin/bash
# A/B of k_pack's XCD-aware tile order (-DTC_PACK_XCD=0 / 1), ON THE GPU BOX: bash scripts/experiments/pack_xcd_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p tightly_coupled_sfm_amd/variants
for v in 0 1; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=on -fPIC -shared -DTC_PACK_XCD=$v tightly_coupled_sfm_amd/csrc/tcsfm_api.hip -o tightly_coupled_sfm_amd/variants/pxcd$v.so
done
bash scripts/experiments/ab_bench.sh ${1:-3} pxcd0.so pxcd1.so
