#!/bin/bash
# Round-5 profiles of the dense mode on the reference's loss, ON THE GPU BOX: bash scripts/collect_r05_dref.sh  (outputs under gpurun_out/r05_dref*)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for cfg in "192 640 2 full 1" "240 320 1 full 1" "192 640 2 quarter 1" "192 640 2 full 6" "240 320 1 full 6" "192 640 2 quarter_free 1" "192 640 2 free 1"; do
  set -- $cfg
  tag=r05_dref_$1x$2_S$3_$4_B$5
  rm -rf $ROOT/gpurun_out/$tag
  TCSFM_PROFILE_B=$5 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$tag -- python3 $ROOT/scripts/dense_ref_profile.py $1 $2 $3 $4 > $ROOT/gpurun_out/$tag.log 2>&1
  cp $(find $ROOT/gpurun_out/$tag -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/${tag}_kernel_stats.csv
  rm -rf $ROOT/gpurun_out/$tag
done
echo done
