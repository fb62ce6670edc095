#!/bin/bash
# Round-3 profile collection, ON THE GPU BOX:  bash scripts/collect_r03.sh   (outputs under gpurun_out/; summarise afterwards)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/collect_profiles.sh r03 > $ROOT/gpurun_out/r03_collect.log 2>&1
echo "bench + traces + pmc done"; tail -2 $ROOT/gpurun_out/r03_collect.log
bash $ROOT/scripts/collect_mode_profiles.sh r03 dense320 dense448 dense_sat window_sel window_ref joint_kitti shard8 scale7 > $ROOT/gpurun_out/r03_modes.log 2>&1
tail -3 $ROOT/gpurun_out/r03_modes.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_posenet/trace -- python $ROOT/scripts/posenet_profile.py 2 100 > $ROOT/gpurun_out/r03_posenet.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_seqS1/trace -- python $ROOT/scripts/seq_profile.py 1 1 8 > $ROOT/gpurun_out/r03_seqS1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_seqS2/trace -- python $ROOT/scripts/seq_profile.py 2 1 8 > $ROOT/gpurun_out/r03_seqS2.log 2>&1
echo "all done"
