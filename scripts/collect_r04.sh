#!/bin/bash
# Round-4 profile collection, ON THE GPU BOX:  bash scripts/collect_r04.sh   (outputs under gpurun_out/; summarise afterwards with
# python scripts/summarise_profiles.py r04)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/collect_profiles.sh r04 > $ROOT/gpurun_out/r04_collect.log 2>&1
echo "bench + traces + pmc done"; tail -2 $ROOT/gpurun_out/r04_collect.log
python $ROOT/scripts/dense_ref_timing.py 2>/dev/null | grep "^{" > $ROOT/gpurun_out/r04_dense_ref_timing.jsonl; cat $ROOT/gpurun_out/r04_dense_ref_timing.jsonl
python $ROOT/scripts/bench_modes.py 2>/dev/null | grep "^{" > $ROOT/gpurun_out/r04_bench_modes.jsonl; tail -3 $ROOT/gpurun_out/r04_bench_modes.jsonl
# the dense mode on the reference's loss under rocprofv3: full-resolution and quarter-resolution unknown (KITTI window, S = 2)
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/r04_dref $ROOT/gpurun_out/r04_dref_q
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_dref -- python $ROOT/scripts/dense_ref_profile.py 192 640 2 > $ROOT/gpurun_out/r04_dref.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_dref_q -- python $ROOT/scripts/dense_ref_profile.py 192 640 2 quarter > $ROOT/gpurun_out/r04_dref_q.log 2>&1
python $ROOT/scripts/seq_cache_timing.py 2>/dev/null | grep "^{" > $ROOT/gpurun_out/r04_seq_cache_timing.jsonl; cat $ROOT/gpurun_out/r04_seq_cache_timing.jsonl
echo "all done"
