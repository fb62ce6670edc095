#!/bin/bash
# Round-5 profile collection, ON THE GPU BOX:  bash scripts/collect_r05.sh   (outputs under gpurun_out/; summarise afterwards with
# python scripts/summarise_profiles.py r05)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/collect_profiles.sh r05 > $ROOT/gpurun_out/r05_collect.log 2>&1
echo "bench + traces + pmc done"; tail -2 $ROOT/gpurun_out/r05_collect.log
python $ROOT/scripts/dense_ref_timing.py 2>/dev/null | grep "^{" > $ROOT/gpurun_out/r05_dense_ref_timing.jsonl; wc -l $ROOT/gpurun_out/r05_dense_ref_timing.jsonl
python $ROOT/scripts/bench_modes.py 2>/dev/null | grep "^{" > $ROOT/gpurun_out/r05_bench_modes.jsonl; tail -2 $ROOT/gpurun_out/r05_bench_modes.jsonl
bash $ROOT/scripts/collect_r05_dref.sh
echo "all done"
