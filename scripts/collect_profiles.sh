#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash scripts/collect_profiles.sh r01_x'): bench line, rocprofv3 kernel stats and the
# PMC passes (separate runs, --pmc never combined with tracing) for the default bench.py workload.
# Outputs land in gpurun_out/<tag>/ ; summarise with scripts/summarise_profiles.py and copy the result into profiles/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT          # (a re-collection must not leave an earlier build's files beside the new ones: rocprofv3 names them by process id)
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python -c "import sys, json; sys.path.insert(0, '$ROOT'); import bench; json.dump({'source_hash': bench.source_hash(), 'tag': '$TAG'}, open('$OUT/meta.json', 'w'))"
python $ROOT/bench.py --steps 2000 --warmup 200 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err      # the driver's block length
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 300 --warmup 50 --cpu-sample 0 --sat-windows 0 --modes-budget 0 --shim-sample 0 > $OUT/trace.log 2>&1
# the same with one call in flight and no merged sequences: every k_linearize launch is ONE B=1 call's and has the chip to itself
# (what bench.py's `roofline` block measures; the default trace above also holds the 20-pair launches of the merged sequences)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes1 -- python $ROOT/bench.py --lanes 1 --coalesce 0 --steps 300 --warmup 50 --cpu-sample 0 --sat-windows 0 --modes-budget 0 --shim-sample 0 > $OUT/trace_lanes1.log 2>&1
# the chip-filling launch (32 windows = 64 directed pairs per call): what bench.py's `roofline_saturated` cites
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sat -- python $ROOT/scripts/sat_workload.py > $OUT/trace_sat.log 2>&1
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- python $ROOT/bench.py --lanes 1 --coalesce 0 --graph-replay 0 --steps 20 --warmup 5 --cpu-sample 0 --sat-windows 0 --modes-budget 0 --shim-sample 0 > $OUT/pmc_$tag.log 2>&1   # (plain launches under counter collection)
done
# rocprofv3's per-dispatch kernel traces are large (tens of MB) and nothing downstream reads them: only the stats travel back (gpurun merges <= 64 MiB)
find $OUT -name '*kernel_trace.csv' -delete
ls $OUT
