#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats + PMC passes (separate runs) of the non-headline modes (scripts/mode_workload.py).
#   bash scripts/collect_mode_profiles.sh <tag> <mode> [<mode> ...]      outputs: gpurun_out/<tag>_<mode>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for MODE in "$@"; do
  OUT=$ROOT/gpurun_out/${TAG}_$MODE
  mkdir -p $OUT
  if [ "$MODE" = "sat" ]; then WL="$ROOT/scripts/sat_workload.py"; else WL="$ROOT/scripts/mode_workload.py $MODE"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $WL > $OUT/trace.log 2>&1
  for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    t=$(echo $P | cut -d' ' -f1)
    rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$t -- python $WL > $OUT/pmc_$t.log 2>&1 || echo "pass $t of $MODE failed"
  done
  echo "$MODE done"
done
