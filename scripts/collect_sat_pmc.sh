#!/bin/bash
# PMC passes of the chip-filling workload (scripts/sat_workload.py); run on the GPU box, outputs in gpurun_out/<tag>/
set -e
TAG=${1:-sat}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
         "SQ_IFETCH SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "GRBM_GUI_ACTIVE"; do
  tag=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$tag -- python $ROOT/scripts/sat_workload.py > $OUT/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
ls $OUT
