#!/bin/bash
# ON THE GPU BOX: kernel trace + two PMC passes of the PoseNet forward on N images (default 32): per-kernel time, VALU / SALU / MFMA-busy per wave
N=${1:-32}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/posenet_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/scripts/posenet_profile.py $N 20 > $OUT/trace.log 2>&1
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  t=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_$t -- python $ROOT/scripts/posenet_profile.py $N 5 > $OUT/pmc_$t.log 2>&1
done
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("%-60s calls %5s avg %8.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "k_pn_conv" not in k: continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w = m["SQ_WAVES"]
    print("%-40s waves %6.0f | per wave: VALU %6.0f SALU %6.0f VMEM_RD %5.0f LDS %5.0f | wave cycles %7.0f wait_any %6.0f wait_inst %6.0f | MFMA busy %.3f of kernel cycles, waves/SIMD resident %.2f" % (
        k[:40], w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_VMEM_RD"] / w, m["SQ_INSTS_LDS"] / w, 4 * m["SQ_WAVE_CYCLES"] / w, 4 * m["SQ_WAIT_ANY"] / w, 4 * m["SQ_WAIT_INST_ANY"] / w,
        m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), 4 * m["SQ_WAVE_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)))
PY
