#!/bin/bash
# ON THE GPU BOX: A/B of k_linearize's SSIM gradient forms (TCSFM_ADJOINT=0: round-3 neighbour-visiting pass B; 1: adjoint form) on ONE box:
# live in-kernel brackets (B=1 one call in flight, four in flight, chip-full), headline value, and the PMC VALU instruction count per wave.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/adjoint_ab
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for a in 0 1; do
  TCSFM_ADJOINT=$a python $ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_adj${a}_rep${rep}.json 2> $OUT/bench_adj${a}_rep${rep}.err
done
done
for a in 0 1; do
  export TCSFM_ADJOINT=$a
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_adj$a -- python3 $ROOT/bench.py --lanes 1 --graph-replay 0 --steps 20 --warmup 5 --cpu-sample 0 --sat-windows 0 --ring-mb 0 > $OUT/pmc_adj$a.log 2>&1
done
python3 - <<PY
import json, glob, csv, collections
for a in (0, 1):
    for rep in (1, 2):
        d = json.loads(open("$OUT/bench_adj%d_rep%d.json" % (a, rep)).read().strip().splitlines()[-1])
        r, s = d["roofline"], d["roofline_saturated"]
        print("ADJ=%d rep %d: value %.0f single %.0f | k_linearize in-kernel %.3f us (in flight %.3f) | chip-full %.2f us frac %.4f fp/s %.0f" % (
            a, rep, d["value"], d["single_stream"]["value"], r["avg_launch_us"], r["in_flight"]["avg_launch_us"], s["avg_launch_us"], s["frac"], s["frame_pairs_per_s"]))
    f = glob.glob("$OUT/pmc_adj%d/*/*counter_collection.csv" % a)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "k_linearize" in k:
            m = {c: sum(x) / len(x) for c, x in v.items()}
            print("ADJ=%d PMC %s: VALU/wave %.1f LDS/wave %.1f SALU/wave %.1f wave_cycles/wave %.0f wait_any/wave %.0f" % (
                a, k[:60], m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_INSTS_LDS"] / m["SQ_WAVES"], m["SQ_INSTS_SALU"] / m["SQ_WAVES"], m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"], m["SQ_WAIT_ANY"] / m["SQ_WAVES"]))
PY
