#!/bin/bash
# ON THE GPU BOX: bench lines of the current build (K=20, two repetitions) -> one summary line each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/quick_ab
mkdir -p $OUT
for rep in 1 2; do
  python $ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_rep${rep}.json 2> $OUT/bench_rep${rep}.err
  python3 - <<PY
import json
d = json.loads(open("$OUT/bench_rep${rep}.json").read().strip().splitlines()[-1])
r, s = d["roofline"], d["roofline_saturated"]
print("rep ${rep}: value %.0f single %.0f | k_linearize in-kernel %.3f us (in flight %.3f) | chip-full %.2f us frac %.4f fp/s %.0f" % (
    d["value"], d["single_stream"]["value"], r["avg_launch_us"], r["in_flight"]["avg_launch_us"], s["avg_launch_us"], s["frac"], s["frame_pairs_per_s"]))
PY
done
