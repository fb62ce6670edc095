#!/bin/bash
# ON THE GPU BOX: merged sequences (tcsfm_refine_window_queued) alternating over 1..4 streams of the handle, for several sequence lengths;
# K=20 blocks (the driver's) and K=200.  One line per setting: merged value, lanes value, single stream.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/coalesce_lanes
mkdir -p $OUT
for K in 20 200; do
for cfg in "10 1" "10 2" "10 3" "10 4" "5 2" "5 4" "16 2" "20 2"; do
  set -- $cfg
  python $ROOT/bench.py --steps $K --warmup 5 --cpu-sample 0 --sat-windows 0 --coalesce $1 --coalesce-lanes $2 > $OUT/b_${K}_$1_$2.json 2> $OUT/b_${K}_$1_$2.err
  python3 - <<PY
import json
d = json.loads(open("$OUT/b_${K}_$1_$2.json").read().strip().splitlines()[-1])
m = d["launch_mode"]["merged"]
print("K=$K calls/sequence %2d streams %d: merged %.0f (same poses %s) | lanes %.0f | single %.0f" % (m["calls_per_sequence"], m["streams"], m["value"], m["same_poses"], d["launch_mode"]["lanes"]["value"], d["single_stream"]["value"]))
PY
done
done
