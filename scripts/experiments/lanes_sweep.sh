#!/bin/bash
# ON THE GPU BOX: headline value against the number of calls in flight (bench.py --lanes N), at the driver's K = 20 and in long blocks
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lanes_sweep
mkdir -p $OUT
for L in 3 4 5 6 8; do
  for K in 20 1000; do
    python $ROOT/bench.py --lanes $L --steps $K --warmup 50 --cpu-sample 0 --sat-windows 0 --ring-mb 120 > $OUT/l${L}_k${K}.json 2> $OUT/l${L}_k${K}.err
    python3 -c "
import json
d = json.loads(open('$OUT/l${L}_k${K}.json').read().strip().splitlines()[-1])
print('lanes $L K $K: value %.0f (%s) other %.0f in-flight k_linearize %.2f us host enqueue %.1f us' % (d['value'], d['launch_mode']['timed'][:12], d['launch_mode']['other_mode']['value'], d['roofline']['in_flight']['avg_launch_us'], d['host_enqueue_us_per_step']))"
  done
done
