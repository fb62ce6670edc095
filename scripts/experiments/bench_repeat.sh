#!/bin/bash
# value / single-stream value / block spread of bench.py over several fresh processes, under environment variants (one per argument,
# "-" = unchanged):   bash scripts/experiments/bench_repeat.sh <runs> <steps> "-" "GPU_MAX_HW_QUEUES=8" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; K=$2; shift; shift
for r in $(seq 1 $R); do
  for V in "$@"; do
    if [ "$V" = "-" ]; then E=""; else E="$V"; fi
    env $E python $ROOT/bench.py --steps $K --warmup 5 --cpu-sample 0 --sat-windows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$V', 'K', d['steps'], 'value', d['value'], 'single', d['single_stream']['value'], 'blocks', d['ms_per_step_blocks'], 'live_lin_us', d['roofline']['live']['avg_launch_us_in_kernel'], 'host_us', d.get('host_enqueue_us_per_step'), 'plain', (d.get('plain_launches') or {}).get('value'), 'plain_host_us', (d.get('plain_launches') or {}).get('host_enqueue_us_per_step'))"
  done
done
