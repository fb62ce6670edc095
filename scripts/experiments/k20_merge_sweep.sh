#!/bin/bash
# The driver's command is `bench.py --steps 20 --warmup 5`: 20 windows per timed block.  How the block's fixed cost (first enqueue, the
# lock-step solves of two equal sequences, the drain) depends on how the 20 queued calls are cut into merged sequences and streams.
# usage: bash scripts/experiments/k20_merge_sweep.sh   (writes gpurun_out/k20_sweep.txt)
export TCSFM_SET_ENV_DEFAULTS=1
out=gpurun_out/k20_sweep.txt; : > $out
for cfg in "10 2" "5 2" "7 2" "4 2" "5 4" "4 4" "10 1" "20 1" "3 3"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --coalesce $1 --coalesce-lanes $2 --modes-budget 0 --shim-sample 0 --cpu-sample 0 --sat-windows 0 > gpurun_out/k20_tmp.json 2>/dev/null || exit 1
    python - "$1" "$2" >> $out <<'PY'
import json,sys
d=json.loads(open('gpurun_out/k20_tmp.json').read().strip().splitlines()[-1])
print(f"coalesce {sys.argv[1]:>2s} streams {sys.argv[2]}  value {d['value']:9.1f}  ms/step {d['ms_per_step']:.5f}  timed_as {d['config'].get('timed_as','?')[:70]}")
PY
  done
done
cat $out
