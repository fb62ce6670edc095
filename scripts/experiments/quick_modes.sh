#!/bin/bash
# quick A/B of the hot kernels on the GPU box: rocprofv3 kernel stats of a few mode workloads -> one line per mode
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for MODE in "$@"; do
  OUT=$ROOT/gpurun_out/quick_$MODE
  rm -rf $OUT; mkdir -p $OUT
  if [ "$MODE" = "bench1" ]; then CMD="$ROOT/bench.py --lanes 1 --steps 300 --warmup 50 --cpu-sample 0 --sat-windows 0";
  elif [ "$MODE" = "sat" ]; then CMD="$ROOT/scripts/sat_workload.py";
  elif [ "$MODE" = "seqS2" ]; then CMD="$ROOT/scripts/seq_profile.py 2 1 8";
  elif [ "$MODE" = "seqS1" ]; then CMD="$ROOT/scripts/seq_profile.py 1 1 8";
  else CMD="$ROOT/scripts/mode_workload.py $MODE 40"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $CMD > $OUT.log 2>&1
  echo "== $MODE"; grep -h "k_linearize\|k_dense\|k_solve\|k_pack" $OUT/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
done
