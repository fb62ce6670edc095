#!/bin/bash
# Diagnostic (profiles/r04_bench_repeats.jsonl): sample the shader clock, power and temperature twice a second while bench.py runs twice
# back to back -- does the in-kernel duration of k_linearize follow the clock?      bash scripts/diag/clock_sampler.sh  (on the GPU box)
mkdir -p gpurun_out
( while true; do echo "t=$(date +%s.%N)"; rocm-smi -d 0 --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +1; sleep 0.5; done ) > gpurun_out/clocks.log 2>&1 &
SAMPLER=$!
python bench.py > gpurun_out/clk_run1.json 2> gpurun_out/clk_run1.err
echo "t=$(date +%s.%N) END_RUN1" >> gpurun_out/clocks.log
python bench.py > gpurun_out/clk_run2.json 2> gpurun_out/clk_run2.err
echo "t=$(date +%s.%N) END_RUN2" >> gpurun_out/clocks.log
kill $SAMPLER
echo done
