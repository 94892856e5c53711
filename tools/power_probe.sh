#!/bin/bash
# Samples power / clocks with rocm-smi every 0.2 s while bench.py runs its timed steps (is the step power-capped?).
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 120); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction\)" | tr '\n' ' ' ; echo; sleep 0.2; done ) > gpurun_out/power_samples.txt &
SMI=$!
python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/power_bench.json 2>/dev/null
kill $SMI 2>/dev/null
sed -n '20,60p' gpurun_out/power_samples.txt | cut -c1-260
rocm-smi --showmaxpower 2>/dev/null | grep -i power
