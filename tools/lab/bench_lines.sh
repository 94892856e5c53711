#!/bin/bash
# The four bench lines of the evidence set (no profiler), after profiles/r03_pmc_families.json is in place.
out=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $out; cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $out/r03_bench_b32_line.json 2>$out/bench.err
python bench.py --config 2 --steps 10 --warmup 5 --no-cpu-baseline > $out/r03_bench_config2_b16_line.json 2>/dev/null
python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline > $out/r03_bench_config5_xs_line.json 2>/dev/null
python bench.py --ragged --steps 10 --warmup 5 --no-cpu-baseline --no-extras > $out/r03_bench_b32_ragged_line.json 2>/dev/null
tail -c 400 $out/r03_bench_b32_line.json
