#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench (the evidence set's in-step timeline and kernel stats), more steps so the
# host is well ahead of the device by the step that is analysed.
out=$GRAFT_REPO_ROOT/gpurun_out/final; rm -rf $out/bench_stats; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/r03_bench_b32_line_under_rocprof.json 2>$out/bench_rocprof.err)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $out/bench_stats -name "*kernel_trace.csv") > $out/r03_bench_b32_in_step_timeline.txt 2>&1
cp $(find $out/bench_stats -name "*kernel_stats.csv" | head -1) $out/r03_bench_b32_kernel_stats.csv
head -4 $out/r03_bench_b32_in_step_timeline.txt
