#!/bin/bash
# In-step timeline of the default bench.py step under rocprofv3 (kernel trace only).  Usage (GPU box): tools/lab/in_step_timeline.sh <outdir-under-gpurun_out> [ENV=VAL ...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/line.json 2>$out/err.txt)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $out/stats -name "*kernel_trace.csv") > $out/timeline.txt 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
rm -rf $out/stats
