#!/bin/bash
# Where the device idles inside a step: kernel + memory-copy + HIP runtime trace of the default bench step, then tools/lab/gap_report.py.
# Usage (GPU box): tools/lab/gap_trace.sh <outdir-under-gpurun_out> [ENV=VAL ...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
rm -rf $out; mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $out/t -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/line.json 2>$out/err.txt)
ls $out/t | head
python $GRAFT_REPO_ROOT/tools/lab/gap_report.py $out/t > $out/gaps.txt 2>&1
head -c 6000 $out/gaps.txt
rm -f $out/t/*hip_api_trace.csv
