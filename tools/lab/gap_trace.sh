#!/bin/bash
# Kernel trace of a few bench steps -> where the device idles (tools/lab/gap_list.py) and the in-step time per symbol.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/gaps/now; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras > $out/line.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/lab/gap_list.py $(find $out -name "*kernel_trace.csv") | head -8
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $out -name "*kernel_trace.csv") | head -${1:-30}
