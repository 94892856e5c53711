#!/bin/bash
# Kernel-by-kernel timeline of one TimeSformer forward+backward (B=32, side stream off so launches serialise).
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o tsf -- python $GRAFT_REPO_ROOT/tools/perf_tsf.py --bwd --iters 1 2>&1 | grep "B="
