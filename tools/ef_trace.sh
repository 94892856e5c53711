#!/bin/bash
# Kernel-by-kernel timeline of one EfficientNet-B0 forward+backward (256 crops, side stream off so launches serialise).
# Usage (on the GPU box): tools/ef_trace.sh <outdir-under-gpurun_out>
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
MT_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o ef -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 1 2>&1 | grep crops=
