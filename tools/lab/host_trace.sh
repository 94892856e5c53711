#!/bin/bash
# Host + device timeline of a few bench steps: kernel, memory-copy and HIP runtime API traces (no counters).
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/host_trace; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $out -o h -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras > $out/line.json 2> $out/err.txt
ls -la $out | head; tail -c 300 $out/line.json
cd $out && for f in h_hip_api_trace.csv h_kernel_trace.csv h_memory_copy_trace.csv; do [ -f $f ] && gzip -f $f; done; ls -la
