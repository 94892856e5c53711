#!/bin/bash
# Kernel-by-kernel timeline of the EfficientNet step (256 crops, side stream off) under tile-variant knobs.
# Usage (GPU box): tools/lab/ef_variants.sh <outdir-under-gpurun_out> "ENV=VAL ..." ["ENV=VAL ..." ...]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p $out
i=0
for envs in "" "$@"; do
  d=$out/v$i
  mkdir -p $d
  echo "== variant $i: $envs" > $out/v$i.txt
  env MT_SIDE_STREAM=0 MT_PLAN=0 $envs timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o ef -- python $GRAFT_REPO_ROOT/tools/perf_ef.py --bwd --iters 1 2>&1 | grep crops= >> $out/v$i.txt
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/ef_timeline.py $f --min 0 >> $out/v$i.txt 2>&1
  rm -rf $d
  i=$((i+1))
done
