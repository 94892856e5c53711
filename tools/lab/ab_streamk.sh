#!/bin/bash
# lab: stream-K for the plane GEMMs (MT_PLANES_STREAMK): EfficientNet step alone and the whole step, A/B/A in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/streamk; o=gpurun_out/streamk/out.txt; : > $o
for v in 0 1 0 1; do
  echo "== MT_PLANES_STREAMK=$v" >> $o
  MT_PLANES_STREAMK=$v python tools/perf_ef.py --bwd --iters 20 2>&1 | grep crops= >> $o
done
for v in 0 1 0; do
  MT_PLANES_STREAMK=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench streamk $v', d['ms_per_step'], d['phases'])" >> $o
done
cat $o
