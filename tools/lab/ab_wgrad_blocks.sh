#!/bin/bash
# lab: block target of the plane weight gradients' K-range split (MT_WGRAD_BLOCKS, default 640), config 3 then config 5, one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wgb; o=gpurun_out/wgb/out.txt; : > $o
for v in 640 320 192 256 448 128 320 640; do
  MT_WGRAD_BLOCKS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 blocks $v', d['ms_per_step'], d['value'])" >> $o
done
for v in 256 192; do
  MT_WGRAD_BLOCKS=$v python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 blocks $v', d['ms_per_step'], d['value'])" >> $o
done
cat $o
