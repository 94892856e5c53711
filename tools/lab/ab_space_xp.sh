#!/bin/bash
# lab: space-attention backward with P^T / dS^T handed to its second phase through LDS (MT_ATTN_SPACE_XP, default 1) vs recomputed (0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/xp; o=gpurun_out/xp/out.txt; : > $o
for v in 0 1 0 1 0 1; do
  MT_ATTN_SPACE_XP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 xp $v', d['ms_per_step'], d['value'])" >> $o
done
for v in 0 1; do
  MT_ATTN_SPACE_XP=$v python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 xp $v', d['ms_per_step'], d['value'])" >> $o
done
cat $o
