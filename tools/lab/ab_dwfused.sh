#!/bin/bash
# MT_DW_FUSED = 0 / 3 / 1 (depthwise data + weight gradient from one pass: off / 3x3 layers / all): serialised kernel time of the
# extractor step per family, then interleaved bench lines.
cd $GRAFT_REPO_ROOT
bash tools/lab/ab_ef_multi.sh MT_DW_FUSED "dwconv" 0 3 1
for r in 1 2; do for v in 0 3 1; do echo -n "MT_DW_FUSED=$v "; MT_DW_FUSED=$v python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done; done
