#!/bin/bash
# lab: Xception pointwise convolutions on plane operands from fewer input channels (MT_XC_PLANES_MIN_C), config 5, A/B/A in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/xcminc; o=gpurun_out/xcminc/out.txt; : > $o
python -m pytest tests/test_gpu_planes.py -q -x -k "gigabytes" 2>&1 | tail -2 >> $o
python -m pytest tests/test_gpu_e2e.py -q -x -k "config5_full_size" 2>&1 | tail -3 >> $o
for v in 256 128 256 128; do
  MT_XC_PLANES_MIN_C=$v python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('min_c $v', d['ms_per_step'], d['value'])" >> $o
done
cat $o
