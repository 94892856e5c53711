#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/conv2lab
o=gpurun_out/conv2lab/out.txt; : > $o
python -m pytest tests/test_gpu_gemm.py -x -q -k "im2col" 2>&1 | tail -2 >> $o
MT_CONV_WG64=1 python -m pytest tests/test_gpu_gemm.py -x -q -k "im2col" 2>&1 | tail -2 >> $o
MT_CONV_WG64=1 LAB_WHICH=wgrad LAB_SAVE=/tmp/dw1.pt python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
LAB_WHICH=wgrad LAB_CMP=/tmp/dw1.pt python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
LAB_N=128 MT_WGRAD_BLOCKS_OLD=256 LAB_WHICH=wgrad python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
LAB_N=256 MT_WGRAD_BLOCKS_OLD=512 LAB_WHICH=wgrad python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
for f in 0.7 1.0; do
  MT_XC_CONV2_WGRAD_SIDE=$f python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 side $f', d['ms_per_step'], d['value'])" >> $o
done
MT_CONV_WG64=1 python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 six-wave tile', d['ms_per_step'], d['value'])" >> $o
cat $o
