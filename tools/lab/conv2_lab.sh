#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/conv2lab
o=gpurun_out/conv2lab/out.txt; : > $o
python -m pytest tests/test_gpu_gemm.py -x -q -k "im2col" 2>&1 | tail -5 >> $o
MT_IM2COL_ANY=1 MT_CONV_WG64=0 LAB_SAVE=/tmp/dw_old.pt python tools/lab/conv2_lab.py >> $o 2>&1
MT_CONV_WG64=0 LAB_CMP=/tmp/dw_old.pt python tools/lab/conv2_lab.py >> $o 2>&1
MT_CONV_WG64=1 LAB_WHICH=wgrad LAB_CMP=/tmp/dw_old.pt python tools/lab/conv2_lab.py >> $o 2>&1
MT_CONV_WG64=2 LAB_WHICH=wgrad LAB_CMP=/tmp/dw_old.pt python tools/lab/conv2_lab.py >> $o 2>&1
grep -v amdgpu.ids $o
