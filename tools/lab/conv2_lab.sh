#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/conv2lab
o=gpurun_out/conv2lab/out.txt; : > $o
python - >> $o 2>&1 <<'PY'
import importlib, sys, torch
sys.path.insert(0, ".")
pkg = importlib.import_module("mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd")
L = importlib.import_module(pkg.__name__ + ".lib")
lib = L.get(); torch.zeros(1, device="cuda")
import ctypes
f = lib.mt_debug_wg64_blocks_per_cu; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_int]
print("blocks per CU: minw3", f(3), "minw2", f(2), "minw4", f(4))
PY
MT_CONV_WG64=4 LAB_WHICH=wgrad python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
MT_CONV_WG64=1 LAB_WHICH=wgrad python tools/lab/conv2_lab.py 2>&1 | grep -v amdgpu.ids >> $o
cat $o
