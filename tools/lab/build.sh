#!/bin/bash
# Builds the GEMM tuning lab against the in-tree library (gfx950 cross-compile works without a GPU).
set -e
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc
make -s -C "$csrc" -j 16
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result "$here/gemm_lab.hip" -o "$here/gemm_lab" -L"$csrc" -lmintime_hip -Wl,-rpath,'$ORIGIN/../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc'
echo built "$here/gemm_lab"
