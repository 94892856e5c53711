#!/bin/bash
# Builds the plane-GEMM tuning lab against the in-tree library (gfx950 cross-compile works without a GPU).
set -e
here=$(cd "$(dirname "$0")" && pwd)
csrc=$here/../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc
make -s -C "$csrc" -j 16
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result $EXTRA "$here/planes_lab.hip" -o "$here/planes_lab${SUFFIX}" -L"$csrc" -lmintime_hip -Wl,-rpath,'$ORIGIN/../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc'
echo built "$here/planes_lab${SUFFIX}"
