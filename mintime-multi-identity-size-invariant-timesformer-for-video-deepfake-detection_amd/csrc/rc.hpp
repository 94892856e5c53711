// Expand convolution recomputed on load ("RC"): EfficientNet-B0's early MBConv blocks (reference efficientnet_pytorch/model.py:93-103)
// widen a 16- / 24-channel block input 6x with a 1x1 convolution and hand the result to the depthwise convolution.  Stored, that
// expanded tensor is the widest of the step (1.2 GB for block 1 of a 256-crop batch) and is read by three kernels (depthwise
// forward, its data gradient, its weight gradient).  The depthwise kernels work on 16-channel chunks of a pixel tile; a chunk of
// the expanded tensor is 16 x Cin MACs per pixel away from the block input, which is 6x narrower and shared by all chunks of the
// tile (their blocks run back to back on one XCD: common.hpp xcd_chunk_tile) -- so the kernels rebuild their chunk
//     z[pix][c0 + c] = sum_k y[pix][k] * We[c0 + c][k]
// on the matrix cores while filling their LDS tile, and the expanded tensor is never written.
//
// v_mfma_f32_16x16x4_f32 (exact fp32, == an fmaf chain): A = We chunk (M = 16 channels, lane l supplies channel l & 15),
// B = y (N = 16 pixels, lane l supplies pixel l & 15), k group kk = l >> 4.  K is walked in a lane-friendly order: MFMA j of a
// 16-wide k group g contracts k = 16 g + 4 kk + j, so a lane holds ONE float4 of its row per group (a 16-byte load of 4 consecutive
// k) and feeds component j to MFMA j; a trailing group of 8 (Cin = 24, 40) uses k = 16 G + 2 kk + j from a float2.
// Result: lane l holds z[pixel l & 15][c0 + 4 (l >> 4) + 0..3] -- a float4 of the NHWC tile, ready for the per-channel affine.
#pragma once
#include <hip/hip_runtime.h>

namespace mt {

typedef float rc_f32x4 __attribute__((ext_vector_type(4)));

template <int CIN>
struct RcFrag {
  static_assert(CIN % 8 == 0 && CIN >= 8, "block widths of EfficientNet-B0's early stages: 16, 24, 40");
  static constexpr int G4 = CIN / 16, G2 = (CIN % 16) / 8;
  float4 v4[G4 > 0 ? G4 : 1];
  float2 v2[G2 > 0 ? G2 : 1];
};

// row = first element of this lane's row (a pixel of y, or an output channel of We), 16-byte aligned, CIN floats long
template <int CIN>
__device__ __forceinline__ void rc_load(RcFrag<CIN>& f, const float* __restrict__ row, int lane) {
  const int kk = lane >> 4;
#pragma unroll
  for (int g = 0; g < RcFrag<CIN>::G4; ++g) f.v4[g] = *reinterpret_cast<const float4*>(row + 16 * g + 4 * kk);
  if constexpr (RcFrag<CIN>::G2 > 0) f.v2[0] = *reinterpret_cast<const float2*>(row + 16 * RcFrag<CIN>::G4 + 2 * kk);
}

// every lane of the wavefront must be active
template <int CIN>
__device__ __forceinline__ float4 rc_mma(const RcFrag<CIN>& w, const RcFrag<CIN>& y) {
  rc_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < RcFrag<CIN>::G4; ++g) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v4[g].x, y.v4[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v4[g].y, y.v4[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v4[g].z, y.v4[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v4[g].w, y.v4[g].w, acc, 0, 0, 0);
  }
  if constexpr (RcFrag<CIN>::G2 > 0) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v2[0].x, y.v2[0].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v2[0].y, y.v2[0].y, acc, 0, 0, 0);
  }
  return make_float4(acc[0], acc[1], acc[2], acc[3]);
}

}  // namespace mt
