// Xception-specific kernels (config 5 extractor, reference models/xception.py), gfx950.
// Everything dense (conv1/conv2 via the im2col prologue, pointwise and strided-skip 1x1 convs) is mt_gemm; the depthwise
// 3x3s are the EfficientNet depthwise kernels with a ReLU / identity input activation.  What is left:
//   mt_conv_weight_pack / _unpack_grad   torch [Co,Ci,k,k] <-> the GEMM's [Co][(kh,kw,ci)] operand layout (and the flipped,
//                                        transposed layout the data-gradient convolution needs)
//   mt_maxpool_add_fwd / _bwd            Block tail: MaxPool2d(3,2,1)(bn(z)) + skipbn(skip(inp))        (xception.py:64-79)
//   mt_bn_bwd_apply                      dz = ka*du + kb*z + kc materialised (only for the dense conv2, whose data gradient
//                                        is itself an im2col GEMM)
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "det.hpp"
#include <float.h>

using namespace mt;

namespace {

// out[r][(kh,kw,c)] (row pitch ld, zero padded):
//   transpose == 0: r = co, c = ci, value w[co][ci][kh][kw]                       (forward / wgrad operand)
//   transpose == 1: r = ci, c = co, value w[co][ci][k-1-kh][k-1-kw]               (data-gradient operand: flipped kernel)
__global__ void conv_weight_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int k, int ld,
                                        int transpose) {
  const int R = transpose ? Ci : Co, Cc = transpose ? Co : Ci;
  const int64_t total = (int64_t)R * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld), col = (int)(i % ld);
    float v = 0.f;
    if (col < k * k * Cc) {
      const int tap = col / Cc, c = col % Cc;
      const int kh = tap / k, kw = tap % k;
      if (!transpose) v = w[(((int64_t)r * Ci + c) * k + kh) * k + kw];
      else v = w[(((int64_t)c * Ci + r) * k + (k - 1 - kh)) * k + (k - 1 - kw)];
    }
    out[i] = v;
  }
}

// dw[co][ci][kh][kw] += dwp[co][(kh,kw,ci)]
__global__ void conv_weight_unpack_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int k, int ld) {
  const int64_t total = (int64_t)Co * Ci * k * k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kw = (int)(i % k);
    int64_t t = i / k;
    const int kh = (int)(t % k); t /= k;
    const int ci = (int)(t % Ci);
    const int co = (int)(t / Ci);
    dw[i] += dwp[(int64_t)co * ld + (kh * k + kw) * Ci + ci];
  }
}

// y[n,oh,ow,c] = max over the 3x3 stride-2 pad-1 window of (z*scale+shift)  +  (zs*scale_s+shift_s)
__global__ __launch_bounds__(256) void maxpool_add_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ zs,
                                                              const float* __restrict__ scale_s, const float* __restrict__ shift_s,
                                                              float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  const int CQ = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * CQ;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    int64_t t = i / CQ;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 + kh - 1;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 + kw - 1;
        if (iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(z + (((int64_t)n * H + ih) * W + iw) * C + cq * 4);
        m.x = fmaxf(m.x, fmaf(v.x, sc.x, sh.x)); m.y = fmaxf(m.y, fmaf(v.y, sc.y, sh.y));
        m.z = fmaxf(m.z, fmaf(v.z, sc.z, sh.z)); m.w = fmaxf(m.w, fmaf(v.w, sc.w, sh.w));
      }
    }
    const float4 s = reinterpret_cast<const float4*>(zs)[i];
    const float4 ss = reinterpret_cast<const float4*>(scale_s)[cq], hs = reinterpret_cast<const float4*>(shift_s)[cq];
    m.x += fmaf(s.x, ss.x, hs.x); m.y += fmaf(s.y, ss.y, hs.y); m.z += fmaf(s.z, ss.z, hs.z); m.w += fmaf(s.w, ss.w, hs.w);
    reinterpret_cast<float4*>(y)[i] = m;
  }
}

// du[n,ih,iw,c] += dy[n,oh,ow,c] at the window's arg-max (first maximum in row-major window order, like torch); du pre-zeroed
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ du, int N, int H, int W, int C, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float sc = scale[c], sh = shift[c];
    float best = -FLT_MAX;
    int64_t arg = -1;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 + kh - 1;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 + kw - 1;
        if (iw < 0 || iw >= W) continue;
        const int64_t off = (((int64_t)n * H + ih) * W + iw) * C + c;
        const float v = fmaf(z[off], sc, sh);
        if (v > best) { best = v; arg = off; }
      }
    }
    if (arg >= 0) atomicAdd(du + arg, dy[i]);
  }
}

// Deterministic mode: the same routing as a gather.  One thread per INPUT element visits the (at most four) windows that contain it in
// (oh, ow) order, recomputes each window's arg-max and adds dy where the arg-max is this element -- every du element has one writer.
__global__ __launch_bounds__(256) void maxpool_bwd_gather_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float* __restrict__ du, int N, int H, int W, int C, int Ho, int Wo) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int iw0 = (int)(t % W); t /= W;
    const int ih0 = (int)(t % H);
    const int n = (int)(t / H);
    const float sc = scale[c], sh = shift[c];
    float acc = 0.f;
    // windows oh with ih0 in {2 oh - 1, 2 oh, 2 oh + 1}: even ih0 -> oh = ih0 / 2; odd -> (ih0 - 1) / 2 and (ih0 + 1) / 2
    const int oh_lo = ih0 >> 1, oh_hi = (ih0 + 1) >> 1, ow_lo = iw0 >> 1, ow_hi = (iw0 + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      if (oh >= Ho) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        if (ow >= Wo) continue;
        float best = -FLT_MAX;
        int64_t arg = -1;
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = oh * 2 + kh - 1;
          if (ih < 0 || ih >= H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 + kw - 1;
            if (iw < 0 || iw >= W) continue;
            const int64_t off = (((int64_t)n * H + ih) * W + iw) * C + c;
            const float v = fmaf(z[off], sc, sh);
            if (v > best) { best = v; arg = off; }
          }
        }
        if (arg == i) acc += dy[(((int64_t)n * Ho + oh) * Wo + ow) * C + c];
      }
    }
    du[i] += acc;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ du, const float* __restrict__ z,
                                                           const float* __restrict__ kabc, float* __restrict__ dz, int64_t total4,
                                                           int C) {
  const int CQ = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    const float4 a = reinterpret_cast<const float4*>(du)[i], b = reinterpret_cast<const float4*>(z)[i];
    const float4 ka = reinterpret_cast<const float4*>(kabc)[cq], kb = reinterpret_cast<const float4*>(kabc + C)[cq],
                 kc = reinterpret_cast<const float4*>(kabc + 2 * C)[cq];
    reinterpret_cast<float4*>(dz)[i] = make_float4(fmaf(ka.x, a.x, fmaf(kb.x, b.x, kc.x)), fmaf(ka.y, a.y, fmaf(kb.y, b.y, kc.y)),
                                                   fmaf(ka.z, a.z, fmaf(kb.z, b.z, kc.z)), fmaf(ka.w, a.w, fmaf(kb.w, b.w, kc.w)));
  }
}

int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mt_conv_weight_pack(const float* w, float* out, int Co, int Ci, int k, int ld, int transpose, void* stream) {
  if (!w || !out) return fail(MT_ERR_ARG, "mt_conv_weight_pack: null pointer");
  const int cols = k * k * (transpose ? Co : Ci);
  if (ld < cols || (ld & 3)) return fail(MT_ERR_ARG, "mt_conv_weight_pack: ld %d must be >= %d and a multiple of 4", ld, cols);
  const int64_t total = (int64_t)(transpose ? Ci : Co) * ld;
  hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, out, Co, Ci, k, ld, transpose);
  return check_launch("mt_conv_weight_pack");
}

extern "C" int mt_conv_weight_unpack_grad(const float* dwp, float* dw, int Co, int Ci, int k, int ld, void* stream) {
  if (!dwp || !dw) return fail(MT_ERR_ARG, "mt_conv_weight_unpack_grad: null pointer");
  hipLaunchKernelGGL(conv_weight_unpack_grad_kernel, dim3(grid_for((int64_t)Co * Ci * k * k)), dim3(256), 0, (hipStream_t)stream, dwp,
                     dw, Co, Ci, k, ld);
  return check_launch("mt_conv_weight_unpack_grad");
}

extern "C" int mt_maxpool_add_fwd(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                                  const float* shift_s, float* y, int N, int H, int W, int C, void* stream) {
  if (!z || !scale || !shift || !zs || !scale_s || !shift_s || !y) return fail(MT_ERR_ARG, "mt_maxpool_add_fwd: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_maxpool_add_fwd: C %% 4 != 0");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_add_fwd_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, z, scale,
                     shift, zs, scale_s, shift_s, y, N, H, W, C, Ho, Wo);
  return check_launch("mt_maxpool_add_fwd");
}

extern "C" int mt_maxpool_bwd(const float* dy, const float* z, const float* scale, const float* shift, float* du, int N, int H,
                              int W, int C, void* stream) {
  if (!dy || !z || !scale || !shift || !du) return fail(MT_ERR_ARG, "mt_maxpool_bwd: null pointer");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (det_enabled()) {
    hipLaunchKernelGGL(maxpool_bwd_gather_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, z, scale,
                       shift, du, N, H, W, C, Ho, Wo);
    return check_launch("mt_maxpool_bwd(deterministic)");
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, dy, z, scale, shift,
                     du, N, H, W, C, Ho, Wo);
  return check_launch("mt_maxpool_bwd");
}

extern "C" int mt_bn_bwd_apply(const float* du, const float* z, const float* kabc, float* dz, int64_t rows, int C, void* stream) {
  if (!du || !z || !kabc || !dz) return fail(MT_ERR_ARG, "mt_bn_bwd_apply: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_bn_bwd_apply: C %% 4 != 0");
  const int64_t total4 = rows * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, du, z, kabc, dz, total4, C);
  return check_launch("mt_bn_bwd_apply");
}
