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
#include "planes.hpp"
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
// `arg` / `zmax` (nullable, pooled shape): the window position kh*3+kw of the arg-max (first maximum in row-major window order, like
// torch; 255 when no element compared greater than -FLT_MAX) and the raw z there -- what the adjoint needs instead of z's windows
template <bool ARG>
__global__ __launch_bounds__(256) void maxpool_add_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ zs,
                                                              const float* __restrict__ scale_s, const float* __restrict__ shift_s,
                                                              float* __restrict__ y, uint8_t* __restrict__ arg, float* __restrict__ zmax,
                                                              int N, int H, int W, int C, int Ho, int Wo) {
  const int CQ = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * CQ;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    int64_t t = i / CQ;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
    float m[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX}, zr[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t am[4] = {255u, 255u, 255u, 255u};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 + kh - 1;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 + kw - 1;
        if (iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(z + (((int64_t)n * H + ih) * W + iw) * C + cq * 4);
        const float raw[4] = {v.x, v.y, v.z, v.w};
        const float u[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (u[e] > m[e]) { m[e] = u[e]; if (ARG) { am[e] = (uint32_t)(kh * 3 + kw); zr[e] = raw[e]; } }
      }
    }
    const float4 s = reinterpret_cast<const float4*>(zs)[i];
    const float4 ss = reinterpret_cast<const float4*>(scale_s)[cq], hs = reinterpret_cast<const float4*>(shift_s)[cq];
    reinterpret_cast<float4*>(y)[i] = make_float4(m[0] + fmaf(s.x, ss.x, hs.x), m[1] + fmaf(s.y, ss.y, hs.y), m[2] + fmaf(s.z, ss.z, hs.z),
                                                  m[3] + fmaf(s.w, ss.w, hs.w));
    if (ARG) {
      reinterpret_cast<uint32_t*>(arg)[i] = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
      reinterpret_cast<float4*>(zmax)[i] = make_float4(zr[0], zr[1], zr[2], zr[3]);
    }
  }
}

// Four channels' share of the pooled gradient that lands on input pixel (n, ih, iw): the (at most four) windows that contain the pixel,
// visited in (oh, ow) order; a window contributes where its recorded arg-max position is this pixel.  One writer per element: no
// zero fill, no atomics, the same bits every run.
__device__ __forceinline__ float4 pool_route4(const float* __restrict__ dy, const uint8_t* __restrict__ arg, int n, int ih, int iw, int c,
                                              int Ho, int Wo, int C) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1, ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    if (oh >= Ho) continue;
    const int kh = ih - 2 * oh + 1;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      if (ow >= Wo) continue;
      const uint32_t k = (uint32_t)(kh * 3 + iw - 2 * ow + 1);
      const int64_t off = (((int64_t)n * Ho + oh) * Wo + ow) * C + c;
      const uint32_t mm = *reinterpret_cast<const uint32_t*>(arg + off) ^ (k * 0x01010101u);   // a zero byte = this pixel is the arg-max
      if (((mm - 0x01010101u) & ~mm & 0x80808080u) == 0u) continue;
      const float4 d = *reinterpret_cast<const float4*>(dy + off);
      if ((mm & 0xffu) == 0u) acc.x += d.x;
      if ((mm & 0xff00u) == 0u) acc.y += d.y;
      if ((mm & 0xff0000u) == 0u) acc.z += d.z;
      if ((mm & 0xff000000u) == 0u) acc.w += d.w;
    }
  }
  return acc;
}

// du[n,ih,iw,c] = the routed gradient, every element written
__global__ __launch_bounds__(256) void maxpool_bwd_arg_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                              float* __restrict__ du, int N, int H, int W, int C, int Ho, int Wo) {
  const int CQ = C >> 2;
  const int64_t total = (int64_t)N * H * W * CQ;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    int64_t t = i / CQ;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    reinterpret_cast<float4*>(du)[i] = pool_route4(dy, arg, n, ih, iw, cq * 4, Ho, Wo, C);
  }
}

// dz = ka * du + kb * z + kc with du routed on the fly, written as operand planes (the pooled block's last pointwise convolution takes
// its data and weight gradient from them): du never exists in memory
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_planes_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                                          const float* __restrict__ z, const float* __restrict__ kabc,
                                                                          int R, int H, int W, int C, int Ho, int Wo, int ncg,
                                                                          PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.x / ncg, cb = (blockIdx.x % ncg) * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    const int iw = r % W, ih = (r / W) % H, n = r / (W * H);
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      if (c + e + 4 <= C) {
        const float4 a = pool_route4(dy, arg, n, ih, iw, c + e, Ho, Wo, C), b = *reinterpret_cast<const float4*>(z + (int64_t)r * C + c + e);
        const float4 ka = *reinterpret_cast<const float4*>(kabc + c + e), kb = *reinterpret_cast<const float4*>(kabc + C + c + e),
                     kc = *reinterpret_cast<const float4*>(kabc + 2 * C + c + e);
        x[e] = fmaf(ka.x, a.x, fmaf(kb.x, b.x, kc.x)); x[e + 1] = fmaf(ka.y, a.y, fmaf(kb.y, b.y, kc.y));
        x[e + 2] = fmaf(ka.z, a.z, fmaf(kb.z, b.z, kc.z)); x[e + 3] = fmaf(ka.w, a.w, fmaf(kb.w, b.w, kc.w));
      }
    }
  }
  planes_store8(o, r, c, x);
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

static int maxpool_add_fwd(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                           const float* shift_s, float* y, uint8_t* arg, float* zmax, int N, int H, int W, int C, void* stream,
                           const char* who) {
  if (!z || !scale || !shift || !zs || !scale_s || !shift_s || !y) return fail(MT_ERR_ARG, "%s: null pointer", who);
  if (C & 3) return fail(MT_ERR_ARG, "%s: C %% 4 != 0", who);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const dim3 grid(grid_for((int64_t)N * Ho * Wo * (C / 4)));
  if (arg)
    hipLaunchKernelGGL(maxpool_add_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, scale, shift, zs, scale_s, shift_s, y, arg,
                       zmax, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_add_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, scale, shift, zs, scale_s, shift_s, y, arg,
                       zmax, N, H, W, C, Ho, Wo);
  return check_launch(who);
}

extern "C" int mt_maxpool_add_fwd(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                                  const float* shift_s, float* y, int N, int H, int W, int C, void* stream) {
  return maxpool_add_fwd(z, scale, shift, zs, scale_s, shift_s, y, nullptr, nullptr, N, H, W, C, stream, "mt_maxpool_add_fwd");
}

extern "C" int mt_maxpool_add_fwd_arg(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                                      const float* shift_s, float* y, uint8_t* arg, float* zmax, int N, int H, int W, int C,
                                      void* stream) {
  if (!arg || !zmax) return fail(MT_ERR_ARG, "mt_maxpool_add_fwd_arg: null pointer");
  return maxpool_add_fwd(z, scale, shift, zs, scale_s, shift_s, y, arg, zmax, N, H, W, C, stream, "mt_maxpool_add_fwd_arg");
}

extern "C" int mt_maxpool_bwd_arg(const float* dy, const uint8_t* arg, float* du, int N, int H, int W, int C, void* stream) {
  if (!dy || !arg || !du) return fail(MT_ERR_ARG, "mt_maxpool_bwd_arg: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_maxpool_bwd_arg: C %% 4 != 0");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_arg_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, arg, du,
                     N, H, W, C, Ho, Wo);
  return check_launch("mt_maxpool_bwd_arg");
}

extern "C" int mt_maxpool_bn_bwd_apply_planes(const float* dy, const uint8_t* arg, const float* z, const float* kabc, void* planes, int N,
                                              int H, int W, int C, void* stream) {
  if (!dy || !arg || !z || !kabc || !planes) return fail(MT_ERR_ARG, "mt_maxpool_bn_bwd_apply_planes: null pointer");
  if ((C & 3) || ((uintptr_t)planes & 15)) return fail(MT_ERR_ARG, "mt_maxpool_bn_bwd_apply_planes: C %% 4 == 0 and 16-byte alignment");
  const int64_t rows = (int64_t)N * H * W;
  if (rows > INT32_MAX - 32) return fail(MT_ERR_ARG, "mt_maxpool_bn_bwd_apply_planes: too many rows");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int cb16 = (C + 15) >> 4, rp = ((int)rows + 31) & ~31;
  const PlaneRef o{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
  const int ncg = (cb16 + 3) / 4;
  hipLaunchKernelGGL(maxpool_bn_bwd_apply_planes_kernel, dim3((unsigned)((int64_t)ncg * (rp / 32))), dim3(256), 0, (hipStream_t)stream, dy,
                     arg, z, kabc, (int)rows, H, W, C, Ho, Wo, ncg, o);
  return check_launch("mt_maxpool_bn_bwd_apply_planes");
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
