// EfficientNet-B0 backward kernels other than the 1x1-convolution dgrad/wgrad (those are mt_gemm), gfx950.
//
// The reference has no hand-written backward (torch autograd over model.py:89-128, 267-288 and the custom swish
// backward utils.py:70-75).  Adjoint design, mirroring the forward's "raw z in HBM, activation on load" rule:
//   * BatchNorm backward is split into (1) a reduction of sum(du) and sum(du*xhat) per channel -- fused into whichever
//     kernel produces du --, (2) mt_bn_bwd_finalize, which turns them into three per-channel vectors so that
//         dz = ka[c]*du + kb[c]*z + kc[c]
//     and (3) consumers (GEMM operand loads, depthwise kernels) applying that affine while loading.  No dz tensor is
//     ever written for the wide (expanded) activations.
//   * swish'(u) = sig(u)*(1+u*(1-sig(u))) is recomputed from z (utils.py:70-75 saves only the input as well).
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include <type_traits>
#include "rc.hpp"
#include "det.hpp"
#include <stdlib.h>

using namespace mt;

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   /* v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the swish kernels are VALU-bound */
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float dswishf_(float u) { const float s = sigmoidf_(u); return s * (1.0f + u * (1.0f - s)); }

__device__ __forceinline__ float4 f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return f4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return f4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int ACT>
__device__ __forceinline__ float4 act4(float4 u) {
  if (ACT == 1) return f4(swishf_(u.x), swishf_(u.y), swishf_(u.z), swishf_(u.w));
  if (ACT == 2) return f4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
  return u;
}
template <int ACT>
__device__ __forceinline__ float4 dact4(float4 u) {     // derivative of the activation at pre-activation u
  if (ACT == 1) return f4(dswishf_(u.x), dswishf_(u.y), dswishf_(u.z), dswishf_(u.w));
  if (ACT == 2) return f4(u.x > 0.f ? 1.f : 0.f, u.y > 0.f ? 1.f : 0.f, u.z > 0.f ? 1.f : 0.f, u.w > 0.f ? 1.f : 0.f);
  return f4(1.f, 1.f, 1.f, 1.f);
}

// block-level reduction of per-thread (s1,s2) float4 partials over the PB row-lanes, then fp64 atomics
__device__ __forceinline__ void reduce_stats(float* red, float4 s1, float4 s2, int cql, int pl, int CQB, int PB, int CQ,
                                             int C, double* stats, int slots) {
  if (pl < PB) {
    float* rr = red + (pl * CQB + cql) * 8;
    rr[0] = s1.x; rr[1] = s1.y; rr[2] = s1.z; rr[3] = s1.w;
    rr[4] = s2.x; rr[5] = s2.y; rr[6] = s2.z; rr[7] = s2.w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CQB * 8; i += blockDim.x) {
    const int q = i >> 3, e = i & 7;
    const int cqq = blockIdx.y * CQB + q;
    if (cqq < CQ) {
      float v = 0.f;
      for (int p = 0; p < PB; ++p) v += red[(p * CQB + q) * 8 + e];
      const int ch = cqq * 4 + (e & 3), which = e >> 2;
      stat_add(stats + ((int64_t)(blockIdx.x % stat_slots(slots)) * 2 + which) * C + ch, stat_limb(slots, C), v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ K1: activation/BN adjoint + sums
// du = (d_in [*gate[n,c] + dpool[n,c]/hw] [*rowscale[n]]) * (act ? swish'(z*scale+shift) : 1)
// sums: S1[c] += du, S2[c] += du * xhat,  xhat = (z-mean)*invstd
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ din, const float* __restrict__ z,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean_invstd, const float* __restrict__ gate,
                                                         const float* __restrict__ dpool, const float* __restrict__ rowscale,
                                                         float* __restrict__ dout, double* __restrict__ stats, int slots,
                                                         int64_t rows, int C, int hw, int act, int CQB, int PB) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  const int cql = tid % CQB, pl = tid / CQB;
  const int CQ = C >> 2;
  const int cq = blockIdx.y * CQB + cql;
  float4 s1 = f4(0, 0, 0, 0), s2 = s1;
  if (pl < PB && cq < CQ) {
    const int c = cq * 4;
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    const float4 mean = ld4(mean_invstd + c), istd = ld4(mean_invstd + C + c);
    const float inv_hw = 1.0f / (float)hw;
    for (int64_t r = (int64_t)blockIdx.x * PB + pl; r < rows; r += (int64_t)gridDim.x * PB) {
      const int n = (int)(r / hw);
      float4 d = ld4(din + r * C + c);
      const float4 zz = ld4(z + r * C + c);
      if (gate) {
        const float4 g = ld4(gate + (int64_t)n * C + c), dp = ld4(dpool + (int64_t)n * C + c);
        d = f4(fmaf(d.x, g.x, dp.x * inv_hw), fmaf(d.y, g.y, dp.y * inv_hw), fmaf(d.z, g.z, dp.z * inv_hw), fmaf(d.w, g.w, dp.w * inv_hw));
      }
      if (rowscale) { const float rs = rowscale[n]; d = f4(d.x * rs, d.y * rs, d.z * rs, d.w * rs); }
      if (act == 1) {
        const float4 u = fma4(zz, sc, sh);
        d = f4(d.x * dswishf_(u.x), d.y * dswishf_(u.y), d.z * dswishf_(u.z), d.w * dswishf_(u.w));
      } else if (act == 2) {
        const float4 u = fma4(zz, sc, sh);
        d = f4(u.x > 0.f ? d.x : 0.f, u.y > 0.f ? d.y : 0.f, u.z > 0.f ? d.z : 0.f, u.w > 0.f ? d.w : 0.f);
      }
      if (dout) st4(dout + r * C + c, d);
      const float4 xh = f4((zz.x - mean.x) * istd.x, (zz.y - mean.y) * istd.y, (zz.z - mean.z) * istd.z, (zz.w - mean.w) * istd.w);
      s1 = add4(s1, d);
      s2 = fma4(d, xh, s2);
    }
  }
  reduce_stats(red, s1, s2, cql, pl, CQB, PB, CQ, C, stats, slots);
}

// ------------------------------------------------------------------------------------------------ K2: BN backward finalize
// block = 32 channels x 8 slot groups (see bn_finalize_kernel)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ stats, int slots, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_invstd,
                                                              float* __restrict__ kabc, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int C, int training) {
  __shared__ double red[8][32][2];
  const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = sg; i < stat_slots(slots); i += 8) {
      a += stat_get(stats + ((int64_t)i * 2) * C + c, stat_limb(slots, C));
      b += stat_get(stats + ((int64_t)i * 2 + 1) * C + c, stat_limb(slots, C));
    }
  red[sg][cl][0] = a; red[sg][cl][1] = b;
  __syncthreads();
  if (sg != 0 || c >= C) return;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int g = 0; g < 8; ++g) { s1 += red[g][cl][0]; s2 += red[g][cl][1]; }
  const float mean = mean_invstd[c], istd = mean_invstd[C + c];
  const float ka = gamma[c] * istd;
  float kb = 0.f, kc = 0.f;
  if (training) {
    const float c1 = (float)(s1 / count), c2 = (float)(s2 / count);
    kb = -ka * c2 * istd;
    kc = ka * c2 * istd * mean - ka * c1;
  }
  kabc[c] = ka; kabc[C + c] = kb; kabc[2 * C + c] = kc;
  if (dgamma) dgamma[c] += (float)s2;
  if (dbeta) dbeta[c] += (float)s1;
}

// ------------------------------------------------------------------------------------------------ K3: d(gate) reduction
// dgate[n,c] = sum_hw da[n,hw,c] * swish(z*scale+shift)
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ dgate, int HW, int C, int CQB, int PB) {
  extern __shared__ float red[];
  const int tid = threadIdx.x;
  const int cql = tid % CQB, pl = tid / CQB;
  const int cq = blockIdx.y * CQB + cql;
  const int CQ = C >> 2;
  const int n = blockIdx.x;
  float4 s = f4(0, 0, 0, 0);
  if (pl < PB && cq < CQ) {
    const float4 sc = ld4(scale + cq * 4), sh = ld4(shift + cq * 4);
    const int64_t base = (int64_t)n * HW * C + cq * 4;
    // blockIdx.z = slice of the image's pixels (one image per block leaves a 256-crop batch at one block per CU)
    const int per = (HW + gridDim.z - 1) / gridDim.z;
    const int p_lo = blockIdx.z * per, p_hi = min(HW, p_lo + per);
    int p = p_lo + pl;
    for (; p + PB < p_hi; p += 2 * PB) {      // two independent pixel pairs in flight
      const float4 z0 = ld4(z + base + (int64_t)p * C), z1 = ld4(z + base + (int64_t)(p + PB) * C);
      const float4 d0 = ld4(da + base + (int64_t)p * C), d1 = ld4(da + base + (int64_t)(p + PB) * C);
      const float4 u0 = fma4(z0, sc, sh), u1 = fma4(z1, sc, sh);
      s = f4(fmaf(d0.x, swishf_(u0.x), s.x), fmaf(d0.y, swishf_(u0.y), s.y), fmaf(d0.z, swishf_(u0.z), s.z), fmaf(d0.w, swishf_(u0.w), s.w));
      s = f4(fmaf(d1.x, swishf_(u1.x), s.x), fmaf(d1.y, swishf_(u1.y), s.y), fmaf(d1.z, swishf_(u1.z), s.z), fmaf(d1.w, swishf_(u1.w), s.w));
    }
    for (; p < p_hi; p += PB) {
      const float4 u = fma4(ld4(z + base + (int64_t)p * C), sc, sh);
      const float4 d = ld4(da + base + (int64_t)p * C);
      s = f4(fmaf(d.x, swishf_(u.x), s.x), fmaf(d.y, swishf_(u.y), s.y), fmaf(d.z, swishf_(u.z), s.z), fmaf(d.w, swishf_(u.w), s.w));
    }
  }
  if (pl < PB) st4(red + (pl * CQB + cql) * 4, s);
  __syncthreads();
  if (pl == 0 && cq < CQ) {
    float4 t = f4(0, 0, 0, 0);
    for (int p = 0; p < PB; ++p) t = add4(t, ld4(red + (p * CQB + cql) * 4));
    float* out = dgate + (int64_t)n * C + cq * 4;
    if (gridDim.z == 1) {
      st4(out, t);
    } else {                                   // slices meet in the pre-zeroed output
      atomicAdd(out + 0, t.x); atomicAdd(out + 1, t.y); atomicAdd(out + 2, t.z); atomicAdd(out + 3, t.w);
    }
  }
}

// ------------------------------------------------------------------------------------------------ K4: SE adjoint
// dpre2[n,c] = dgate*g*(1-g);  dh[n,j] = (sum_c dpre2[n,c] W2[c,j]) * swish'(hidden[n,j]);  dpooled[n,c] = sum_j dh[n,j] W1[j,c].
// One block per image walking W2 by column was latency-bound (77 us at C = 1152, 0.48 ms per step).  Now (1) one block per
// (image, 128-channel slab): the slab's W2 rows go coalesced into LDS ([128][CS+1]), partial dh per slab -> scratch[n][slab][CS];
// (2) one block per (image, 256 channels): sums the slab partials, applies swish', and forms dpooled with four load chains.
constexpr int SE_SLAB = 128;
constexpr int SE_JW = 12;            // CS_MAX / 4 squeeze channels per wavefront
__device__ __forceinline__ void se_load_slab(float* slab, const float* __restrict__ src, int count, int CS, int tid) {
  const int pitch = CS + 1;
  const float inv_cs = 1.0f / (float)CS;
  for (int e0 = tid; e0 < count; e0 += 256 * 8) {                // eight loads in flight per thread, then the LDS writes
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + 256 * u, count - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      const int r = (int)(((float)e + 0.5f) * inv_cs);          // e / CS, exact for e < 2^16
      if (e < count) slab[r * pitch + (e - r * CS)] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void se_bwd_slab_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ w2, float* __restrict__ dpre2,
                                                          float* __restrict__ dh_part, int C, int CS) {
  extern __shared__ float sm[];     // dp2[SE_SLAB] + slab[SE_SLAB][CS+1]
  float* dp2 = sm;
  float* slab = sm + SE_SLAB;
  const int n = blockIdx.x, sl = blockIdx.y, c0 = sl * SE_SLAB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = min(SE_SLAB, C - c0), pitch = CS + 1;
  if (tid < SE_SLAB) {
    float v = 0.f;
    if (tid < rows) {
      const float g = gate[(int64_t)n * C + c0 + tid];
      v = dgate[(int64_t)n * C + c0 + tid] * g * (1.0f - g);
      dpre2[(int64_t)n * C + c0 + tid] = v;
    }
    dp2[tid] = v;
  }
  se_load_slab(slab, w2 + (int64_t)c0 * CS, rows * CS, CS, tid);
  __syncthreads();
  const float d0 = dp2[lane], d1 = dp2[lane + 64];
#pragma unroll
  for (int q = 0; q < SE_JW; ++q) {
    const int j = wave + 4 * q;
    if (j < CS) {                  // rows past a partial slab hold stale LDS (possibly NaN): select, do not multiply by zero
      const float w0 = lane < rows ? slab[lane * pitch + j] : 0.f, w1v = lane + 64 < rows ? slab[(lane + 64) * pitch + j] : 0.f;
      float a = fmaf(d0, w0, d1 * w1v);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
      if (lane == 0) dh_part[((int64_t)n * gridDim.y + sl) * CS + j] = a;
    }
  }
}

__global__ __launch_bounds__(256) void se_bwd_finish_kernel(const float* __restrict__ dh_part, int slabs, const float* __restrict__ hidden,
                                                            const float* __restrict__ w1, float* __restrict__ dhid,
                                                            float* __restrict__ dpooled, int C, int CS) {
  extern __shared__ float dh[];     // [CS]
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < CS; j += 256) {
    float a = 0.f;
    for (int sl = 0; sl < slabs; ++sl) a += dh_part[((int64_t)n * slabs + sl) * CS + j];
    const float v = a * dswishf_(hidden[(int64_t)n * CS + j]);
    dh[j] = v;
    if (blockIdx.y == 0) dhid[(int64_t)n * CS + j] = v;
  }
  __syncthreads();
  const int c = blockIdx.y * 256 + tid;
  if (c < C) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;    // four chains: the column's CS loads are independent, keep them in flight
    int j = 0;
    for (; j + 3 < CS; j += 4) {
      const float w0 = w1[(int64_t)j * C + c], w1v = w1[(int64_t)(j + 1) * C + c], w2v = w1[(int64_t)(j + 2) * C + c],
                  w3 = w1[(int64_t)(j + 3) * C + c];
      a0 = fmaf(dh[j], w0, a0); a1 = fmaf(dh[j + 1], w1v, a1); a2 = fmaf(dh[j + 2], w2v, a2); a3 = fmaf(dh[j + 3], w3, a3);
    }
    for (; j < CS; ++j) a0 = fmaf(dh[j], w1[(int64_t)j * C + c], a0);
    dpooled[(int64_t)n * C + c] = (a0 + a1) + (a2 + a3);
  }
}

// ------------------------------------------------------------------------------------------------ K5: SE weight grads
// dW2[c,j] += sum_n dpre2[n,c]*swish(hidden[n,j]) ; db2[c] += sum_n dpre2[n,c]
// dW1[j,c] += sum_n dhid[n,j]*pooled[n,c] ;         db1[j] += sum_n dhid[n,j]
// Two [C x CS] products over the batch (K = N images).  Thread = (channel, quarter of the squeeze channels): 2 x 12 accumulators
// instead of 2 x 48; the per-image vectors go through LDS 16 images at a time (broadcast reads) with the 16 images' channel
// values requested up front; dW2 is transposed through LDS so that both weight gradients leave as coalesced atomics
// (a thread per channel issued 96 atomics, each touching 64 cache lines: 80 us at C = 1152, 1.9 ms per step next to the main queue).
constexpr int CS_MAX = 48;
constexpr int SEW_IPC = 16;          // images per LDS chunk
constexpr int SEW_JW = CS_MAX / 4;   // squeeze channels per thread
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ dpre2, const float* __restrict__ dhid,
                                                       const float* __restrict__ hidden, const float* __restrict__ pooled,
                                                       float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                       float* __restrict__ db2, int N, int C, int CS, int imgs_per_block) {
  __shared__ float s1s[SEW_IPC][CS_MAX], dhs[SEW_IPC][CS_MAX];
  __shared__ float tr[64][CS_MAX + 1];
  const int tid = threadIdx.x, cl = tid & 63, jg = tid >> 6;
  const int c = blockIdx.x * 64 + cl, cc = min(c, C - 1);
  const int JG = (CS + 3) / 4, j0 = jg * JG;
  const int n0 = blockIdx.y * imgs_per_block, n1 = min(N, n0 + imgs_per_block);
  const float inv_cs = 1.0f / (float)CS;
  int jq[SEW_JW];
  float a2[SEW_JW], a1[SEW_JW];
#pragma unroll
  for (int q = 0; q < SEW_JW; ++q) { a2[q] = 0.f; a1[q] = 0.f; jq[q] = min(j0 + q, CS - 1); }
  float b2 = 0.f;
  for (int nc = n0; nc < n1; nc += SEW_IPC) {
    __syncthreads();
    for (int i = tid; i < SEW_IPC * CS; i += 256) {
      const int in = (int)(((float)i + 0.5f) * inv_cs), j = i - in * CS;
      const bool ok = nc + in < n1;
      const int n = min(nc + in, n1 - 1);
      s1s[in][j] = ok ? swishf_(hidden[(int64_t)n * CS + j]) : 0.f;      // images past the chunk contribute zeros
      dhs[in][j] = ok ? dhid[(int64_t)n * CS + j] : 0.f;
    }
    float d2[SEW_IPC], pc[SEW_IPC];
#pragma unroll
    for (int u = 0; u < SEW_IPC; ++u) {
      const int n = min(nc + u, n1 - 1);
      d2[u] = dpre2[(int64_t)n * C + cc];
      pc[u] = pooled[(int64_t)n * C + cc];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SEW_IPC; ++u) {
      if (nc + u < n1) b2 += d2[u];
#pragma unroll
      for (int q = 0; q < SEW_JW; ++q) {
        a2[q] = fmaf(d2[u], s1s[u][jq[q]], a2[q]);
        a1[q] = fmaf(dhs[u][jq[q]], pc[u], a1[q]);
      }
    }
  }
  // dW1[j][c]: lanes run over c -> coalesced atomics
#pragma unroll
  for (int q = 0; q < SEW_JW; ++q)
    if (q < JG && j0 + q < CS && c < C) atomicAdd(dw1 + (int64_t)(j0 + q) * C + c, a1[q]);
  if (jg == 0 && c < C) atomicAdd(db2 + c, b2);
  // dW2[c][j]: through LDS, then the block's 64 x CS patch (contiguous in memory) as consecutive atomics
#pragma unroll
  for (int q = 0; q < SEW_JW; ++q)
    if (q < JG && j0 + q < CS) tr[cl][j0 + q] = a2[q];
  __syncthreads();
  for (int i = tid; i < 64 * CS; i += 256) {
    const int r = (int)(((float)i + 0.5f) * inv_cs), j = i - r * CS;
    if (blockIdx.x * 64 + r < C) atomicAdd(dw2 + ((int64_t)blockIdx.x * 64 + r) * CS + j, tr[r][j]);
  }
  if (blockIdx.x == 0 && tid < CS) {
    float b1 = 0.f;
    for (int n = n0; n < n1; ++n) b1 += dhid[(int64_t)n * CS + tid];
    atomicAdd(db1 + tid, b1);
  }
}

// ------------------------------------------------------------------------------------------------ K6: depthwise wgrad, LDS-tiled
// One block = one 16-channel chunk, grid-strided over T x T output tiles of all images.  The activated input tile (with halo)
// and the dz tile are built ONCE per tile in LDS (swish / BN-backward affine evaluated once per element instead of once per
// tap); thread (cq, kh, ps) then accumulates the K taps of kernel row kh for channel quad cq over its share of the pixels.
// Accumulators persist across the block's tiles -> one atomic per (channel, tap) per block.
// RC = Cin > 0: zin is the block input y [N,H,W,Cin]; the depthwise input chunk is rebuilt as y . We^T while staging (rc.hpp).
template <int K, int S, int T, int ACT, int CC, int RC = 0>
__global__ __launch_bounds__(256) void dwconv_wgrad_tiled_kernel(
    const float* __restrict__ du, const float* __restrict__ z, const float* __restrict__ kabc, const float* __restrict__ zin,
    const float* __restrict__ scale_in, const float* __restrict__ shift_in, float* __restrict__ dw, int N, int H, int W, int C,
    int Ho, int Wo, const DetLog det, const float* __restrict__ we) {
  static_assert(RC == 0 || CC == 16, "the recompute yields 16-channel chunks");
  constexpr int P = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int CQN = CC / 4;
  constexpr int IH = (T - 1) * S + K;
  constexpr int IWP = IH | 1;                    // odd row pitch (in pixels) spreads kernel rows over LDS banks
  constexpr int PS = (256 / CQN) / K;            // pixel-split groups
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* a_t = lds;                              // [IH][IWP][CC]
  float* dz_t = lds + IH * IWP * CC;             // [T][T][CC]
  const int tid = threadIdx.x;
  const int cq = tid % CQN, r = tid / CQN;
  const int kh = r % K, ps = r / K;
  const bool worker = ps < PS;
  int chunk_id; int64_t tile0, tile_stride;
  xcd_chunk_tile(C / CC, chunk_id, tile0, tile_stride);
  const int c0 = chunk_id * CC;
  const int ty_n = (Ho + T - 1) / T, tx_n = (Wo + T - 1) / T;
  const int64_t ntiles = (int64_t)N * ty_n * tx_n;

  // per-thread BN vectors for the loader role (channel quad = tid & 3)
  const float4 ka = ld4(kabc + c0 + cq * 4), kb = ld4(kabc + C + c0 + cq * 4), kc = ld4(kabc + 2 * C + c0 + cq * 4);
  // staging role of the input tile: plain = (slot, cq); RC = the MFMA result layout (pixel lane & 15 of the wavefront's group, quad lane >> 4)
  const int lane = tid & 63;
  const int f_q = RC ? (lane >> 4) : cq, f_slot = RC ? ((tid >> 6) * 16 + (lane & 15)) : tid / CQN;
  const float4 sc = ld4(scale_in + c0 + f_q * 4), sh = ld4(shift_in + c0 + f_q * 4);
  constexpr int RCW = RC ? RC : 8;
  RcFrag<RCW> wfrag;
  if constexpr (RC > 0) rc_load<RCW>(wfrag, we + (int64_t)(c0 + (lane & 15)) * RC, lane);

  float4 acc[K];
#pragma unroll
  for (int i = 0; i < K; ++i) acc[i] = f4(0, 0, 0, 0);

  // Software pipeline: the loads of the block's NEXT tile (input tile with halo, du and z tiles) are in flight while the current
  // one is multiplied out of LDS.  Unconditional loads on clamped addresses + validity masks (a predicated load de-pipelines).
  constexpr int NSL = 256 / CQN;
  constexpr int NL = (IH * IH + NSL - 1) / NSL, ND = (T * T + NSL - 1) / NSL;
  static_assert(NL <= 32 && ND <= 32, "validity masks are 32 bits");
  const int slot = tid / CQN;
  float4 pre[RC ? 1 : NL], pdu[ND], pz[ND];
  RcFrag<RCW> ypre[RC ? NL : 1];
  unsigned pre_ok = 0, pd_ok = 0;
  auto fetch = [&](int64_t tile) {
    int n, ty, tx;
    tile_nyx(tile, ty_n, tx_n, n, ty, tx);
    const int oh0 = ty * T, ow0 = tx * T;
    const int ih0 = oh0 * S - P, iw0 = ow0 * S - P;
    const float* img = RC ? zin + (int64_t)n * H * W * RC : zin + (int64_t)n * H * W * C + c0 + cq * 4;
    pre_ok = 0; pd_ok = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pix = f_slot + i * NSL;
      const int iy = pix / IH, ix = pix - iy * IH;
      const int ih = ih0 + iy, iw = iw0 + ix;
      if (pix < IH * IH && ih >= 0 && ih < H && iw >= 0 && iw < W) pre_ok |= 1u << i;
      const int64_t poff = (int64_t)min(max(ih, 0), H - 1) * W + min(max(iw, 0), W - 1);
      if constexpr (RC > 0) rc_load<RCW>(ypre[i], img + poff * RC, lane);
      else pre[i] = ld4(img + poff * C);
    }
    const int64_t obase = (int64_t)n * Ho * Wo * C + c0 + cq * 4;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int pix = slot + i * NSL;
      const int oy = pix / T, ox = pix - oy * T;
      const int oh = oh0 + oy, ow = ow0 + ox;
      if (pix < T * T && oh < Ho && ow < Wo) pd_ok |= 1u << i;
      const int64_t off = obase + ((int64_t)min(oh, Ho - 1) * Wo + min(ow, Wo - 1)) * C;
      pdu[i] = ld4(du + off);
      pz[i] = ld4(z + off);
    }
  };
  int64_t tile = tile0;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tile_stride) {
    __syncthreads();                              // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pix = f_slot + i * NSL;
      float4 zraw;
      if constexpr (RC > 0) zraw = rc_mma<RCW>(wfrag, ypre[i]);       // (all lanes: the MFMA runs outside the range test)
      else zraw = pre[i];
      if (pix < IH * IH) {
        const int iy = pix / IH, ix = pix - iy * IH;
        const float4 v = act4<ACT>(fma4(zraw, sc, sh));
        const bool ok = (pre_ok >> i) & 1u;
        st4(a_t + (iy * IWP + ix) * CC + f_q * 4, f4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f));
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int pix = slot + i * NSL;
      if (pix < T * T) {
        const float4 v = fma4(ka, pdu[i], fma4(kb, pz[i], kc));
        const bool ok = (pd_ok >> i) & 1u;
        st4(dz_t + pix * CC + cq * 4, f4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f));
      }
    }
    __syncthreads();
    if (tile + tile_stride < ntiles) fetch(tile + tile_stride);
    if (worker) {
      for (int p = ps; p < T * T; p += PS) {
        const int oy = p / T, ox = p - oy * T;
        const float4 d = ld4(dz_t + p * CC + cq * 4);
        const float* arow = a_t + ((oy * S + kh) * IWP + ox * S) * CC + cq * 4;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) acc[kw] = fma4(d, ld4(arow + kw * CC), acc[kw]);
      }
    }
  }
  // reduce over the PS pixel-split groups through LDS, then atomics
  __syncthreads();
  float* red = lds;                               // [PS][K(kh)][CQN(cq)][K(kw)] float4
  if (worker) {
#pragma unroll
    for (int kw = 0; kw < K; ++kw) st4(red + (((ps * K + kh) * CQN + cq) * K + kw) * 4, acc[kw]);
  }
  __syncthreads();
  if (tid < K * CQN * K) {                        // (kh, cq, kw)
    const int kw = tid % K, q = (tid / K) % CQN, khh = tid / (CQN * K);
    float4 t = f4(0, 0, 0, 0);
    for (int g = 0; g < PS; ++g) t = add4(t, ld4(red + (((g * K + khh) * CQN + q) * K + kw) * 4));
    const int c = c0 + q * 4, tp = khh * K + kw;
    if (det.vals) {                               // deterministic mode: group = channel chunk, rank = the block's first tile (det.hpp)
      const int rk = (int)tile0, jb = q * 4 * K * K + tp;
      det_put(det, chunk_id, rk, jb, t.x); det_put(det, chunk_id, rk, jb + K * K, t.y);
      det_put(det, chunk_id, rk, jb + 2 * K * K, t.z); det_put(det, chunk_id, rk, jb + 3 * K * K, t.w);
      if (rk == 0 && tid == 0) det_base(det, chunk_id, (int64_t)c0 * K * K);
    } else {
      atomicAdd(dw + (c + 0) * K * K + tp, t.x); atomicAdd(dw + (c + 1) * K * K + tp, t.y);
      atomicAdd(dw + (c + 2) * K * K + tp, t.z); atomicAdd(dw + (c + 3) * K * K + tp, t.w);
    }
  }
}

template <int K, int S, int T, int ACT, int CC, int RC = 0>
int launch_dw_wgrad_tiled(const float* du, const float* z, const float* kabc, const float* zin, const float* scale_in,
                          const float* shift_in, float* dw, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s,
                          const float* we = nullptr) {
  constexpr int IH = (T - 1) * S + K;
  constexpr int IWP = IH | 1;
  constexpr int CQN = CC / 4;
  constexpr int PS = (256 / CQN) / K;
  size_t lds = (size_t)(IH * IWP * CC + T * T * CC) * sizeof(float);
  const size_t red = (size_t)PS * K * CQN * K * 4 * sizeof(float);
  if (red > lds) lds = red;
  const int chunks = C / CC;
  const int64_t ntiles = (int64_t)N * ((Ho + T - 1) / T) * ((Wo + T - 1) / T);
  const unsigned bx = xcd_chunk_grid(chunks, ntiles, 4096);
  auto k = dwconv_wgrad_tiled_kernel<K, S, T, ACT, CC, RC>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_dwconv_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  DetScope det(s, chunks, (int)((bx >> 3) / chunks) * 8, CC * K * K);     // ranks = first tiles of a chunk's blocks (xcd_chunk_tile)
  hipLaunchKernelGGL(k, dim3(bx), dim3(256), lds, s, du, z, kabc, zin, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, det.log, we);
  const int rc = check_launch("mt_dwconv_bwd(weight, tiled)");
  return rc ? rc : det.reduce_f32(dw);
}

// ------------------------------------------------------------------------------------------------ K7: depthwise dgrad, LDS-tiled
// One block = one 16-channel chunk, grid-strided over T x T INPUT tiles.  dz = ka*du+kb*z+kc over the output positions
// the tile's taps can reach is built once in LDS; thread (cq, slot) then gathers its input pixels' taps from LDS,
// applies swish' of the input-side BatchNorm and accumulates that BatchNorm's backward sums in registers.
// WG = true also produces the depthwise WEIGHT gradient from the same tile (input-centric form of the same sum:
//   dW[kh,kw] = sum over input pixels of a_in[iy,ix] * dz[(iy+P-kh)/S, (ix+P-kw)/S]  -- exactly the taps the data gradient gathers),
// so du, z and the dw input are streamed once instead of once per kernel (the separate weight-gradient kernel re-reads all three:
// 10.5 of the EfficientNet step's 79 GB, and evaluates swish and the BatchNorm-backward affine a second time).  Two phases per tile
// (round 6): the gather leaves a_in = act(u) of its pixels in an LDS tile; after a barrier the threads change role to (channel quad,
// kernel row kh, pixel share) like the stand-alone weight-gradient kernel and walk a_in x dz out of LDS with K float4 accumulators each
// (rounds 2-5 kept K*K accumulators per gather thread: 216-254 VGPRs, two wavefronts per SIMD, slower than the two kernels).
// Accumulators persist across the block's tiles; reduced through LDS at the end, one atomic per (channel, tap) and block.
thread_local int g_res_stride = 1;      // set by mt_dwconv_bwd_res2 around its call (one more kernel argument, no new instantiations)

// RC = Cin > 0: zin is the block input y [N,H,W,Cin]; the tile's raw depthwise input chunk z = y . We^T is rebuilt into an LDS tile
// (rc.hpp) next to the dz tile and the gather reads it from there (the plain form loads it from global memory inside the gather).
template <int K, int S, int T, int ACT, int CC, bool WG = false, int RC = 0>
__global__ __launch_bounds__(256) void dwconv_dgrad_tiled_kernel(
    const float* __restrict__ du, const float* __restrict__ z, const float* __restrict__ kabc, const float* __restrict__ w,
    const float* __restrict__ zin, const float* __restrict__ scale_in, const float* __restrict__ shift_in,
    const float* __restrict__ mi_in, float* __restrict__ du_in, double* __restrict__ stats, int slots, int N, int H, int W,
    int C, int Ho, int Wo, const float* __restrict__ res_pre, const float* __restrict__ res_post, float* __restrict__ dw,
    int res_stride, const float* __restrict__ we) {
  static_assert(RC == 0 || CC == 16, "the recompute yields 16-channel chunks");
  constexpr int P = S == 1 ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int CQN = CC / 4, NSLOT = 256 / CQN;
  constexpr int OT = (T - 1 + K - 1) / S + 2;       // output rows/cols a T-wide input tile can touch (upper bound)
  constexpr int OTP = OT | 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // dz_t [OT][OTP][CC]; later the stats reduction buffer
  const int tid = threadIdx.x;
  const int cq = tid % CQN, slot = tid / CQN;
  int chunk_id; int64_t tile0, tile_stride;
  xcd_chunk_tile(C / CC, chunk_id, tile0, tile_stride);
  const int c0 = chunk_id * CC;
  const int ty_n = (H + T - 1) / T, tx_n = (W + T - 1) / T;
  const int64_t ntiles = (int64_t)N * ty_n * tx_n;
  const int c = c0 + cq * 4;
  const float4 sc = ld4(scale_in + c), sh = ld4(shift_in + c);
  const float4 mean = mi_in ? ld4(mi_in + c) : f4(0, 0, 0, 0), istd = mi_in ? ld4(mi_in + C + c) : f4(0, 0, 0, 0);
  // weights in LDS behind the dz tile, read as broadcasts, for 5x5 (25 float4 = 100 VGPRs) and for stride 1 (the rolled kernel-row
  // loop below; 3x3 stride 1 with its 36 weight registers sat at 188 VGPRs = two wavefronts per SIMD); 3x3 stride 2 keeps registers
  constexpr bool WLDS = K > 3 || S == 1;
  float* w_t = lds + OT * OTP * CC;
  float* z_t = w_t + (WLDS ? K * K * CC : 0);        // RC: raw input chunk of the tile, [T*T][CC]
  constexpr int RCW = RC ? RC : 8;
  constexpr int NLZ = (T * T + 63) / 64;
  const int lane = tid & 63;
  const int f_q = lane >> 4, f_slot = (tid >> 6) * 16 + (lane & 15);      // RC staging role (MFMA result layout)
  RcFrag<RCW> wfrag, ypre[RC ? NLZ : 1];
  if constexpr (RC > 0) rc_load<RCW>(wfrag, we + (int64_t)(c0 + (lane & 15)) * RC, lane);
  float4 wt[WLDS ? 1 : K * K];
  if constexpr (WLDS) {
    if (slot == 0)
      for (int i = 0; i < K * K; ++i)
        st4(w_t + i * CC + cq * 4, f4(w[(c + 0) * K * K + i], w[(c + 1) * K * K + i], w[(c + 2) * K * K + i], w[(c + 3) * K * K + i]));
  } else {
#pragma unroll
    for (int i = 0; i < K * K; ++i)
      wt[i] = f4(w[(c + 0) * K * K + i], w[(c + 1) * K * K + i], w[(c + 2) * K * K + i], w[(c + 3) * K * K + i]);
  }
  const float4 ka = ld4(kabc + c), kb = ld4(kabc + C + c), kc = ld4(kabc + 2 * C + c);
  float4 s1 = f4(0, 0, 0, 0), s2 = s1;
  // WG: phase-2 role (cq2, kh2, ps2) and its K tap accumulators; a_t = activated input of the tile's pixels, [T*T][CC]
  float* a_t = z_t + (RC ? T * T * CC : 0);
  constexpr int PS2 = NSLOT / K;
  const int r2 = tid / CQN, kh2 = r2 % K, ps2 = r2 / K;
  float4 wacc[WG ? K : 1];
#pragma unroll
  for (int i = 0; i < (WG ? K : 1); ++i) wacc[i] = f4(0, 0, 0, 0);
  // Software pipeline: du / z of the outputs reachable from the block's NEXT tile are in flight while the current tile is
  // gathered out of LDS (prefetching the tile's own raw inputs as well cost more in registers than it hid).  Unconditional loads on clamped addresses + a validity mask.
  constexpr int ND = (OT * OT + NSLOT - 1) / NSLOT;
  static_assert(ND <= 32, "validity mask is 32 bits");
  float4 pdu[ND], pz[ND];
  unsigned pd_ok = 0;
  auto lows = [&](int64_t tile, int& n, int& ih0, int& iw0, int& oh_lo, int& ow_lo) {
    int ty, tx;
    tile_nyx(tile, ty_n, tx_n, n, ty, tx);
    ih0 = ty * T; iw0 = tx * T;
    // first output row/col reachable from this tile: ceil((ih0 + P - (K-1)) / S), possibly negative
    const int nh = ih0 + P - (K - 1), nw = iw0 + P - (K - 1);
    oh_lo = nh >= 0 ? (nh + S - 1) / S : -((-nh) / S);
    ow_lo = nw >= 0 ? (nw + S - 1) / S : -((-nw) / S);
  };
  auto fetch = [&](int64_t tile) {
    int n, ih0, iw0, oh_lo, ow_lo;
    lows(tile, n, ih0, iw0, oh_lo, ow_lo);
    const int64_t obase = (int64_t)n * Ho * Wo * C + c;
    pd_ok = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int pix = slot + i * NSLOT;
      const int oy = pix / OT, ox = pix - oy * OT;
      const int oh = oh_lo + oy, ow = ow_lo + ox;
      if (pix < OT * OT && oh >= 0 && oh < Ho && ow >= 0 && ow < Wo) pd_ok |= 1u << i;
      const int64_t off = obase + ((int64_t)min(max(oh, 0), Ho - 1) * Wo + min(max(ow, 0), Wo - 1)) * C;
      pdu[i] = ld4(du + off);
      pz[i] = ld4(z + off);
    }
    if constexpr (RC > 0) {
      const float* img = zin + (int64_t)n * H * W * RC;
#pragma unroll
      for (int i = 0; i < NLZ; ++i) {
        const int pix = f_slot + i * 64;
        const int iy = pix / T, ix = pix - iy * T;
        rc_load<RCW>(ypre[i], img + ((int64_t)min(ih0 + iy, H - 1) * W + min(iw0 + ix, W - 1)) * RC, lane);
      }
    }
  };
  int64_t tile = tile0;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tile_stride) {
    int n, ih0, iw0, oh_lo, ow_lo;
    lows(tile, n, ih0, iw0, oh_lo, ow_lo);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int pix = slot + i * NSLOT;
      if (pix < OT * OT) {
        const int oy = pix / OT, ox = pix - oy * OT;
        const float4 v = fma4(ka, pdu[i], fma4(kb, pz[i], kc));
        const bool ok = (pd_ok >> i) & 1u;
        st4(lds + (oy * OTP + ox) * CC + cq * 4, f4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f));
      }
    }
    if constexpr (RC > 0) {
#pragma unroll
      for (int i = 0; i < NLZ; ++i) {
        const int pix = f_slot + i * 64;
        const float4 zq = rc_mma<RCW>(wfrag, ypre[i]);                 // (all lanes)
        if (pix < T * T) st4(z_t + pix * CC + f_q * 4, zq);
      }
    }
    __syncthreads();
    if (tile + tile_stride < ntiles) fetch(tile + tile_stride);
#pragma unroll 1
    for (int p = slot; p < T * T; p += NSLOT) {
      const int iy = p / T, ix = p - iy * T;
      const int ih = ih0 + iy, iw = iw0 + ix;
      if constexpr (WG) {
        if (!(ih < H && iw < W)) st4(a_t + p * CC + cq * 4, f4(0, 0, 0, 0));      // pixels past the image edge contribute nothing
      }
      if (ih < H && iw < W) {
        float4 acc = f4(0, 0, 0, 0);
        const int64_t off = (((int64_t)n * H + ih) * W + iw) * C + c;
        float4 zz;
        if constexpr (RC > 0) zz = ld4(z_t + p * CC + cq * 4);
        else zz = ld4(zin + off);
        const float4 u = fma4(zz, sc, sh);
        if constexpr (WG) st4(a_t + p * CC + cq * 4, act4<ACT>(u));       // the depthwise conv's input at this pixel, for phase 2
        // The taps are read WITHOUT per-tap range tests: the dz tile holds every output position the tile's taps can reach, zero
        // where that lies outside the image (rows / columns from oh_lo / ow_lo on, negative ones included), so the K*K LDS reads of
        // a pixel are independent and go out back to back.  (Rounds 2-5 skipped "negative" taps one by one: every ds_read sat in its
        // own exec-masked branch behind an s_waitcnt lgkmcnt(0) -- K*K serial LDS round trips per pixel.)
        int wq = cq * 4;
        if constexpr (WLDS) asm volatile("" : "+v"(wq));      // (opaque per pixel: hoisted out of the pixel loop the 25 weight quads are 100 VGPRs)
        auto tap = [&](int kh, int kw, int oy, int ox) {
          float4 ww;
          if constexpr (WLDS) ww = ld4(w_t + (kh * K + kw) * CC + wq);
          else ww = wt[kh * K + kw];
          acc = fma4(ld4(lds + (oy * OTP + ox) * CC + cq * 4), ww, acc);
        };
        if constexpr (S == 1) {
          const int oy0 = ih + P - oh_lo, ox0 = iw + P - ow_lo;          // tap (kh, kw) meets dz[oy0 - kh][ox0 - kw]
          if constexpr (WLDS) {
            // 5x5: weights come from LDS; a ROLLED loop over kernel rows keeps one row (5 + 5 reads) in flight -- unrolled, the compiler
            // hoists all 50 reads of a pixel and the kernel needs 256 VGPRs (one wavefront per SIMD)
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
#pragma unroll
              for (int kw = 0; kw < K; ++kw) tap(kh, kw, oy0 - kh, ox0 - kw);
            }
          } else {
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
#pragma unroll
              for (int kw = 0; kw < K; ++kw) tap(kh, kw, oy0 - kh, ox0 - kw);
              __builtin_amdgcn_sched_barrier(0);      // one kernel row of reads in flight
            }
          }
        } else {
          // stride 2: only the taps with kh = (ih + P) mod 2, kw = (iw + P) mod 2 (mod 2) meet an output; (ih + P - kh) is even there,
          // so the arithmetic shift divides exactly, also below zero
          const int ph = (ih + P) & 1, pw = (iw + P) & 1;
          const int oyb = ((ih + P - ph) >> 1) - oh_lo, oxb = ((iw + P - pw) >> 1) - ow_lo;
          auto taps = [&](auto PH, auto PW) {
            constexpr int ph_ = decltype(PH)::value, pw_ = decltype(PW)::value;
#pragma unroll
            for (int kh = ph_; kh < K; kh += 2) {
#pragma unroll
              for (int kw = pw_; kw < K; kw += 2) tap(kh, kw, oyb - (kh - ph_) / 2, oxb - (kw - pw_) / 2);
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          using I0 = std::integral_constant<int, 0>;
          using I1 = std::integral_constant<int, 1>;
          if (ph == 0) { if (pw == 0) taps(I0{}, I0{}); else taps(I0{}, I1{}); }
          else { if (pw == 0) taps(I1{}, I0{}); else taps(I1{}, I1{}); }
        }
        // res_stride 2: the residual gradient comes from a stride-2 1x1 convolution (Xception's skip path): it exists at even
        // (ih, iw) only, as [N, ceil(H/2), ceil(W/2), C] -- no zero-filled full-resolution copy of it is made
        const bool r_on = res_stride == 1 || !((ih | iw) & 1);
        const int64_t roff = res_stride == 1 ? off : (((int64_t)n * ((H + 1) >> 1) + (ih >> 1)) * ((W + 1) >> 1) + (iw >> 1)) * C + c;
        if (res_pre && r_on) acc = add4(acc, ld4(res_pre + roff));       // another consumer of the same activated tensor
        float4 d = mul4(acc, dact4<ACT>(u));
        if (res_post && r_on) d = add4(d, ld4(res_post + roff));         // a consumer of the raw (pre-activation) tensor
        st4(du_in + off, d);
        const float4 xh = f4((zz.x - mean.x) * istd.x, (zz.y - mean.y) * istd.y, (zz.z - mean.z) * istd.z, (zz.w - mean.w) * istd.w);
        s1 = add4(s1, d);
        s2 = fma4(d, xh, s2);
      }
    }
    if constexpr (WG) {
      // phase 2: weight gradient of this tile out of LDS (a_t x dz tile), K taps of kernel row kh2 per thread
      __syncthreads();
      if (ps2 < PS2) {
        const int ohb = ih0 + P - kh2;                  // output row index (times S) reached from input row ih0 + iy through tap row kh2
#pragma unroll 1
        for (int p = ps2; p < T * T; p += PS2) {
          const int iy = p / T, ix = p - iy * T;
          const int ohn = ohb + iy;
          if (S == 2 && (ohn & 1)) continue;
          const float4 a = ld4(a_t + p * CC + cq * 4);
          const float* drow = lds + (((ohn >> (S - 1)) - oh_lo) * OTP) * CC + cq * 4;     // (even below zero: the shift divides exactly)
          const int owb = iw0 + ix + P;
          if constexpr (S == 1) {                       // unconditional reads (zero halo in the dz tile), all K in flight
#pragma unroll
            for (int kw = 0; kw < K; ++kw) wacc[kw] = fma4(ld4(drow + (owb - kw - ow_lo) * CC), a, wacc[kw]);
          } else {
            const int pw = owb & 1, oxb = ((owb - pw) >> 1) - ow_lo;
            if (pw == 0) {
#pragma unroll
              for (int kw = 0; kw < K; kw += 2) wacc[kw] = fma4(ld4(drow + (oxb - kw / 2) * CC), a, wacc[kw]);
            } else {
#pragma unroll
              for (int kw = 1; kw < K; kw += 2) wacc[kw] = fma4(ld4(drow + (oxb - (kw - 1) / 2) * CC), a, wacc[kw]);
            }
          }
        }
      }
    }
  }
  if constexpr (WG) {
    // sum the accumulators over the PS2 pixel shares through LDS, then one atomic per (channel, tap) per block
    __syncthreads();
    float* red = lds;                               // [PS2][K(kh)][CQN(cq)][K(kw)] float4
    if (ps2 < PS2) {
#pragma unroll
      for (int kw = 0; kw < K; ++kw) st4(red + (((ps2 * K + kh2) * CQN + cq) * K + kw) * 4, wacc[kw]);
    }
    __syncthreads();
    if (tid < K * CQN * K) {                        // (kh, cq, kw)
      const int kw = tid % K, q = (tid / K) % CQN, khh = tid / (CQN * K);
      float4 t = f4(0, 0, 0, 0);
      for (int g = 0; g < PS2; ++g) t = add4(t, ld4(red + (((g * K + khh) * CQN + q) * K + kw) * 4));
      const int ch = c0 + q * 4, tp = khh * K + kw;
      atomicAdd(dw + (ch + 0) * K * K + tp, t.x); atomicAdd(dw + (ch + 1) * K * K + tp, t.y);
      atomicAdd(dw + (ch + 2) * K * K + tp, t.z); atomicAdd(dw + (ch + 3) * K * K + tp, t.w);
    }
  }
  if (!stats) return;
  __syncthreads();
  float* rr = lds + tid * 8;
  rr[0] = s1.x; rr[1] = s1.y; rr[2] = s1.z; rr[3] = s1.w; rr[4] = s2.x; rr[5] = s2.y; rr[6] = s2.z; rr[7] = s2.w;
  __syncthreads();
  if (tid < CQN * 8) {
    const int q = tid >> 3, e = tid & 7;
    float v = 0.f;
    for (int sl = 0; sl < NSLOT; ++sl) v += lds[(sl * CQN + q) * 8 + e];
    const int ch = c0 + q * 4 + (e & 3), which = e >> 2;
    stat_add(stats + ((int64_t)(blockIdx.x % stat_slots(slots)) * 2 + which) * C + ch, stat_limb(slots, C), v);
  }
}

template <int K, int S, int T, int ACT, int CC, bool WG = false, int RC = 0>
int launch_dw_dgrad_tiled(const float* du, const float* z, const float* kabc, const float* w, const float* zin,
                          const float* scale_in, const float* shift_in, const float* mi_in, float* du_in, double* stats, int slots,
                          int N, int H, int W, int C, int Ho, int Wo, const float* res_pre, const float* res_post, hipStream_t s,
                          float* dw = nullptr, const float* we = nullptr) {
  constexpr int OT = (T - 1 + K - 1) / S + 2;
  constexpr int OTP = OT | 1;
  constexpr int CQN = CC / 4, NSLOT = 256 / CQN;
  size_t lds = (size_t)(OT * OTP * CC + ((K > 3 || S == 1) ? K * K * CC : 0) + (RC ? T * T * CC : 0) + (WG ? T * T * CC : 0)) * sizeof(float);
  if (lds < 256 * 8 * sizeof(float)) lds = 256 * 8 * sizeof(float);
  if (WG && lds < (size_t)(NSLOT / K) * K * CQN * K * 4 * sizeof(float)) lds = (size_t)(NSLOT / K) * K * CQN * K * 4 * sizeof(float);
  const int chunks = C / CC;
  const int64_t ntiles = (int64_t)N * ((H + T - 1) / T) * ((W + T - 1) / T);
  // the fused form keeps K*K tap accumulators per thread across tiles: fewer, longer-lived blocks keep the final atomics rare
  const unsigned bx = xcd_chunk_grid(chunks, ntiles, WG ? 4096 : 8192);
  auto kfn = dwconv_dgrad_tiled_kernel<K, S, T, ACT, CC, WG, RC>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kfn), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_dwconv_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kfn, dim3(bx), dim3(256), lds, s, du, z, kabc, w, zin,
                     scale_in, shift_in, mi_in, du_in, stats, slots != 0 ? slots : 1, N, H, W, C, Ho, Wo, res_pre, res_post, dw,
                     g_res_stride, we);
  return check_launch(WG ? "mt_dwconv_bwd(data + weight, fused)" : "mt_dwconv_bwd(data, tiled)");
}

template <int K, int S, int ACT>
int launch_dw_bwd_tiled_any(const float* du, const float* z, const float* kabc, const float* w, const float* zin, const float* scale_in,
                            const float* shift_in, const float* mi_in, float* du_in, double* stats, int slots, float* dw, int N, int H,
                            int W, int C, int Ho, int Wo, int parts, const float* res_pre, const float* res_post, hipStream_t s) {
  int rc = 0;
  static const int t7_mask = getenv("MT_DW_T7") ? atoi(getenv("MT_DW_T7")) : 0;       // lab: bit 1 = 7 x 7 tiles in the data gradient, bit 2 = in the weight gradient
  const bool t14o = Ho >= 14 && !(t7_mask & 4), t14i = H >= 14 && !(t7_mask & 2);
  if (parts == 3) {       // one pass over du / z / the dw input for both gradients
    if (C % 16 == 0) return t14i ? launch_dw_dgrad_tiled<K, S, 14, ACT, 16, true>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s, dw)
                                 : launch_dw_dgrad_tiled<K, S, 7, ACT, 16, true>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s, dw);
    return t14i ? launch_dw_dgrad_tiled<K, S, 14, ACT, 8, true>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s, dw)
                : launch_dw_dgrad_tiled<K, S, 7, ACT, 8, true>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s, dw);
  }
  if (parts & 1) {
    if (C % 16 == 0) rc = t14o ? launch_dw_wgrad_tiled<K, S, 14, ACT, 16>(du, z, kabc, zin, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, s)
                               : launch_dw_wgrad_tiled<K, S, 7, ACT, 16>(du, z, kabc, zin, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, s);
    else rc = t14o ? launch_dw_wgrad_tiled<K, S, 14, ACT, 8>(du, z, kabc, zin, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, s)
                   : launch_dw_wgrad_tiled<K, S, 7, ACT, 8>(du, z, kabc, zin, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, s);
    if (rc) return rc;
  }
  if (parts & 2) {
    if (C % 16 == 0) rc = t14i ? launch_dw_dgrad_tiled<K, S, 14, ACT, 16>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s)
                               : launch_dw_dgrad_tiled<K, S, 7, ACT, 16>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s);
    else rc = t14i ? launch_dw_dgrad_tiled<K, S, 14, ACT, 8>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s)
                   : launch_dw_dgrad_tiled<K, S, 7, ACT, 8>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, N, H, W, C, Ho, Wo, res_pre, res_post, s);
  }
  return rc;
}

// ------------------------------------------------------------------------------------------------ K8: stem wgrad
// dW[co,ci,kh,kw] += sum_pix dz0[pix,co] * x[n, 2oh+kh-P, 2ow+kw-P, ci],  dz0 = ka*du+kb*z+kc
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ du, const float* __restrict__ z,
                                                         const float* __restrict__ kabc, const void* __restrict__ xv, int x_u8,
                                                         float* __restrict__ dw, int N, int H, int W, int Ho, int Wo, int pad0,
                                                         const DetLog det) {
  const float* x = reinterpret_cast<const float*>(xv);
  const uint8_t* xb = reinterpret_cast<const uint8_t*>(xv);
  constexpr int CO = 32, TP = 64, TAPS = 27;
  __shared__ float dzt[TP][CO + 1];
  __shared__ float xt[TP][TAPS + 1];
  const int tid = threadIdx.x;
  const int co = tid & 31, g = tid >> 5;
  const float ka = kabc[co], kb = kabc[CO + co], kc = kabc[2 * CO + co];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t total = (int64_t)N * Ho * Wo;
  const int64_t ntiles = (total + TP - 1) / TP;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p0 = tile * TP;
    __syncthreads();
    for (int i = tid; i < TP * CO; i += 256) {
      const int p = i >> 5, c2 = i & 31;
      const int64_t pix = p0 + p;
      float v = 0.f;
      if (pix < total) v = fmaf(kabc[c2], du[pix * CO + c2], fmaf(kabc[CO + c2], z[pix * CO + c2], kabc[2 * CO + c2]));
      dzt[p][c2] = v;
    }
    for (int i = tid; i < TP * TAPS; i += 256) {
      const int p = i / TAPS, tp = i - p * TAPS;
      const int ci = tp % 3, kk = tp / 3, kw = kk % 3, kh = kk / 3;
      const int64_t pix = p0 + p;
      float v = 0.f;
      if (pix < total) {
        const int ow = (int)(pix % Wo);
        const int64_t t = pix / Wo;
        const int oh = (int)(t % Ho), n = (int)(t / Ho);
        const int ih = oh * 2 + kh - pad0, iw = ow * 2 + kw - pad0;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
          const int64_t off = (((int64_t)n * H + ih) * W + iw) * 3 + ci;
          v = x_u8 ? (float)xb[off] : x[off];
        }
      }
      xt[p][tp] = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int p = 0; p < TP; ++p) {
      const float d = dzt[p][co];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tp = g + 8 * i;
        if (tp < TAPS) acc[i] = fmaf(d, xt[p][tp], acc[i]);
      }
    }
  }
  (void)ka; (void)kb; (void)kc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tp = g + 8 * i;
    if (tp < TAPS) {
      const int ci = tp % 3, kk = tp / 3, kw = kk % 3, kh = kk / 3;
      if (det.vals) det_put(det, 0, blockIdx.x, ((co * 3 + ci) * 3 + kh) * 3 + kw, acc[i]);     // deterministic mode: block order
      else atomicAdd(dw + ((co * 3 + ci) * 3 + kh) * 3 + kw, acc[i]);
    }
  }
}

int pick_cqb(int CQ) {
  int best = 1;
  for (int d = 1; d <= 64 && d <= CQ; ++d)
    if (CQ % d == 0) best = d;
  return best;
}

template <int K, int S>
int launch_dw_bwd(const float* du, const float* z, const float* kabc, const float* w, const float* zin, const float* scale_in,
                  const float* shift_in, const float* mi_in, float* du_in, double* stats, int slots, float* dw, int N, int H,
                  int W, int C, int parts, int act, const float* res_pre, const float* res_post, hipStream_t s) {
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S;
  if (C % 8 != 0) return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd: channel count %d is not a multiple of 8", C);
  if (act == 1) return launch_dw_bwd_tiled_any<K, S, 1>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, dw, N, H, W, C, Ho, Wo, parts, res_pre, res_post, s);
  if (act == 2) return launch_dw_bwd_tiled_any<K, S, 2>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, dw, N, H, W, C, Ho, Wo, parts, res_pre, res_post, s);
  return launch_dw_bwd_tiled_any<K, S, 0>(du, z, kabc, w, zin, scale_in, shift_in, mi_in, du_in, stats, slots, dw, N, H, W, C, Ho, Wo, parts, res_pre, res_post, s);
}

}  // namespace

extern "C" int mt_bn_act_bwd(const float* din, const float* z, const float* scale, const float* shift,
                             const float* mean_invstd, const float* gate, const float* dpool, const float* rowscale,
                             float* dout, double* stats, int slots, int64_t rows, int C, int hw, int act, void* stream) {
  if (!din || !z || !scale || !shift || !mean_invstd || !stats) return fail(MT_ERR_ARG, "mt_bn_act_bwd: null pointer");
  if ((gate == nullptr) != (dpool == nullptr)) return fail(MT_ERR_ARG, "mt_bn_act_bwd: gate and dpool go together");
  if (C & 3) return fail(MT_ERR_ARG, "mt_bn_act_bwd: C %% 4 != 0");
  const int CQ = C / 4, CQB = pick_cqb(CQ), PB = 256 / CQB;
  int64_t nb = (rows + PB - 1) / PB;
  const int64_t cap = 4096 / (CQ / CQB) > 0 ? 4096 / (CQ / CQB) : 1;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((unsigned)nb, CQ / CQB), dim3(CQB * PB), (size_t)PB * CQB * 8 * sizeof(float),
                     (hipStream_t)stream, din, z, scale, shift, mean_invstd, gate, dpool, rowscale, dout, stats,
                     slots != 0 ? slots : 1, rows, C, hw > 0 ? hw : 1, act, CQB, PB);
  return check_launch("mt_bn_act_bwd");
}

extern "C" int mt_bn_bwd_finalize(const double* stats, int slots, double count, const float* gamma, const float* mean_invstd,
                                  float* kabc, float* dgamma, float* dbeta, int C, int training, void* stream) {
  if (!stats || !gamma || !mean_invstd || !kabc) return fail(MT_ERR_ARG, "mt_bn_bwd_finalize: null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, stats, slots, count, gamma,
                     mean_invstd, kabc, dgamma, dbeta, C, training);
  return check_launch("mt_bn_bwd_finalize");
}

static int se_adjoint(const float* dgate, const float* gate, const float* hidden, const float* w1, const float* w2, float* dpre2,
                      float* dhid, float* dpooled, float* scratch, int N, int C, int CS, hipStream_t s) {
  const int slabs = (C + SE_SLAB - 1) / SE_SLAB;
  hipLaunchKernelGGL(se_bwd_slab_kernel, dim3(N, slabs), dim3(256), (size_t)(SE_SLAB + SE_SLAB * (CS + 1)) * sizeof(float), s, dgate, gate,
                     w2, dpre2, scratch, C, CS);
  int rc = check_launch("mt_se_bwd(slab)");
  if (rc) return rc;
  hipLaunchKernelGGL(se_bwd_finish_kernel, dim3(N, (C + 255) / 256), dim3(256), (size_t)CS * sizeof(float), s, scratch, slabs, hidden, w1,
                     dhid, dpooled, C, CS);
  return check_launch("mt_se_bwd(finish)");
}

extern "C" int mt_se_scratch_floats(int N, int C, int CS) { return N * ((C + SE_SLAB - 1) / SE_SLAB) * CS; }

extern "C" int mt_se_bwd(const float* da, const float* z, const float* scale, const float* shift, const float* gate,
                         const float* hidden, const float* pooled, const float* w1, const float* w2, float* dgate,
                         float* dpre2, float* dhid, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int N,
                         int HW, int C, int CS, int parts_mask, float* scratch, void* stream) {
  // parts_mask: 1 = d-gate reduction over da, z + per-image adjoint; 2 = weight gradients; 4 = per-image adjoint only (dgate was
  // already reduced by mt_gemm's MT_EPI_SE_RED epilogue; da / z / scale / shift are not read)
  if (((parts_mask & 1) && (!da || !z || !scale || !shift)) || !gate || !hidden || !pooled || !w1 || !w2 || !dgate || !dpre2 || !dhid ||
      !dpooled || !dw1 || !db1 || !dw2 || !db2 || ((parts_mask & 5) && !scratch))
    return fail(MT_ERR_ARG, "mt_se_bwd: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_se_bwd: C %% 4 != 0");
  if (CS > CS_MAX) return fail(MT_ERR_UNSUPPORTED, "mt_se_bwd: squeeze width %d > %d", CS, CS_MAX);
  hipStream_t s = (hipStream_t)stream;
  const int CQ = C / 4, CQB = pick_cqb(CQ), PB = 256 / CQB;
  int rc = 0;
  if (parts_mask & 1) {
  int parts = 1;
  while ((int64_t)N * (CQ / CQB) * parts < 2048 && HW / (parts * 2) >= PB * 16) parts *= 2;
  if (det_enabled()) parts = 1;                  // deterministic mode: one block per (image, channel chunk), plain stores
  if (parts > 1 && hipMemsetAsync(dgate, 0, (size_t)N * C * sizeof(float), s) != hipSuccess)
    return fail(MT_ERR_LAUNCH, "mt_se_bwd: memset failed");
  hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(N, CQ / CQB, parts), dim3(CQB * PB), (size_t)PB * CQB * 4 * sizeof(float), s, da, z,
                     scale, shift, dgate, HW, C, CQB, PB);
  rc = check_launch("mt_se_bwd(reduce)");
  if (rc) return rc;
  rc = se_adjoint(dgate, gate, hidden, w1, w2, dpre2, dhid, dpooled, scratch, N, C, CS, s);
  if (rc) return rc;
  }
  if (parts_mask & 4) {
    rc = se_adjoint(dgate, gate, hidden, w1, w2, dpre2, dhid, dpooled, scratch, N, C, CS, s);
    if (rc) return rc;
  }
  if (!(parts_mask & 2)) return 0;
  // images per block: enough blocks to cover the chip a few times over, few enough that the atomic partial sums stay cheap
  int ipb = 64;
  while (ipb > 16 && (int64_t)((C + 63) / 64) * ((N + ipb - 1) / ipb) < 128) ipb >>= 1;
  if (det_enabled()) ipb = N;                    // deterministic mode: every output element has one contributing block
  hipLaunchKernelGGL(se_wgrad_kernel, dim3((C + 63) / 64, (N + ipb - 1) / ipb), dim3(256), 0, s, dpre2, dhid, hidden, pooled, dw1,
                     db1, dw2, db2, N, C, CS, ipb);
  return check_launch("mt_se_bwd(wgrad)");
}

extern "C" int mt_dwconv_bwd(const float* du, const float* z, const float* kabc, const float* w, const float* zin,
                             const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in,
                             double* stats_in, int slots, float* dw, int N, int H, int W, int C, int k, int stride,
                             int parts, int act, const float* res_pre, const float* res_post, void* stream) {
  if (!du || !z || !kabc || !zin || !scale_in || !shift_in) return fail(MT_ERR_ARG, "mt_dwconv_bwd: null pointer");
  if ((parts & 1) && !dw) return fail(MT_ERR_ARG, "mt_dwconv_bwd: weight part needs dw");
  if ((parts & 2) && (!w || !du_in)) return fail(MT_ERR_ARG, "mt_dwconv_bwd: data part needs w and du_in");
  if ((parts & 2) && ((stats_in == nullptr) != (mean_invstd_in == nullptr))) return fail(MT_ERR_ARG, "mt_dwconv_bwd: stats_in and mean_invstd_in go together");
  if (stride == 2 && ((H & 1) || (W & 1))) return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd: stride 2 needs even H, W");
  hipStream_t s = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_dw_bwd<3, 1>(du, z, kabc, w, zin, scale_in, shift_in, mean_invstd_in, du_in, stats_in, slots, dw, N, H, W, C, parts, act, res_pre, res_post, s);
  if (k == 3 && stride == 2) return launch_dw_bwd<3, 2>(du, z, kabc, w, zin, scale_in, shift_in, mean_invstd_in, du_in, stats_in, slots, dw, N, H, W, C, parts, act, res_pre, res_post, s);
  if (k == 5 && stride == 1) return launch_dw_bwd<5, 1>(du, z, kabc, w, zin, scale_in, shift_in, mean_invstd_in, du_in, stats_in, slots, dw, N, H, W, C, parts, act, res_pre, res_post, s);
  if (k == 5 && stride == 2) return launch_dw_bwd<5, 2>(du, z, kabc, w, zin, scale_in, shift_in, mean_invstd_in, du_in, stats_in, slots, dw, N, H, W, C, parts, act, res_pre, res_post, s);
  return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd: k=%d stride=%d unsupported", k, stride);
}

// The adjoint with the depthwise input rebuilt from the block input (rc.hpp): instances as in effnet_fwd.hip (MT_DW_RC_INSTANCES).
#define MT_DW_RC_INSTANCES(X) X(3, 2, 16, 14) X(3, 1, 24, 14) X(5, 2, 24, 7)       /* (k, stride, Cin, weight-gradient tile) */
extern "C" int mt_dwconv_rc_supported(int cin, int C, int k, int stride, int H);

extern "C" int mt_dwconv_bwd_rc(const float* du, const float* z, const float* kabc, const float* w, const float* y, const float* we,
                                int cin, const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in,
                                double* stats_in, int slots, float* dw, int N, int H, int W, int C, int k, int stride, int parts,
                                void* stream) {
  if (!du || !z || !kabc || !y || !we || !scale_in || !shift_in) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: null pointer");
  if (((uintptr_t)y | (uintptr_t)we) & 15) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: y and we must be 16-byte aligned");
  if (parts != 1 && parts != 2) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: parts must be 1 (weight) or 2 (data)");
  if (parts == 1 && !dw) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: weight part needs dw");
  if (parts == 2 && (!w || !du_in)) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: data part needs w and du_in");
  if (parts == 2 && ((stats_in == nullptr) != (mean_invstd_in == nullptr))) return fail(MT_ERR_ARG, "mt_dwconv_bwd_rc: stats_in and mean_invstd_in go together");
  if (stride == 2 && ((H & 1) || (W & 1))) return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd_rc: stride 2 needs even H, W");
  if (!mt_dwconv_rc_supported(cin, C, k, stride, H))
    return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd_rc: no instance for k=%d stride=%d Cin=%d C=%d H=%d", k, stride, cin, C, H);
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  hipStream_t s = (hipStream_t)stream;
#define MT_CASE(K_, S_, CIN_, T_)                                                                                               \
  if (k == K_ && stride == S_ && cin == CIN_) {                                                                                 \
    if (parts == 1)                                                                                                             \
      return launch_dw_wgrad_tiled<K_, S_, T_, 1, 16, CIN_>(du, z, kabc, y, scale_in, shift_in, dw, N, H, W, C, Ho, Wo, s, we); \
    return launch_dw_dgrad_tiled<K_, S_, 14, 1, 16, false, CIN_>(du, z, kabc, w, y, scale_in, shift_in, mean_invstd_in, du_in, \
                                                                  stats_in, slots, N, H, W, C, Ho, Wo, nullptr, nullptr, s,    \
                                                                  nullptr, we);                                                 \
  }
  MT_DW_RC_INSTANCES(MT_CASE)
#undef MT_CASE
  return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd_rc: no instance");
}

extern "C" int mt_dwconv_bwd_res2(const float* du, const float* z, const float* kabc, const float* w, const float* zin,
                                  const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in,
                                  double* stats_in, int slots, float* dw, int N, int H, int W, int C, int k, int stride, int parts,
                                  int act, const float* res_pre, const float* res_post, void* stream) {
  if (stride != 1) return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_bwd_res2: stride 1 only");
  g_res_stride = 2;
  const int rc = mt_dwconv_bwd(du, z, kabc, w, zin, scale_in, shift_in, mean_invstd_in, du_in, stats_in, slots, dw, N, H, W, C, k, stride,
                               parts, act, res_pre, res_post, stream);
  g_res_stride = 1;
  return rc;
}

static int stem_conv_wgrad(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N, int H, int W,
                           bool valid, void* stream) {
  if (!du || !z || !kabc || !x || !dw) return fail(MT_ERR_ARG, "mt_stem_conv_wgrad: null pointer");
  if (valid && (H < 3 || W < 3)) return fail(MT_ERR_ARG, "mt_stem_conv_wgrad_valid: crops of at least 3 x 3");
  const int Ho = valid ? (H - 3) / 2 + 1 : (H + 1) / 2, Wo = valid ? (W - 3) / 2 + 1 : (W + 1) / 2;
  const int padt = valid ? 0 : max((Ho - 1) * 2 + 3 - H, 0);
  // (deterministic mode takes the outer-product kernel: the MFMA form meets its row groups with LDS atomics)
  if (Wo <= 128 && !det_enabled()) return stem_wgrad_mfma(du, z, kabc, x, x_is_u8, dw, N, H, W, Ho, Wo, padt / 2, (hipStream_t)stream);
  DetScope det((hipStream_t)stream, 1, 1024, 32 * 27, true, true);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, du, z, kabc, x, x_is_u8, dw, N, H, W, Ho, Wo, padt / 2, det.log);
  const int rc = check_launch("mt_stem_conv_wgrad");
  return rc ? rc : det.reduce_f32(dw);
}

extern "C" int mt_stem_conv_wgrad(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N,
                                  int H, int W, void* stream) {
  return stem_conv_wgrad(du, z, kabc, x, x_is_u8, dw, N, H, W, false, stream);
}

extern "C" int mt_stem_conv_wgrad_valid(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N,
                                        int H, int W, void* stream) {
  return stem_conv_wgrad(du, z, kabc, x, x_is_u8, dw, N, H, W, true, stream);
}
