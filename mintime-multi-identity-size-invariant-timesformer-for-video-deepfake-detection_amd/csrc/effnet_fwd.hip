// EfficientNet-B0 forward kernels other than the 1x1 convolutions (those are mt_gemm), gfx950.
//
// Data layout decisions (MI355X-first):
//   * activations are NHWC ([N*H*W, C] "pixel-major") everywhere -> a 1x1 conv is a plain GEMM, the extractor's last
//     tensor is already the TimeSformer's token-major [B*F*49, C] operand, and channel vectors are float4-coalesced.
//   * what lives in HBM between kernels is the RAW convolution output z (pre-BatchNorm).  Every consumer applies
//     y = swish(z*scale[c] + shift[c]) while loading.  One code path serves eval mode (scale/shift from running
//     stats) and train mode (from batch statistics, which only exist after the producing kernel has finished), and the
//     backward pass needs z anyway (swish' and BN backward), so nothing is stored twice.
//   * every producer accumulates per-channel sum / sum-of-squares of z in fp64 (block partials in fp32, one fp64
//     atomic per block and channel into one of `slots` replicated accumulators) -> BatchNorm batch statistics cost no
//     extra pass over the tensor.
//
// Reference ops replaced (models/efficientnet/efficientnet_pytorch):
//   mt_stem_conv_fwd    _conv_stem (Conv2dStaticSamePadding k3 s2, utils.py:248-276; model.py:173,276)
//   mt_dwconv_fwd       _bn0/_swish on load + _depthwise_conv (model.py:98-103)
//   mt_bn_finalize      nn.BatchNorm2d statistics, running-stat update, affine folding (model.py:62,72,86,174,202)
//   mt_se_pool_fwd      _bn1 + _swish + F.adaptive_avg_pool2d (model.py:104-108)
//   mt_se_gate_fwd      _se_reduce + swish + _se_expand + sigmoid (model.py:109-112)
//   mt_bn_act_fwd       _bn2 (+ residual, model.py:117-127) and the head's _bn1 + swish (model.py:286)
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "planes.hpp"
#include "rc.hpp"
#include <stdlib.h>

using namespace mt;

#ifndef MT_DW_FWD_ROLL5
#define MT_DW_FWD_ROLL5 1
#endif

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   /* v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the swish kernels are VALU-bound */
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }

__device__ __forceinline__ float4 bn_swish4(float4 z, float4 sc, float4 sh) {
  float4 a;
  a.x = swishf_(fmaf(z.x, sc.x, sh.x)); a.y = swishf_(fmaf(z.y, sc.y, sh.y));
  a.z = swishf_(fmaf(z.z, sc.z, sh.z)); a.w = swishf_(fmaf(z.w, sc.w, sh.w));
  return a;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { atomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------ depthwise, LDS-tiled
// One block = one 16-channel chunk, grid-strided over T x T output tiles.  The activated input tile (with halo) is built
// once in LDS (swish evaluated once per element, not once per tap); thread (cq = tid&3, slot = tid>>2) then produces the
// outputs slot, slot+64, ... of the tile for its channel quad with its K*K weights held in registers.  BatchNorm sums
// stay in registers across the block's tiles -> one fp64 atomic per channel per block.
template <int ACT>
__device__ __forceinline__ float4 act_affine4(float4 z, float4 sc, float4 sh) {
  float4 a = make_float4(fmaf(z.x, sc.x, sh.x), fmaf(z.y, sc.y, sh.y), fmaf(z.z, sc.z, sh.z), fmaf(z.w, sc.w, sh.w));
  if (ACT == 1) { a.x = swishf_(a.x); a.y = swishf_(a.y); a.z = swishf_(a.z); a.w = swishf_(a.w); }
  if (ACT == 2) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  return a;
}

// ACT: input activation applied after the per-channel affine (0 none, 1 swish [EfficientNet], 2 relu [Xception]).
// CC : channels per block (16, or 8 for channel counts like Xception's 728 that are not multiples of 16).
// RC : 0 = zin is the conv's raw input [N,H,W,C]; Cin > 0 = zin is the BLOCK input y [N,H,W,Cin] and the raw input chunk is rebuilt
//      as y . We^T while the tile is staged (rc.hpp; we = the expand weight [C, Cin]) -- the expanded tensor is not read.
template <int K, int S, int T, int ACT, int CC, int RC = 0>
__global__ __launch_bounds__(256) void dwconv_tiled_kernel(const float* __restrict__ zin, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ w,
                                                           float* __restrict__ zout, double* __restrict__ stats, int slots,
                                                           int N, int H, int W, int C, int Ho, int Wo, int pad0, PlaneRef po,
                                                           const float* __restrict__ we) {
  static_assert(RC == 0 || CC == 16, "the recompute yields 16-channel chunks");
  constexpr int CQN = CC / 4, NSLOT = 256 / CQN;
  constexpr int IH = (T - 1) * S + K;
  constexpr int IWP = IH | 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // a_t [IH][IWP][CC]; later the stats reduction buffer
  const int tid = threadIdx.x;
  const int cq = tid % CQN, slot = tid / CQN;
  int chunk_id; int64_t tile0, tile_stride;
  xcd_chunk_tile(C / CC, chunk_id, tile0, tile_stride);
  const int c0 = chunk_id * CC;
  const int ty_n = (Ho + T - 1) / T, tx_n = (Wo + T - 1) / T;
  const int64_t ntiles = (int64_t)N * ty_n * tx_n;
  // 3x3 weights live in registers; the 25 float4 of a 5x5 would cost 100 VGPRs on top of the prefetch registers (occupancy 1),
  // so they sit in LDS behind the input tile ([K*K][CC], read as 4 broadcast addresses per wavefront)
  constexpr bool WLDS = K > 3;
  constexpr bool ROLL5 = MT_DW_FWD_ROLL5;
  float* w_t = lds + IH * IWP * CC;
  float4 wt[WLDS ? 1 : K * K];
  {
    const int c = c0 + cq * 4;
    if constexpr (WLDS) {
      if (slot == 0)
        for (int i = 0; i < K * K; ++i)
          *reinterpret_cast<float4*>(w_t + i * CC + cq * 4) =
              make_float4(w[(c + 0) * K * K + i], w[(c + 1) * K * K + i], w[(c + 2) * K * K + i], w[(c + 3) * K * K + i]);
    } else {
#pragma unroll
      for (int i = 0; i < K * K; ++i)
        wt[i] = make_float4(w[(c + 0) * K * K + i], w[(c + 1) * K * K + i], w[(c + 2) * K * K + i], w[(c + 3) * K * K + i]);
    }
  }
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  // Input tiles are software-pipelined through registers: the global loads of the block's NEXT tile are in flight while the
  // current one is convolved out of LDS (strided / 5x5 tiles are load-heavy: 4.3 input pixels per output at stride 2).
  // Loads are unconditional on clamped addresses + a validity mask; a predicated `ok ? *p : 0` would de-pipeline them.
  constexpr int NSL = 256 / CQN;                           // pixels covered per pass of the block
  constexpr int NL = (IH * IH + NSL - 1) / NSL;            // float4 slots per thread
  static_assert(NL <= 32, "validity mask is 32 bits");
  // staging role: (pixel slot, channel quad) of the element this thread brings into the LDS tile.  Plain: the compute role's.
  // RC: the MFMA result layout -- lane l of a wavefront holds pixel (l & 15) of its 16-pixel group, channel quad l >> 4.
  const int lane = tid & 63;
  const int f_q = RC ? (lane >> 4) : cq, f_slot = RC ? ((tid >> 6) * 16 + (lane & 15)) : slot;
  const float4 sc_q = *reinterpret_cast<const float4*>(scale + c0 + f_q * 4);
  const float4 sh_q = *reinterpret_cast<const float4*>(shift + c0 + f_q * 4);
  constexpr int RCW = RC ? RC : 8;
  RcFrag<RCW> wfrag, ypre[RC ? NL : 1];
  if constexpr (RC > 0) rc_load<RCW>(wfrag, we + (int64_t)(c0 + (lane & 15)) * RC, lane);
  float4 pre[RC ? 1 : NL];
  unsigned pre_ok = 0;
  auto fetch = [&](int64_t tile) {
    int n, ty, tx;
    tile_nyx(tile, ty_n, tx_n, n, ty, tx);
    const int ih0 = ty * T * S - pad0, iw0 = tx * T * S - pad0;
    const float* img = RC ? zin + (int64_t)n * H * W * RC : zin + (int64_t)n * H * W * C + c0 + cq * 4;
    pre_ok = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pix = f_slot + i * NSL;
      const int iy = pix / IH, ix = pix - iy * IH;
      const int ih = ih0 + iy, iw = iw0 + ix;
      if (pix < IH * IH && ih >= 0 && ih < H && iw >= 0 && iw < W) pre_ok |= 1u << i;
      const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
      if constexpr (RC > 0) rc_load<RCW>(ypre[i], img + ((int64_t)ihc * W + iwc) * RC, lane);
      else pre[i] = *reinterpret_cast<const float4*>(img + ((int64_t)ihc * W + iwc) * C);
    }
  };
  int64_t tile = tile0;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tile_stride) {
    int n, ty, tx;
    tile_nyx(tile, ty_n, tx_n, n, ty, tx);
    const int oh0 = ty * T, ow0 = tx * T;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int pix = f_slot + i * NSL;
      float4 zraw;
      if constexpr (RC > 0) zraw = rc_mma<RCW>(wfrag, ypre[i]);       // (all lanes: the MFMA runs outside the range test)
      else zraw = pre[i];
      if (pix < IH * IH) {
        const int iy = pix / IH, ix = pix - iy * IH;
        const float4 a = act_affine4<ACT>(zraw, sc_q, sh_q);
        const bool ok = (pre_ok >> i) & 1u;
        *reinterpret_cast<float4*>(lds + (iy * IWP + ix) * CC + f_q * 4) =
            make_float4(ok ? a.x : 0.f, ok ? a.y : 0.f, ok ? a.z : 0.f, ok ? a.w : 0.f);
      }
    }
    __syncthreads();
    if (tile + tile_stride < ntiles) fetch(tile + tile_stride);
    for (int p = slot; p < T * T; p += NSLOT) {
      const int oy = p / T, ox = p - oy * T;
      const int oh = oh0 + oy, ow = ow0 + ox;
      if (oh < Ho && ow < Wo) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* base = lds + ((oy * S) * IWP + ox * S) * CC + cq * 4;
        auto row = [&](int kh) {
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const float4 a = *reinterpret_cast<const float4*>(base + (kh * IWP + kw) * CC);
            float4 ww;
            if constexpr (WLDS) ww = *reinterpret_cast<const float4*>(w_t + (kh * K + kw) * CC + cq * 4);
            else ww = wt[kh * K + kw];
            acc.x = fmaf(a.x, ww.x, acc.x); acc.y = fmaf(a.y, ww.y, acc.y);
            acc.z = fmaf(a.z, ww.z, acc.z); acc.w = fmaf(a.w, ww.w, acc.w);
          }
        };
        if constexpr (WLDS && ROLL5) {
#pragma unroll 1
          for (int kh = 0; kh < K; ++kh) row(kh);         // 5x5: one kernel row (5 + 5 LDS reads) in flight -- see effnet_bwd.hip K7
        } else {
#pragma unroll
          for (int kh = 0; kh < K; ++kh) row(kh);
        }
        if (po.p) {
          // the output as operand planes (Xception: the pointwise convolution that follows reads nothing else); the thread that owns
          // the last channels also zeroes the padding columns of the last 16-column block
          const int row = (n * Ho + oh) * Wo + ow, c = c0 + cq * 4;
          planes_store4(po, row, c, acc.x, acc.y, acc.z, acc.w);
          if (c + 4 == C)
            for (int cz = C; cz < po.cb16 * 16; cz += 4) planes_store4(po, row, cz, 0.f, 0.f, 0.f, 0.f);
        } else {
          *reinterpret_cast<float4*>(zout + (((int64_t)n * Ho + oh) * Wo + ow) * C + c0 + cq * 4) = acc;
        }
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x += acc.x * acc.x; s2.y += acc.y * acc.y; s2.z += acc.z * acc.z; s2.w += acc.w * acc.w;
      }
    }
  }
  if (stats) {
    __syncthreads();
    float* rr = lds + tid * 8;          // [256][8]
    rr[0] = s1.x; rr[1] = s1.y; rr[2] = s1.z; rr[3] = s1.w; rr[4] = s2.x; rr[5] = s2.y; rr[6] = s2.z; rr[7] = s2.w;
    __syncthreads();
    if (tid < CQN * 8) {                // (cq, e)
      const int q = tid >> 3, e = tid & 7;
      float v = 0.f;
      for (int sl = 0; sl < NSLOT; ++sl) v += lds[(sl * CQN + q) * 8 + e];
      const int ch = c0 + q * 4 + (e & 3), which = e >> 2;
      stat_add(stats + ((int64_t)(blockIdx.x % stat_slots(slots)) * 2 + which) * C + ch, stat_limb(slots, C), v);
    }
  }
}

template <int K, int S, int T, int ACT, int CC, int RC = 0>
int launch_dw_tiled(const float* zin, const float* scale, const float* shift, const float* w, float* zout, double* stats, int slots,
                    int N, int H, int W, int C, int Ho, int Wo, int pad0, hipStream_t s, const PlaneRef& po,
                    const float* we = nullptr) {
  constexpr int IH = (T - 1) * S + K;
  constexpr int IWP = IH | 1;
  size_t lds = (size_t)(IH * IWP * CC + (K > 3 ? K * K * CC : 0)) * sizeof(float);
  if (lds < 256 * 8 * sizeof(float)) lds = 256 * 8 * sizeof(float);
  const int chunks = C / CC;
  const int64_t ntiles = (int64_t)N * ((Ho + T - 1) / T) * ((Wo + T - 1) / T);
  const unsigned bx = xcd_chunk_grid(chunks, ntiles, 8192);
  auto k = dwconv_tiled_kernel<K, S, T, ACT, CC, RC>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_dwconv_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3(bx), dim3(256), lds, s, zin, scale, shift, w, zout, stats, slots != 0 ? slots : 1, N, H,
                     W, C, Ho, Wo, pad0, po, we);
  return check_launch("mt_dwconv_fwd(tiled)");
}

template <int K, int S, int ACT>
int launch_dw_tiled_any(const float* zin, const float* scale, const float* shift, const float* w, float* zout, double* stats,
                        int slots, int N, int H, int W, int C, int Ho, int Wo, int pad0, hipStream_t s, const PlaneRef& po) {
  static const int t7_mask = getenv("MT_DW_T7") ? atoi(getenv("MT_DW_T7")) : 0;       // lab: bit 0 = 7 x 7 tiles in the forward kernel
  const bool t14 = Ho >= 14 && !(t7_mask & 1);
  if (C % 16 == 0) {
    if (t14) return launch_dw_tiled<K, S, 14, ACT, 16>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad0, s, po);
    return launch_dw_tiled<K, S, 7, ACT, 16>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad0, s, po);
  }
  if (t14) return launch_dw_tiled<K, S, 14, ACT, 8>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad0, s, po);
  return launch_dw_tiled<K, S, 7, ACT, 8>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad0, s, po);
}

// ------------------------------------------------------------------------------------------------ BatchNorm finalize
// train: mean/var from the fp64 accumulators (biased var normalises, unbiased var updates running_var), momentum update
// eval : running stats.  Either way emits scale = gamma*invstd, shift = beta - mean*scale, and (mean, invstd) for backward.
// block = 32 channels x 8 slot groups: the 2 x `slots` fp64 partial sums of a channel are fetched by 8 threads in parallel (one thread
// walking all 32 slots is 64 dependent-latency loads: 13 us for a kernel that runs 98 times per step)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats, int slots, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_invstd, int C, float eps, float momentum,
                                                          int training) {
  __shared__ double red[8][32][2];
  const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  if (training) {
    double s = 0.0, q = 0.0;
    if (c < C)
      for (int i = sg; i < stat_slots(slots); i += 8) {
        s += stat_get(stats + ((int64_t)i * 2) * C + c, stat_limb(slots, C));
        q += stat_get(stats + ((int64_t)i * 2 + 1) * C + c, stat_limb(slots, C));
      }
    red[sg][cl][0] = s; red[sg][cl][1] = q;
    __syncthreads();
  }
  if (sg != 0 || c >= C) return;
  float mean, var;
  if (training) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) { s += red[g][cl][0]; q += red[g][cl][1]; }
    const double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
    if (running_mean) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  } else {
    mean = running_mean[c]; var = running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (mean_invstd) { mean_invstd[c] = mean; mean_invstd[C + c] = invstd; }
}

// ------------------------------------------------------------------------------------------------ SE: pooling
// pooled[n, c] = mean over hw of swish(z[n,hw,c]*scale+shift).   grid (N, ceil(CQ/CQB)); block = CQB x PB
__global__ __launch_bounds__(256) void se_pool_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ pooled,
                                                      int HW, int C, int CQB, int PB) {
  extern __shared__ float red[];   // [PB][CQB*4]
  const int tid = threadIdx.x;
  const int cql = tid % CQB, pl = tid / CQB;
  const int cq = blockIdx.y * CQB + cql;
  const int CQ = C >> 2;
  const int n = blockIdx.x;
  // blockIdx.z = slice of the image's pixels (one image per block would leave a 256-crop batch at 1 block per CU)
  const int per = (HW + gridDim.z - 1) / gridDim.z;
  const int p_lo = blockIdx.z * per, p_hi = min(HW, p_lo + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < PB && cq < CQ) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + cq * 4);
    const float* base = z + (int64_t)n * HW * C + cq * 4;
    int p = p_lo + pl;
    for (; p + 3 * PB < p_hi; p += 4 * PB) {     // 4 independent loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)p * C);
      const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)(p + PB) * C);
      const float4 v2 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 2 * PB) * C);
      const float4 v3 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 3 * PB) * C);
      const float4 a0 = bn_swish4(v0, sc, sh), a1 = bn_swish4(v1, sc, sh), a2 = bn_swish4(v2, sc, sh), a3 = bn_swish4(v3, sc, sh);
      s.x += (a0.x + a1.x) + (a2.x + a3.x); s.y += (a0.y + a1.y) + (a2.y + a3.y);
      s.z += (a0.z + a1.z) + (a2.z + a3.z); s.w += (a0.w + a1.w) + (a2.w + a3.w);
    }
    for (; p < p_hi; p += PB) {
      const float4 a = bn_swish4(*reinterpret_cast<const float4*>(base + (int64_t)p * C), sc, sh);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  if (pl < PB) *reinterpret_cast<float4*>(red + (pl * CQB + cql) * 4) = s;
  __syncthreads();
  if (pl == 0 && cq < CQ) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < PB; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(red + (q * CQB + cql) * 4);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const float inv = 1.0f / (float)HW;
    // partial[n][slice][c]: the slices are summed, in a fixed order, by the gate kernel (no atomics: eval stays deterministic)
    float* out = pooled + ((int64_t)n * gridDim.z + blockIdx.z) * C + cq * 4;
    *reinterpret_cast<float4*>(out) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
  }
}

// ------------------------------------------------------------------------------------------------ SE: gate
// gate[n,c] = sigmoid(W2[c,:] . swish(W1 . pooled[n,:] + b1) + b2[c]).
// Two tiny matrix products per image (C x CS with CS = 4..48).  One block per image doing both was latency-bound at 88 us for
// C = 1152 (a wavefront walking one W1 row at a time: C/64 dependent round trips per row; a thread per channel walking its W2 row:
// 64 cache lines per wave-instruction) -- 0.57 ms per step for 30 MFLOP.  Now: (A) hidden pre-activations, a wavefront per four W1
// rows walked together (independent loads in flight), ceil(CS/16) blocks per image; (B) the gate, one block per (image, 128-channel
// slab): the slab's W2 rows are copied coalesced into LDS ([128][CS+1], odd pitch) and multiplied from there.
constexpr int SE_SLAB = 128;
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ partial, int parts, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, float* __restrict__ pooled,
                                                        float* __restrict__ hidden, int C, int CS) {
  extern __shared__ float pv[];    // pooled[C]
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C; i += 256) {
    float v = 0.f;
    for (int q = 0; q < parts; ++q) v += partial[((int64_t)n * parts + q) * C + i];
    pv[i] = v;
    if (pooled && blockIdx.y == 0) pooled[(int64_t)n * C + i] = v;
  }
  __syncthreads();
  const int j0 = (blockIdx.y * 4 + wave) * 4;
  if (j0 >= CS) return;
  const float* r0 = w1 + (int64_t)min(j0 + 0, CS - 1) * C;     // rows past CS alias the last one (results dropped): no branch
  const float* r1 = w1 + (int64_t)min(j0 + 1, CS - 1) * C;     // around the loads, so all four stay in flight together
  const float* r2 = w1 + (int64_t)min(j0 + 2, CS - 1) * C;
  const float* r3 = w1 + (int64_t)min(j0 + 3, CS - 1) * C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
  for (int i = lane; i < C; i += 64) {
    const float w0 = r0[i], w1v = r1[i], w2v = r2[i], w3 = r3[i], pvi = pv[i];
    a0 = fmaf(w0, pvi, a0); a1 = fmaf(w1v, pvi, a1); a2 = fmaf(w2v, pvi, a2); a3 = fmaf(w3, pvi, a3);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); a3 += __shfl_xor(a3, o);
  }
  if (lane < 4 && j0 + lane < CS) {
    const float a = lane == 0 ? a0 : (lane == 1 ? a1 : (lane == 2 ? a2 : a3));
    hidden[(int64_t)n * CS + j0 + lane] = a + b1[j0 + lane];
  }
}

// loads `count` consecutive floats of src into the [rows][CS+1] slab image, eight loads in flight per thread
__device__ __forceinline__ void se_load_slab(float* slab, const float* __restrict__ src, int count, int CS, int tid) {
  const int pitch = CS + 1;
  const float inv_cs = 1.0f / (float)CS;
  for (int e0 = tid; e0 < count; e0 += 256 * 8) {
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

__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ hidden, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ gate, int C, int CS) {
  extern __shared__ float sm[];    // hid[CS] + slab[SE_SLAB][CS+1]
  float* hid = sm;
  float* slab = sm + CS;
  const int n = blockIdx.x, c0 = blockIdx.y * SE_SLAB, tid = threadIdx.x;
  const int rows = min(SE_SLAB, C - c0), pitch = CS + 1;
  if (tid < CS) hid[tid] = swishf_(hidden[(int64_t)n * CS + tid]);
  se_load_slab(slab, w2 + (int64_t)c0 * CS, rows * CS, CS, tid);
  __syncthreads();
  // two threads per channel, each half of the CS taps; the halves meet in the slab's padding column
  const int r = tid & (SE_SLAB - 1), half = tid >> 7;
  float a = 0.f;
  if (r < rows) {
    const int j0 = half ? (CS + 1) / 2 : 0, j1 = half ? CS : (CS + 1) / 2;
    for (int j = j0; j < j1; ++j) a = fmaf(slab[r * pitch + j], hid[j], a);
  }
  if (half && r < rows) slab[r * pitch + CS] = a;
  __syncthreads();
  if (!half && r < rows) gate[(int64_t)n * C + c0 + r] = sigmoidf_(a + slab[r * pitch + CS] + b2[c0 + r]);
}

// ------------------------------------------------------------------------------------------------ BN apply (+swish) (+residual)
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ res,
                                                     float* __restrict__ y, int64_t total4, int CQ, int act,
                                                     const float* __restrict__ rowscale, int rows_per_group) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int cq = (int)(i % CQ);
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq];
    const float4 sh = reinterpret_cast<const float4*>(shift)[cq];
    float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (act == 1) { o.x = swishf_(o.x); o.y = swishf_(o.y); o.z = swishf_(o.z); o.w = swishf_(o.w); }
    if (act == 2) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (rowscale) {   // per-sample drop-connect gate (utils.py:129-154)
      const float g = rowscale[(i / CQ) / rows_per_group];
      o.x *= g; o.y *= g; o.z *= g; o.w *= g;
    }
    if (res) { const float4 r = reinterpret_cast<const float4*>(res)[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

int pick_cqb(int CQ) {
  int best = 1;
  for (int d = 1; d <= 64 && d <= CQ; ++d)
    if (CQ % d == 0) best = d;
  return best;
}

}  // namespace

namespace {
template <int K, int S>
int launch_dw(const float* zin, const float* scale, const float* shift, const float* w, float* zout, double* stats, int slots,
              int N, int H, int W, int C, int act, hipStream_t s, const PlaneRef& po) {
  const int Ho = (H + S - 1) / S, Wo = (W + S - 1) / S;
  const int pad = max((Ho - 1) * S + K - H, 0) / 2;              // TF-SAME pad-before
  if (act == 1) return launch_dw_tiled_any<K, S, 1>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad, s, po);
  if (act == 2) return launch_dw_tiled_any<K, S, 2>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad, s, po);
  return launch_dw_tiled_any<K, S, 0>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad, s, po);
}

int dwconv_fwd(const float* zin, const float* scale, const float* shift, const float* w, float* zout, double* stats, int slots, int N,
               int H, int W, int C, int k, int stride, int act, hipStream_t s, const PlaneRef& po, const char* who) {
  if (C & 7) return fail(MT_ERR_ARG, "%s: C must be a multiple of 8 (got %d)", who, C);
  if (act < 0 || act > 2) return fail(MT_ERR_ARG, "%s: act must be 0, 1 or 2", who);
  if (k == 3 && stride == 1) return launch_dw<3, 1>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, act, s, po);
  if (k == 3 && stride == 2) return launch_dw<3, 2>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, act, s, po);
  if (k == 5 && stride == 1) return launch_dw<5, 1>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, act, s, po);
  if (k == 5 && stride == 2) return launch_dw<5, 2>(zin, scale, shift, w, zout, stats, slots, N, H, W, C, act, s, po);
  return fail(MT_ERR_UNSUPPORTED, "%s: k=%d stride=%d unsupported", who, k, stride);
}
}  // namespace

extern "C" int mt_dwconv_fwd(const float* zin, const float* scale, const float* shift, const float* w, float* zout,
                             double* stats, int slots, int N, int H, int W, int C, int k, int stride, int act, void* stream) {
  if (!zin || !scale || !shift || !w || !zout) return fail(MT_ERR_ARG, "mt_dwconv_fwd: null pointer");
  return dwconv_fwd(zin, scale, shift, w, zout, stats, slots, N, H, W, C, k, stride, act, (hipStream_t)stream, PlaneRef{nullptr, 0, 0, 0},
                    "mt_dwconv_fwd");
}

// instances of the recompute form: (k, stride, Cin, output tile) of EfficientNet-B0's blocks 1-3 (the expanded tensors of the 112^2 /
// 56^2 grids).  5x5 stride 2 takes 7 x 7 tiles: a 14 x 14 tile's 31 x 31 halo is 16 operand fragments of prefetch per thread (occupancy 1).
#define MT_DW_RC_INSTANCES(X) X(3, 2, 16, 14) X(3, 1, 24, 14) X(5, 2, 24, 7)

extern "C" int mt_dwconv_rc_supported(int cin, int C, int k, int stride, int H) {
  if (C <= 0 || (C & 15) || (H + stride - 1) / stride < 14) return 0;
#define MT_CASE(K_, S_, CIN_, T_) if (k == K_ && stride == S_ && cin == CIN_) return 1;
  MT_DW_RC_INSTANCES(MT_CASE)
#undef MT_CASE
  return 0;
}

extern "C" int mt_dwconv_fwd_rc(const float* y, const float* we, int cin, const float* scale, const float* shift, const float* w,
                                float* zout, double* stats, int slots, int N, int H, int W, int C, int k, int stride, void* stream) {
  if (!y || !we || !scale || !shift || !w || !zout) return fail(MT_ERR_ARG, "mt_dwconv_fwd_rc: null pointer");
  if (((uintptr_t)y | (uintptr_t)we) & 15) return fail(MT_ERR_ARG, "mt_dwconv_fwd_rc: y and we must be 16-byte aligned");
  if (!mt_dwconv_rc_supported(cin, C, k, stride, H))
    return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_fwd_rc: no instance for k=%d stride=%d Cin=%d C=%d H=%d", k, stride, cin, C, H);
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int pad = max((Ho - 1) * stride + k - H, 0) / 2;
  const PlaneRef po{nullptr, 0, 0, 0};
#define MT_CASE(K_, S_, CIN_, T_)                                                                                              \
  if (k == K_ && stride == S_ && cin == CIN_)                                                                                  \
    return launch_dw_tiled<K_, S_, T_, 1, 16, CIN_>(y, scale, shift, w, zout, stats, slots, N, H, W, C, Ho, Wo, pad, (hipStream_t)stream, po, we);
  MT_DW_RC_INSTANCES(MT_CASE)
#undef MT_CASE
  return fail(MT_ERR_UNSUPPORTED, "mt_dwconv_fwd_rc: no instance");
}

extern "C" int mt_dwconv_fwd_planes(const float* zin, const float* scale, const float* shift, const float* w, void* planes, int N, int H,
                                    int W, int C, int k, int stride, int act, void* stream) {
  if (!zin || !scale || !shift || !w || !planes) return fail(MT_ERR_ARG, "mt_dwconv_fwd_planes: null pointer");
  if ((uintptr_t)planes & 15) return fail(MT_ERR_ARG, "mt_dwconv_fwd_planes: planes must be 16-byte aligned");
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int64_t rows = (int64_t)N * Ho * Wo;
  if (rows > INT32_MAX - 32) return fail(MT_ERR_ARG, "mt_dwconv_fwd_planes: too many rows");
  const int cb16 = (C + 15) >> 4, rp = ((int)rows + 31) & ~31;
  const PlaneRef po{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
  hipStream_t s = (hipStream_t)stream;
  if (rp != rows)        // the padding rows of the last 32-row block: zero the block, the kernel then writes its real rows
    for (int pl = 0; pl < 3; ++pl)
      if (hipMemsetAsync(po.p + pl * po.pstride + (int64_t)(rp / 32 - 1) * cb16 * 512, 0, (size_t)cb16 * 512 * sizeof(__bf16), s) != hipSuccess)
        return fail(MT_ERR_LAUNCH, "mt_dwconv_fwd_planes: memset failed");
  return dwconv_fwd(zin, scale, shift, w, nullptr, nullptr, 1, N, H, W, C, k, stride, act, s, po, "mt_dwconv_fwd_planes");
}

extern "C" int mt_bn_finalize(const double* stats, int slots, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* scale, float* shift, float* mean_invstd,
                              int C, float eps, float momentum, int training, void* stream) {
  if (!gamma || !beta || !scale || !shift) return fail(MT_ERR_ARG, "mt_bn_finalize: null pointer");
  if (training && !stats) return fail(MT_ERR_ARG, "mt_bn_finalize: training needs stats");
  if (!training && (!running_mean || !running_var)) return fail(MT_ERR_ARG, "mt_bn_finalize: eval needs running stats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, stats, slots, count, gamma,
                     beta, running_mean, running_var, scale, shift, mean_invstd, C, eps, momentum, training);
  return check_launch("mt_bn_finalize");
}

static int se_pool_parts(int N, int HW, int C) {
  // enough blocks to fill the chip: slice each image's pixels while a slice still has >= 16 pixels per thread
  const int CQ = C / 4, CQB = pick_cqb(CQ), PB = 256 / CQB;
  int parts = 1;
  while ((int64_t)N * (CQ / CQB) * parts < 2048 && HW / (parts * 2) >= PB * 16) parts *= 2;
  return parts;
}

extern "C" int mt_se_pool_parts(int N, int HW, int C) { return (C & 3) || C <= 0 ? 1 : se_pool_parts(N, HW, C); }

extern "C" int mt_se_pool_fwd(const float* z, const float* scale, const float* shift, float* partial, int N, int HW, int C,
                              int parts, void* stream) {
  if (!z || !scale || !shift || !partial) return fail(MT_ERR_ARG, "mt_se_pool_fwd: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_se_pool_fwd: C %% 4 != 0");
  if (parts != se_pool_parts(N, HW, C)) return fail(MT_ERR_ARG, "mt_se_pool_fwd: parts must come from mt_se_pool_parts");
  const int CQ = C / 4, CQB = pick_cqb(CQ), PB = 256 / CQB;
  hipLaunchKernelGGL(se_pool_kernel, dim3(N, CQ / CQB, parts), dim3(CQB * PB), (size_t)PB * CQB * 4 * sizeof(float),
                     (hipStream_t)stream, z, scale, shift, partial, HW, C, CQB, PB);
  return check_launch("mt_se_pool_fwd");
}

extern "C" int mt_se_gate_fwd(const float* partial, int parts, const float* w1, const float* b1, const float* w2, const float* b2,
                              float* pooled, float* gate, float* hidden, int N, int C, int CS, void* stream) {
  if (!partial || !w1 || !b1 || !w2 || !b2 || !gate || !hidden || parts < 1) return fail(MT_ERR_ARG, "mt_se_gate_fwd: null pointer");
  if (CS < 1 || CS > 256) return fail(MT_ERR_UNSUPPORTED, "mt_se_gate_fwd: squeeze width %d", CS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(se_hidden_kernel, dim3(N, (CS + 15) / 16), dim3(256), (size_t)C * sizeof(float), s, partial, parts, w1, b1, pooled,
                     hidden, C, CS);
  int rc = check_launch("mt_se_gate_fwd(hidden)");
  if (rc) return rc;
  hipLaunchKernelGGL(se_gate_kernel, dim3(N, (C + SE_SLAB - 1) / SE_SLAB), dim3(256), (size_t)(CS + SE_SLAB * (CS + 1)) * sizeof(float), s,
                     hidden, w2, b2, gate, C, CS);
  return check_launch("mt_se_gate_fwd");
}

extern "C" int mt_bn_act_fwd(const float* z, const float* scale, const float* shift, const float* res, float* y,
                             int64_t rows, int C, int act, const float* rowscale, int rows_per_group, void* stream) {
  if (!z || !scale || !shift || !y) return fail(MT_ERR_ARG, "mt_bn_act_fwd: null pointer");
  if (C & 3) return fail(MT_ERR_ARG, "mt_bn_act_fwd: C %% 4 != 0");
  const int64_t total4 = rows * (C / 4);
  int64_t nb = (total4 + 255) / 256; if (nb > 4096) nb = 4096; const int blocks = (int)nb;
  hipLaunchKernelGGL(bn_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, scale, shift, res, y, total4, C / 4, act,
                     rowscale, rows_per_group > 0 ? rows_per_group : 1);
  return check_launch("mt_bn_act_fwd");
}
