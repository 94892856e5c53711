// fp32 MFMA GEMM family for gfx950 (MI355X).
//
//   C[M,N] (+)= op(A)[M,K] * op(B)[K,N]      exact fp32: v_mfma_f32_32x32x2_f32 (157 TF peak, no TF32 on CDNA4)
//
// One template covers every dense contraction of the MINTIME hot path:
//   * Linear / 1x1-conv forward   (A rows k-contiguous, B = torch weight [N,K] k-contiguous)
//   * dgrad                        (A = dY k-contiguous, B = weight [Kc,N] n-contiguous)
//   * wgrad (split-K, atomics)     (A = dY^T m-contiguous, B = X n-contiguous)
// with the elementwise work of the reference's separate passes folded into the operand load
// (prologue: BN-affine + swish + squeeze-excite gate, ...) and the accumulator store
// (epilogue: bias, residual, GEGLU, BN batch-statistics, ...).
//
// Tiling: block = WAVES_M x WAVES_N wavefronts (64 lanes), each owning TM x TN 32x32 MFMA tiles.
// LDS tiles are k-major ([BK][rows+pad]): lane l reads As[k0 + (l>>5)][row0 + (l&31)] -> 32 consecutive
// floats per half-wave = conflict-free ds_read_b32, exactly the 32x32x2 operand layout
// (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]).  fp32 MFMA issues once per 64 cycles per SIMD, so one b32
// read per operand per MFMA is far below LDS bandwidth; the design effort goes into the global side:
// register-staged double buffering (next tile's global loads in flight under the MFMAs), float4
// global accesses, XCD-aware block order so an A row-panel is reused out of one XCD's L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hpp"
#include "det.hpp"

#ifndef MT_BK
#define MT_BK 16
#endif
#ifndef MT_MIN_WAVES
#define MT_MIN_WAVES 3
#endif

namespace mt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int { LAYOUT_KCONTIG = 0, LAYOUT_KMAJOR = 1 };

// prologues applied to A elements on the global->LDS path
enum : int {
  PRO_NONE = 0,
  PRO_BN_SWISH_GATE = 1,   // a = swish(z*scale[k]+shift[k]) * gate[(m/hw)*K + k]      (MBConv project conv input)
  PRO_BN_SWISH = 2,        // a = swish(z*scale[k]+shift[k])
  PRO_AFFINE = 3,          // a = z*scale[k]+shift[k]
  PRO_IM2COL = 5,          // a = act(x[n, oh*s+kh-p, ow*s+kw-p, ci]*scale[ci]+shift[ci]) gathered on the fly from an NHWC image
                           //     (dense k x k convolution / strided 1x1 as a GEMM; m = (n,oh,ow), k = (kh,kw,ci); no im2col buffer)
  PRO_BN_BWD = 4,          // a = ka[c]*A[.] + kb[c]*A2[.] + kc[c]   (BatchNorm backward folded into the operand load:
                           //     A = d(bn output), A2 = z (bn input); ka,kb,kc = scale, shift, gate vectors; c = channel)
  PRO_IM2COL_ANY = 15,     // internal: PRO_IM2COL by the generic per-element gather (3-channel / uint8 images, k > 5); PRO_IM2COL itself is
                           //     the granule form (float image, C % 4 == 0: gemm.hip im2col_granule_ok)
};

// prologue applied to B elements (k-major B only): B = swish(z*b_scale[n]+b_shift[n]) * b_gate[(k/b_hw)*N + n]
enum : int { BPRO_NONE = 0, BPRO_BN_SWISH_GATE = 1, BPRO_IM2COL = 2, BPRO_IM2COL_ANY = 3 };   // BPRO_IM2COL: B[k=(n,oh,ow)][n=(kh,kw,ci)] gathered (conv wgrad); _ANY: see PRO_IM2COL_ANY

// epilogues
enum : int {
  EPI_STORE = 0,        // C = acc (+bias[n])
  EPI_BIAS_RES = 1,     // C = acc + bias[n] + R[m,n]
  EPI_GEGLU = 2,        // h[m,j] = (acc_a+ba[j]) * gelu(acc_g+bg[j]);  optional u store (pre-activations, interleaved a_j,g_j)
  EPI_STATS = 3,        // C = acc ; per-column sum / sum-of-squares accumulated in fp64 (BatchNorm batch statistics)
  EPI_ATOMIC = 4,       // C += acc (+bias on the first K-slice) via fp32 atomics (split-K); C pre-zeroed or holding the residual
  EPI_GEGLU_BWD = 5,    // acc = dh[m,j]; du[m,j] = dh*gelu(g), du[m,Nh+j] = dh*a*gelu'(g)   (a,g from u)
  EPI_ACCUM = 6,        // C = C + acc (+bias)  -- gradient accumulation into an existing tensor
  // MBConv backward, squeeze-excite stage, with acc = da (the project conv's data gradient, never written to memory) and
  // z = C2[m,n] the depthwise conv's raw output, u = z*e_scale[n] + e_shift[n], img = m / e_hw:
  EPI_SE_RED = 7,       // C[img, n] += sum over the image's rows of acc * swish(u)           (d gate; fp32 atomics, C pre-zeroed)
  EPI_ACT_BWD = 8,      // C[m,n] = du = (acc*e_gate[img,n] + e_dpool[img,n]/e_hw) * swish'(u);  BatchNorm-backward sums of du:
                        //   stats[..][0][n] += du, stats[..][1][n] += du * (z - e_mi[n]) * e_mi[N+n]   (fp64 slots like EPI_STATS)
};

struct RowMap {   // out_row = (r / gin) * gout + off + r % gin   (gin == 0 -> identity)
  int gin, gout, off;
};

__device__ __forceinline__ int64_t map_row(const RowMap& rm, int r) {
  if (rm.gin == 0) return r;
  int g = r / rm.gin;
  return (int64_t)g * rm.gout + rm.off + (r - g * rm.gin);
}

struct ConvDesc {   // geometry of an im2col prologue
  int H, W, C, Ho, Wo, k, stride, pad, act;   // act: 0 none, 2 relu (applied after the optional per-channel affine)
  int src_u8;                                 // source image is uint8 (raw BGR crops), converted on the fly; scalar-gather path only
  FastDiv dWo, dHo, dC, dk;                   // multiply-shift reciprocals of Wo, Ho, C, k (conv_desc_of)
};

inline ConvDesc conv_desc_of(int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int act, int src_u8) {
  return ConvDesc{H, W, C, Ho, Wo, k, stride, pad, act, src_u8, fast_div_of(Wo), fast_div_of(Ho), fast_div_of(C), fast_div_of(k)};
}

__device__ __forceinline__ float conv_act(float v, int act) { return act == 2 ? fmaxf(v, 0.f) : v; }

// value of the virtual im2col matrix at (pixel row m, column kk) for 4 consecutive kk (kk % 4 == 0)
__device__ __forceinline__ float4 im2col_gather4(const float* __restrict__ x, const ConvDesc& cd, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, int m, int kk) {
  const int t = cd.dWo.div(m), ow = m - t * cd.Wo;
  const int n = cd.dHo.div(t), oh = t - n * cd.Ho;
  const int kt = cd.k * cd.k * cd.C;
  float out[4];
  if ((cd.C & 3) == 0) {
    const int tap = cd.dC.div(kk), ci = kk - tap * cd.C;
    const int kh = cd.dk.div(tap), kw = tap - kh * cd.k;
    const int ih = oh * cd.stride + kh - cd.pad, iw = ow * cd.stride + kw - cd.pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kk < kt && ih >= 0 && ih < cd.H && iw >= 0 && iw < cd.W) {
      v = *reinterpret_cast<const float4*>(x + (((int64_t)n * cd.H + ih) * cd.W + iw) * cd.C + ci);
      if (scale) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + ci), sh = *reinterpret_cast<const float4*>(shift + ci);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
      v.x = conv_act(v.x, cd.act); v.y = conv_act(v.y, cd.act); v.z = conv_act(v.z, cd.act); v.w = conv_act(v.w, cd.act);
    }
    return v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k1 = kk + e;
    float v = 0.f;
    if (k1 < kt) {
      const int tap = cd.dC.div(k1), ci = k1 - tap * cd.C;
      const int kh = cd.dk.div(tap), kw = tap - kh * cd.k;
      const int ih = oh * cd.stride + kh - cd.pad, iw = ow * cd.stride + kw - cd.pad;
      if (ih >= 0 && ih < cd.H && iw >= 0 && iw < cd.W) {
        const int64_t off = (((int64_t)n * cd.H + ih) * cd.W + iw) * cd.C + ci;
        v = cd.src_u8 ? (float)reinterpret_cast<const uint8_t*>(x)[off] : x[off];
        if (scale) v = fmaf(v, scale[ci], shift[ci]);
        v = conv_act(v, cd.act);
      }
    }
    out[e] = v;
  }
  return make_float4(out[0], out[1], out[2], out[3]);
}

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  RowMap a_map;        // applied to A's row index (KCONTIG: m ; KMAJOR: k)
  RowMap b_map;        // applied to B's row index when B is KMAJOR (k)
  RowMap c_map;        // applied to C's / R's row index
  const float* bias;   // [N] or null
  const float* R; int64_t ldr;          // residual
  // prologue
  const float* scale; const float* shift; const float* gate; int hw;
  // epilogue extras
  float* C2; int64_t ldc2;              // GEGLU: u store (may be null) ; GEGLU_BWD: u (read)
  double* stats; int stats_slots;       // EPI_STATS: [slots][2][N] fp64
  int n_half;                           // GEGLU: N/2 of the weight (h width)
  int k_chunk;                          // split-K: contraction length per blockIdx.y (0 = whole K)
  const float* A2;                      // PRO_BN_BWD second source (same layout / lda / row map as A)
  const float* b_scale; const float* b_shift; const float* b_gate; int b_hw;   // B prologue
  ConvDesc conv;                        // PRO_IM2COL / BPRO_IM2COL geometry
  int group_n;                          // > 0: L2-blocked tile order with this many tile columns per group
  long long* trace;                     // tuning aid (mt_debug_gemm_trace): per block {t_start, t_prologue, t_loop, t_end, hw_id}
  float* col_sum;                       // GEGLU_BWD: optional column sums of the stored values (bias gradient), fp32 atomics
  const void* b_planes; int64_t b_pstride;   // split loop: B as three pre-split bf16 planes [N][K] (plane stride in elements), or null
  const void* a_planes; int64_t a_pstride;   // plane loop (gemm_planes.hpp): A as three bf16 planes with A's shape and leading dimension
  void* c_planes; int64_t c_pstride; int64_t ldcp;   // plane loop: optional bf16 planes of the stored result (row-major, leading dimension ldcp)
  // EPI_SE_RED / EPI_ACT_BWD (epilogue-side vectors; the prologue's scale/shift/gate are taken by PRO_BN_BWD)
  const float* e_scale; const float* e_shift; const float* e_gate; const float* e_dpool; const float* e_mi; int e_hw;
  // stream-K (gemm_planes.hpp): persistent grid, per-block partial-tile slabs [grid][BM*BN] fp32 + one flag word per block
  float* sk_ws; int* sk_flags; int sk_on;
  int wave_prio;                        // plane loop: s_setprio level of every wave (0-3): main-queue GEMMs above the weight-gradient stream's
  DetLog det;                           // deterministic mode: log of the epilogue's column sums (col_sum)
  int64_t det_slab;                     // deterministic mode: C is a [splits][M][N] workspace, this is M * N (0 = atomics)
  int xcd_k;                            // split-K weight gradients: every XCD owns whole K-ranges (gemm_split.hpp), grid = (tiles, splits % 8 == 0)
};

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   /* v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the swish kernels are VALU-bound */
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float dswish_gemm_(float u) { const float s = sigmoidf_(u); return s * (1.0f + u * (1.0f - s)); }
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 rounding level) -- libm's erff costs ~3x the VALU slots,
// and the GEGLU epilogues evaluate it 64x per lane while no MFMA of that wave is in flight.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));     // v_rcp_f32 (1 ulp; the series itself is good to 1.5e-7)
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  const float r = 1.0f - y * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// gelu(x) and its derivative together (the GEGLU-backward epilogues need both for every element): ONE exponential -- erf's
// exp(-(x/sqrt 2)^2) is the Gaussian of the density term -- and one polynomial instead of two of each.
__device__ __forceinline__ void gelu_erf_both(float x, float& gelu, float& grad) {
  const float z = x * 0.70710678118654752440f;
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  const float e = __expf(-az * az);                  // = exp(-x^2 / 2)
  const float erfv = copysignf(1.0f - y * t * e, z);
  gelu = 0.5f * x * (1.0f + erfv);                   // (the same expression as gelu_erf: bit-identical values)
  grad = fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.0f + erfv));
}

// ---- XCD-aware tile order.  Block b runs on XCD b % 8 (observed dispatch; used for speed only, never for correctness).
// Returns false for a padding block of the L2-blocked order.
template <int BM, int BN>
__device__ __forceinline__ bool tile_coords(const GemmArgs& p, int& mt_, int& nt_) {
  const int n_tiles = (p.N + BN - 1) / BN;     // for GEGLU p.N is the full GEMM width (2*n_half)
  const int m_tiles = (p.M + BM - 1) / BM;
  if (p.group_n > 0) {
    // L2-blocked order for tall problems: every XCD owns a contiguous band of tile ROWS and sweeps it one group of
    // `group_n` tile columns at a time, so the group's B panels (group_n x BN x K floats, sized to ~2 MB) stay in that XCD's
    // 4 MB L2 while the A row-panels stream past once per group.  Without it an 8 MB weight matrix is re-fetched from
    // the Infinity Cache for every 128-row panel (FETCH_SIZE 20x the algorithmic read on the FF1 GEMM).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = m_tiles >> 3, r = m_tiles & 7;
    const int row0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int rows = q + (xcd < r ? 1 : 0);
    if (rows == 0) return false;
    const int per_group = rows * p.group_n;
    const int groups = (n_tiles + p.group_n - 1) / p.group_n;
    int g = idx / per_group;
    if (g > groups - 1) g = groups - 1;
    const int rem = idx - g * per_group;
    const int gw = (g == groups - 1) ? n_tiles - g * p.group_n : p.group_n;
    const int ml = rem / gw;
    if (ml >= rows) return false;               // padding block (bands differ by one row)
    mt_ = row0 + ml;
    nt_ = g * p.group_n + (rem - ml * gw);
  } else {
    // default: consecutive logical tiles (sharing an A row-panel) land on one XCD's L2
    const int nblk = n_tiles * m_tiles;
    int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    mt_ = bid / n_tiles;
    nt_ = bid - mt_ * n_tiles;
  }
  return true;
}

// Accumulator store with the fused epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// FAST = the wavefront's whole tile lies inside the matrix and rows are not re-mapped: no per-element bounds test or row map,
// one base pointer per lane and element offsets that are wave-uniform multiples of the leading dimension (the checked form costs
// ~60 ISA instructions per stored element -- a tenth of a K = 512 problem's run time).
template <int TM, int TN, int EPI, bool FAST>
__device__ __forceinline__ void gemm_epilogue_body(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int split) {
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  const int mw = m0 + wm * TM * 32;               // first row of this wavefront's tile

  if constexpr (EPI == EPI_GEGLU) {
    // acc[i][0] = 'a' pre-activation, acc[i][1] = gate pre-activation, same (row, col) in one lane
    const int j = (n0 >> 1) + wn * 32 + col_l;
    const bool jok = FAST || j < p.n_half;
    const float ba = (jok && p.bias) ? p.bias[j] : 0.f;
    const float bg = (jok && p.bias) ? p.bias[p.n_half + j] : 0.f;
    float* hp = p.C + (int64_t)(mw + row_h) * p.ldc + j;
    float* up = p.C2 ? p.C2 + (int64_t)(mw + row_h) * p.ldc2 + 2 * j : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
        if (FAST || (mw + row_h + dr < p.M && jok)) {
          const float a = acc[i][0][r] + ba;
          const float g = acc[i][1][r] + bg;
          hp[(int64_t)dr * p.ldc] = a * gelu_erf(g);
          // pre-activations interleaved (a_j, g_j): one 8-byte store per lane here, one 8-byte load in GEGLU_BWD
          if (up) *reinterpret_cast<float2*>(up + (int64_t)dr * p.ldc2) = make_float2(a, g);
        }
      }
    }
    return;
  } else if constexpr (EPI == EPI_SE_RED || EPI == EPI_ACT_BWD) {
    // A lane owns one column and 16*TM rows in increasing order, so the image index m / e_hw only ever steps forward: it is
    // tracked incrementally (one division per lane) and per-image state (running d-gate sum / gate and pooled-gradient values)
    // is flushed or reloaded when it changes.
    const float inv_hw = 1.0f / (float)p.e_hw;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + col_l;
      const bool nok = FAST || n < p.N;
      const int nn = nok ? n : 0;
      const float esc = p.e_scale[nn], esh = p.e_shift[nn];
      float mean = 0.f, istd = 0.f;
      if constexpr (EPI == EPI_ACT_BWD) { mean = p.e_mi[nn]; istd = p.e_mi[p.N + nn]; }
      const int mfirst = mw + row_h;
      int img = mfirst / p.e_hw;
      int rem = mfirst - img * p.e_hw;
      float run = 0.f, s1 = 0.f, s2 = 0.f, g = 0.f, dp = 0.f;
      if constexpr (EPI == EPI_ACT_BWD) {
        const int64_t gi = (int64_t)min(img, (p.M - 1) / p.e_hw) * p.N + nn;
        g = p.e_gate[gi]; dp = p.e_dpool[gi] * inv_hw;
      }
      const float* zp = p.C2 + (int64_t)mfirst * p.ldc2 + nn;
      float* cp = p.C + (int64_t)mfirst * p.ldc + nn;
      int prev = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
          rem += dr - prev;
          prev = dr;
          if (rem >= p.e_hw) {                                   // next image (rows advance by at most 8 < e_hw ... or more: loop)
            if constexpr (EPI == EPI_SE_RED) {
              if (nok && run != 0.f) atomicAdd(p.C + (int64_t)img * p.ldc + n, run);
              run = 0.f;
            }
            do { rem -= p.e_hw; ++img; } while (rem >= p.e_hw);
            if constexpr (EPI == EPI_ACT_BWD) {
              const int64_t gi = (int64_t)min(img, (p.M - 1) / p.e_hw) * p.N + nn;
              g = p.e_gate[gi]; dp = p.e_dpool[gi] * inv_hw;
            }
          }
          if (FAST || (mfirst + dr < p.M && nok)) {
            const float z = zp[(int64_t)dr * p.ldc2];
            const float u = fmaf(z, esc, esh);
            if constexpr (EPI == EPI_SE_RED) {
              run = fmaf(acc[i][j][r], swishf_(u), run);
            } else {
              const float d = fmaf(acc[i][j][r], g, dp) * dswish_gemm_(u);
              cp[(int64_t)dr * p.ldc] = d;
              s1 += d;
              s2 = fmaf(d, (z - mean) * istd, s2);
            }
          }
        }
      }
      if constexpr (EPI == EPI_SE_RED) {
        if (nok && run != 0.f) atomicAdd(p.C + (int64_t)img * p.ldc + n, run);
      } else {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lane < 32 && nok) {
          double* st = p.stats + (int64_t)(blockIdx.x % stat_slots(p.stats_slots)) * 2 * p.N;
          stat_add(st + n, stat_limb(p.stats_slots, p.N), s1);
          stat_add(st + p.N + n, stat_limb(p.stats_slots, p.N), s2);
        }
      }
    }
    return;
  } else {
    // Epilogues that READ global memory per element (residual, accumulate, the GEGLU pre-activations) run in two phases: every load
    // of the wavefront's tiles first, folded into the accumulators, then every store.  Element by element the compiler has to keep
    // "load, store, load, store" in program order (R / C2 may alias C) and puts s_waitcnt vmcnt(0) in front of every use -- and on
    // gfx9 that counter also holds the PREVIOUS element's store until it is acknowledged: 64 exposed round trips per wavefront
    // (the residual add of a K = 512 out-projection cost more than its main loop).
    // (GEGLU_BWD stays element-wise here: its second gradient would need 16 x TM x TN more registers, which the 168-register variants
    //  of the in-kernel-split loop do not have -- 321 -> 575 us when tried; the plane loop's own epilogue (gemm_planes.hpp) is two-phase)
    constexpr bool GATHER = EPI == EPI_BIAS_RES || EPI == EPI_ACCUM;
    if constexpr (GATHER) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + col_l;
        const bool nok = FAST || n < p.N;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
            const int m = mw + row_h + dr;
            if (FAST || (m < p.M && nok)) {
              const int64_t crow = FAST ? (int64_t)m : map_row(p.c_map, m);
              if constexpr (EPI == EPI_BIAS_RES) acc[i][j][r] += bias + p.R[crow * p.ldr + n];
              else acc[i][j][r] += bias + p.C[crow * p.ldc + n];
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + col_l;
      const bool nok = FAST || n < p.N;
      float bias = 0.f;
      if constexpr (EPI == EPI_STORE)
        bias = (nok && p.bias) ? p.bias[n] : 0.f;
      if constexpr (EPI == EPI_ATOMIC)      // split-K: the bias rides on the first K-slice only
        bias = (nok && p.bias && split == 0) ? p.bias[n] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      float* cp = p.C + (int64_t)(mw + row_h) * p.ldc + n;                         // FAST path base
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
          const int m = mw + row_h + dr;
          if (FAST || (m < p.M && nok)) {
            float v = acc[i][j][r];
            float* c;
            if constexpr (FAST) c = cp + (int64_t)dr * p.ldc;
            else c = p.C + map_row(p.c_map, m) * p.ldc + n;
            if constexpr (EPI == EPI_STORE) {
              *c = v + bias;
            } else if constexpr (EPI == EPI_BIAS_RES || EPI == EPI_ACCUM) {
              *c = v;                               // (bias and the residual / previous value were folded in by the gather phase)
            } else if constexpr (EPI == EPI_STATS) {
              *c = v;
              s1 += v; s2 += v * v;
            } else if constexpr (EPI == EPI_ATOMIC) {
              if (p.det_slab) c[(int64_t)split * p.det_slab] = v + bias;     // deterministic mode: partial tile into its split's slab (det.hpp)
              else atomicAdd(c, v + bias);
            } else if constexpr (EPI == EPI_GEGLU_BWD) {
              // n indexes h columns [0, n_half); u = [a | g] pre-activations
              const float2 ag = *reinterpret_cast<const float2*>(p.C2 + (int64_t)m * p.ldc2 + 2 * n);
              float gl, gr;
              gelu_erf_both(ag.y, gl, gr);
              const float da = v * gl, dg = v * ag.x * gr;
              c[0] = da;
              c[p.n_half] = dg;
              s1 += da; s2 += dg;
            }
          }
        }
      }
      if constexpr (EPI == EPI_GEGLU_BWD) {
        if (p.col_sum) {                      // bias gradient of the first feed-forward Linear: column sums of du
          s1 += __shfl_xor(s1, 32);
          s2 += __shfl_xor(s2, 32);
          if (lane < 32 && nok) {
            if (p.det.vals) {                 // deterministic mode: group = 32-column block, rank = 32-row block of the wave's first row (det.hpp)
              const int rk = mw >> 5;
              det_put(p.det, n >> 5, rk, n & 31, s1);
              det_put(p.det, (p.n_half + n) >> 5, rk, n & 31, s2);
              if (rk == 0 && (n & 31) == 0) { det_base(p.det, n >> 5, n); det_base(p.det, (p.n_half + n) >> 5, p.n_half + n); }
            } else {
              atomicAdd(p.col_sum + n, s1);
              atomicAdd(p.col_sum + p.n_half + n, s2);
            }
          }
        }
      }
      if constexpr (EPI == EPI_STATS) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lane < 32 && nok) {
          double* st = p.stats + (int64_t)(blockIdx.x % stat_slots(p.stats_slots)) * 2 * p.N;
          stat_add(st + n, stat_limb(p.stats_slots, p.N), s1);
          stat_add(st + p.N + n, stat_limb(p.stats_slots, p.N), s2);
        }
      }
    }
  }
}

template <int TM, int TN, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int split = -1) {
  if (split < 0) split = blockIdx.y;                 // split-K index (the XCD-major weight-gradient order passes its own)
  // wave-uniform: is this wavefront's TM*32 x TN*32 tile entirely inside the output, with identity row order?
  const int mw = m0 + wm * TM * 32;
  bool interior = p.c_map.gin == 0 && mw + TM * 32 <= p.M;
  if constexpr (EPI == EPI_GEGLU) interior = interior && (n0 >> 1) + wn * 32 + 32 <= p.n_half;
  else interior = interior && n0 + wn * TN * 32 + TN * 32 <= p.N;
  if (interior) gemm_epilogue_body<TM, TN, EPI, true>(p, acc, m0, n0, wm, wn, lane, split);
  else gemm_epilogue_body<TM, TN, EPI, false>(p, acc, m0, n0, wm, wn, lane, split);
}

// WAVES_K = 2 (split-K weight gradients with the atomic epilogue only): two groups of WAVES_M x WAVES_N wavefronts share the staged
// tiles, group wk takes MFMA slice group wk of every k-step and adds its own partial tile in the epilogue.  For the one-tile
// 64 x 288 weight gradient this makes the block 12 wavefronts = 3 per SIMD: a 6-wavefront block sits (2, 2, 1, 1) on the four SIMDs
// and a second one does not fit beside it at 3 wavefronts per SIMD (measured: one block per CU, run time linear in the block count
// from 256 blocks on), so half of the matrix pipes ran at half load.
template <int WAVES_M, int WAVES_N, int TM, int TN, int AL, int BL, int PRO, int EPI, int BPRO = BPRO_NONE, int WAVES_K = 1>
__global__ __launch_bounds__(WAVES_M * WAVES_N * WAVES_K * 64, MT_MIN_WAVES)
void gemm_kernel(const GemmArgs p) {
  static_assert(WAVES_K == 1 || (WAVES_K == 2 && EPI == EPI_ATOMIC && MT_BK == 16), "K groups: one per MFMA slice group, partial tiles added atomically");
  constexpr int NT = WAVES_M * WAVES_N * WAVES_K * 64;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int BK = MT_BK;
  constexpr int KQ = BK / 4;                    // float4 units per k-contiguous row
  // LDS tile per operand follows its GLOBAL layout so the global->LDS path is a straight float4 copy:
  //   k-contiguous operand -> [rows][BK+4]  (fragment = one ds_read_b128: 4 consecutive k for 4 MFMA steps; the +4 pad makes
  //                                           the 16-lane b128 service groups hit 16 distinct 16-byte slots)
  //   k-major operand      -> [BK][rows+4]  (fragment = ds_read_b32 of 32 consecutive rows, conflict-free)
  // Both use the same k order inside a tile: MFMA step (g,t), lane half kh consumes k = 8g + 4kh + t.
  // LDS tiles are k-major [BK][rows + 4]: the MFMA fragment read is a conflict-free ds_read_b32 in exactly the operand layout
  // (a row-major [rows][BK + 4] image read with ds_read_b128 measured the same, so the simpler one stayed)
  constexpr int LDA_S = BM + 4;
  constexpr int LDB_S = BN + 4;
  constexpr int A_TILE = BK * LDA_S;
  constexpr int B_TILE = BK * LDB_S;
  constexpr int A_UNITS = (BM * BK / 4 + NT - 1) / NT;   // float4 units per thread
  constexpr int B_UNITS = (BN * BK / 4 + NT - 1) / NT;
  constexpr bool A_EXACT = (BM * BK / 4) % NT == 0;
  constexpr bool B_EXACT = (BN * BK / 4) % NT == 0;

  __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
  float* As = smem;                       // [2][A_TILE]
  float* Bs = smem + 2 * A_TILE;          // [2][B_TILE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WAVES_M * WAVES_N);          // K group (0 unless WAVES_K == 2)
  const int wm = (wave % (WAVES_M * WAVES_N)) / WAVES_N;
  const int wn = wave % WAVES_N;

  int mt_, nt_;
  if (!tile_coords<BM, BN>(p, mt_, nt_)) return;
  const int m0 = mt_ * BM;
  const int n0 = nt_ * BN;

  int k_begin = 0, k_end = p.K;
  if (p.k_chunk > 0) {
    k_begin = blockIdx.y * p.k_chunk;
    k_end = min(p.K, k_begin + p.k_chunk);
    if (k_begin >= k_end) return;
  }
  const int nk = (k_end - k_begin + BK - 1) / BK;

  // B-tile row -> global weight row (GEGLU interleaves the 'a' and 'gate' halves so one lane holds both)
  auto b_row = [&](int r) -> int {
    if constexpr (EPI == EPI_GEGLU) {
      static_assert(EPI != EPI_GEGLU || TN == 2, "GEGLU wants TN == 2");
      const int w = r / 64, sel = (r >> 5) & 1, c = r & 31;
      const int j = (n0 >> 1) + w * 32 + c;
      return j < p.n_half ? sel * p.n_half + j : -1;
    } else {
      const int n = n0 + r;
      return n < p.N ? n : -1;
    }
  };

  // ---- im2col prologues, granule form (a float NHWC image with C % 4 == 0: every user but the 3-channel stem, which keeps
  // im2col_gather4).  The generic gather decomposed (row -> image, oh, ow) and (column -> tap, channel) and tested the window for
  // every 16-byte load: ~70 VALU instructions per load, 300 per k-step next to 16 MFMAs (Xception's conv2 ran VALU-bound on the
  // matrix-core kernel).  Here everything that does not change from one k-step to the next is kept per staging unit:
  //   A side (rows = output pixels, fixed per unit): the window origin's element offset and a bit mask of its in-image taps; a k-step
  //     costs one tap decomposition per THREAD (all units of a thread share the k granule) and a bit test + an add per unit;
  //   B side (rows = the contraction = pixels, 16 further per k-step; columns = (tap, channel), fixed per unit): the column's element
  //     offset, and the pixel's offset / (oh, ow) advanced incrementally.
  constexpr bool A_IM = PRO == PRO_IM2COL, B_IM = BPRO == BPRO_IM2COL;
  constexpr int A_ST = A_IM ? A_UNITS : 1, B_ST = B_IM ? B_UNITS : 1;
  const ConvDesc& cd = p.conv;
  const bool im_relu = cd.act == 2;
  int64_t a_pix[A_ST]; uint32_t a_msk[A_ST];
  int64_t b_pix[B_ST]; int b_col[B_ST], b_oh[B_ST], b_ow[B_ST], b_tap[B_ST];
  float4 b_sc[B_ST], b_sh[B_ST];                              // the column's affine (identity without b_scale): loop-invariant per unit
  constexpr bool A_KB = B_IM && PRO == PRO_BN_BWD && AL == LAYOUT_KMAJOR && BM <= 64;   // the one-tile weight gradient: ka / kb / kc of the unit's 4 channels kept too
  float4 a_ka[A_KB ? A_UNITS : 1], a_kb[A_KB ? A_UNITS : 1], a_kc[A_KB ? A_UNITS : 1];
  if constexpr (A_IM) {
    static_assert(!A_IM || (AL == LAYOUT_KCONTIG && NT % KQ == 0), "im2col A: k-contiguous rows, one k granule per thread");
    {
#pragma unroll
      for (int i = 0; i < A_UNITS; ++i) {
        const int u = tid + i * NT, m = m0 + u / KQ;
        a_pix[i] = 0; a_msk[i] = 0;
        if ((A_EXACT || u < BM * BK / 4) && m < p.M) {
          const int t = cd.dWo.div(m), ow = m - t * cd.Wo;
          const int n = cd.dHo.div(t), oh = t - n * cd.Ho;
          const int ih0 = oh * cd.stride - cd.pad, iw0 = ow * cd.stride - cd.pad;
          a_pix[i] = (((int64_t)n * cd.H + ih0) * cd.W + iw0) * cd.C;
          uint32_t msk = 0;
          for (int kh = 0, t2 = 0; kh < cd.k; ++kh)
            for (int kw = 0; kw < cd.k; ++kw, ++t2)
              if ((unsigned)(ih0 + kh) < (unsigned)cd.H && (unsigned)(iw0 + kw) < (unsigned)cd.W) msk |= 1u << t2;
          a_msk[i] = msk;
        }
      }
    }
  }
  if constexpr (B_IM) {
    static_assert(!B_IM || BL == LAYOUT_KMAJOR, "im2col B: k-major (weight gradients)");
    {
      constexpr int QPR = BN / 4;
#pragma unroll
      for (int i = 0; i < B_UNITS; ++i) {
        const int u = tid + i * NT, kk = u / QPR, n = n0 + (u - kk * QPR) * 4;
        const int tap = cd.dC.div(n), ci = n - tap * cd.C;
        const int kh = cd.dk.div(tap), kw = tap - kh * cd.k;
        b_col[i] = (kh * cd.W + kw) * cd.C + ci;
        b_sc[i] = make_float4(1.f, 1.f, 1.f, 1.f); b_sh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.b_scale && n < p.N && tap < cd.k * cd.k) {
          b_sc[i] = *reinterpret_cast<const float4*>(p.b_scale + ci); b_sh[i] = *reinterpret_cast<const float4*>(p.b_shift + ci);
        }
        b_tap[i] = ((B_EXACT || u < BN * BK / 4) && n < p.N && tap < cd.k * cd.k) ? (kh | (kw << 8)) : -1;
        const int k = k_begin + kk;                       // this unit's pixel in the first k-step; later ones are BK further
        const int t = cd.dWo.div(k), ow = k - t * cd.Wo;
        const int img = cd.dHo.div(t), oh = t - img * cd.Ho;
        b_oh[i] = oh; b_ow[i] = ow;
        b_pix[i] = (((int64_t)img * cd.H + (oh * cd.stride - cd.pad)) * cd.W + (ow * cd.stride - cd.pad)) * cd.C;
      }
    }
  }

  if constexpr (A_KB) {
    constexpr int QPRA = BM / 4;
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int u = tid + i * NT, m = m0 + (u % QPRA) * 4;
      a_ka[i] = a_kb[i] = a_kc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((A_EXACT || u < BM * BK / 4) && m < p.M) {
        a_ka[i] = *reinterpret_cast<const float4*>(p.scale + m); a_kb[i] = *reinterpret_cast<const float4*>(p.shift + m);
        a_kc[i] = *reinterpret_cast<const float4*>(p.gate + m);
      }
    }
  }
  float4 ra[A_UNITS], rb[B_UNITS];

  auto load_tiles = [&](int kt) {
    const int k0 = k_begin + kt * BK;
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int u = tid + i * NT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_EXACT || u < BM * BK / 4) {
        if constexpr (AL == LAYOUT_KCONTIG) {
          const int row = u / KQ, kq = u % KQ;
          const int m = m0 + row, k = k0 + kq * 4;
          if constexpr (PRO == PRO_IM2COL) {
            {
              // (tap, channel) of this thread's k granule: the same for all its units, the compiler keeps one copy
              const int tap = cd.dC.div(k), ci = k - tap * cd.C;
              const int kh = cd.dk.div(tap), kw = tap - kh * cd.k;
              if (k < k_end && tap < cd.k * cd.k && ((a_msk[i] >> tap) & 1u)) {
                v = *reinterpret_cast<const float4*>(p.A + (a_pix[i] + ((kh * cd.W + kw) * cd.C + ci)));
                if (p.scale) {
                  const float4 sc = *reinterpret_cast<const float4*>(p.scale + ci), sh = *reinterpret_cast<const float4*>(p.shift + ci);
                  v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                }
                if (im_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              }
            }
          } else if constexpr (PRO == PRO_IM2COL_ANY) {
            if (m < p.M && k < k_end) v = im2col_gather4(p.A, p.conv, p.scale, p.shift, m, k);
          } else if (m < p.M && k < k_end) {
            v = *reinterpret_cast<const float4*>(p.A + map_row(p.a_map, m) * p.lda + k);
            if constexpr (PRO == PRO_BN_BWD) {
              const float4 z2 = *reinterpret_cast<const float4*>(p.A2 + map_row(p.a_map, m) * p.lda + k);
              const float4 ka = *reinterpret_cast<const float4*>(p.scale + k);
              const float4 kb = *reinterpret_cast<const float4*>(p.shift + k);
              const float4 kc = *reinterpret_cast<const float4*>(p.gate + k);
              v.x = fmaf(ka.x, v.x, fmaf(kb.x, z2.x, kc.x)); v.y = fmaf(ka.y, v.y, fmaf(kb.y, z2.y, kc.y));
              v.z = fmaf(ka.z, v.z, fmaf(kb.z, z2.z, kc.z)); v.w = fmaf(ka.w, v.w, fmaf(kb.w, z2.w, kc.w));
            }
            if constexpr (PRO == PRO_BN_SWISH_GATE || PRO == PRO_BN_SWISH || PRO == PRO_AFFINE) {
              const float4 sc = *reinterpret_cast<const float4*>(p.scale + k);
              const float4 sh = *reinterpret_cast<const float4*>(p.shift + k);
              v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
              if constexpr (PRO != PRO_AFFINE) {
                v.x = swishf_(v.x); v.y = swishf_(v.y); v.z = swishf_(v.z); v.w = swishf_(v.w);
              }
              if constexpr (PRO == PRO_BN_SWISH_GATE) {
                const float4 g = *reinterpret_cast<const float4*>(p.gate + (int64_t)(m / p.hw) * p.K + k);
                v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
              }
            }
          }
        } else {
          constexpr int QPR = BM / 4;
          const int kk = u / QPR, mq = u - kk * QPR;
          const int k = k0 + kk, m = m0 + mq * 4;
          if (k < k_end && m < p.M) {
            const int64_t off = map_row(p.a_map, k) * p.lda + m;
            v = *reinterpret_cast<const float4*>(p.A + off);
            if constexpr (PRO == PRO_BN_BWD) {      // channel index is the output row m here
              const float4 z2 = *reinterpret_cast<const float4*>(p.A2 + off);
              float4 ka, kb, kc;
              if constexpr (A_KB) { ka = a_ka[i]; kb = a_kb[i]; kc = a_kc[i]; }
              else {
                ka = *reinterpret_cast<const float4*>(p.scale + m);
                kb = *reinterpret_cast<const float4*>(p.shift + m);
                kc = *reinterpret_cast<const float4*>(p.gate + m);
              }
              v.x = fmaf(ka.x, v.x, fmaf(kb.x, z2.x, kc.x)); v.y = fmaf(ka.y, v.y, fmaf(kb.y, z2.y, kc.y));
              v.z = fmaf(ka.z, v.z, fmaf(kb.z, z2.z, kc.z)); v.w = fmaf(ka.w, v.w, fmaf(kb.w, z2.w, kc.w));
            }
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
      const int u = tid + i * NT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_EXACT || u < BN * BK / 4) {
        if constexpr (BL == LAYOUT_KCONTIG) {
          const int row = u / KQ, kq = u % KQ;
          const int n = b_row(row), k = k0 + kq * 4;
          if (n >= 0 && k < k_end) v = *reinterpret_cast<const float4*>(p.B + (int64_t)n * p.ldb + k);
        } else {
          constexpr int QPR = BN / 4;
          const int kk = u / QPR, nq = u - kk * QPR;
          const int k = k0 + kk, n = n0 + nq * 4;
          if constexpr (BPRO == BPRO_IM2COL) {
            {
              const int kh = b_tap[i] & 255, kw = b_tap[i] >> 8;
              const int ih = b_oh[i] * cd.stride - cd.pad + kh, iw = b_ow[i] * cd.stride - cd.pad + kw;
              if (k < k_end && b_tap[i] >= 0 && (unsigned)ih < (unsigned)cd.H && (unsigned)iw < (unsigned)cd.W) {
                v = *reinterpret_cast<const float4*>(p.B + (b_pix[i] + b_col[i]));
                if (p.b_scale) {
                  v.x = fmaf(v.x, b_sc[i].x, b_sh[i].x); v.y = fmaf(v.y, b_sc[i].y, b_sh[i].y);
                  v.z = fmaf(v.z, b_sc[i].z, b_sh[i].z); v.w = fmaf(v.w, b_sc[i].w, b_sh[i].w);
                }
                if (im_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              }
              // next k-step: BK pixels further (load_tiles is called once per k-step, in order)
              b_ow[i] += BK; b_pix[i] += (int64_t)BK * cd.stride * cd.C;
              while (b_ow[i] >= cd.Wo) {
                b_ow[i] -= cd.Wo; b_oh[i] += 1; b_pix[i] += ((int64_t)cd.W - cd.Wo) * cd.stride * cd.C;
                if (b_oh[i] >= cd.Ho) { b_oh[i] = 0; b_pix[i] += ((int64_t)cd.H - (int64_t)cd.Ho * cd.stride) * cd.W * cd.C; }
              }
            }
          } else if constexpr (BPRO == BPRO_IM2COL_ANY) {
            if (k < k_end && n < p.N) v = im2col_gather4(p.B, p.conv, p.b_scale, p.b_shift, k, n);
          } else if (k < k_end && n < p.N) {
            v = *reinterpret_cast<const float4*>(p.B + map_row(p.b_map, k) * p.ldb + n);
            if constexpr (BPRO == BPRO_BN_SWISH_GATE) {
              const float4 sc = *reinterpret_cast<const float4*>(p.b_scale + n);
              const float4 sh = *reinterpret_cast<const float4*>(p.b_shift + n);
              const float4 g = *reinterpret_cast<const float4*>(p.b_gate + (int64_t)(k / p.b_hw) * p.N + n);
              v.x = swishf_(fmaf(v.x, sc.x, sh.x)) * g.x; v.y = swishf_(fmaf(v.y, sc.y, sh.y)) * g.y;
              v.z = swishf_(fmaf(v.z, sc.z, sh.z)) * g.z; v.w = swishf_(fmaf(v.w, sc.w, sh.w)) * g.w;
            }
          }
        }
      }
      rb[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int i = 0; i < A_UNITS; ++i) {
      const int u = tid + i * NT;
      if (A_EXACT || u < BM * BK / 4) {
        if constexpr (AL == LAYOUT_KCONTIG) {
          const int row = u / KQ, kq = u % KQ;
          as[(kq * 4 + 0) * LDA_S + row] = ra[i].x; as[(kq * 4 + 1) * LDA_S + row] = ra[i].y;
          as[(kq * 4 + 2) * LDA_S + row] = ra[i].z; as[(kq * 4 + 3) * LDA_S + row] = ra[i].w;
        } else {
          constexpr int QPR = BM / 4;
          const int kk = u / QPR, mq = u - kk * QPR;
          *reinterpret_cast<float4*>(as + kk * LDA_S + mq * 4) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_UNITS; ++i) {
      const int u = tid + i * NT;
      if (B_EXACT || u < BN * BK / 4) {
        if constexpr (BL == LAYOUT_KCONTIG) {
          const int row = u / KQ, kq = u % KQ;
          bs[(kq * 4 + 0) * LDB_S + row] = rb[i].x; bs[(kq * 4 + 1) * LDB_S + row] = rb[i].y;
          bs[(kq * 4 + 2) * LDB_S + row] = rb[i].z; bs[(kq * 4 + 3) * LDB_S + row] = rb[i].w;
        } else {
          constexpr int QPR = BN / 4;
          const int kk = u / QPR, nq = u - kk * QPR;
          *reinterpret_cast<float4*>(bs + kk * LDB_S + nq * 4) = rb[i];
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  long long* tr = p.trace ? p.trace + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (tr && threadIdx.x == 0) { tr[0] = wall_clock64(); tr[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  if (tr && threadIdx.x == 0) tr[1] = wall_clock64();

  const int a_row = wm * TM * 32 + (lane & 31);
  const int b_frag = wn * TN * 32 + (lane & 31);
  const int khalf = lane >> 5;

  // Main loop, software-pipelined at the source level (hipcc keeps this order):
  //   * fragments for MFMA group g+1 are read from LDS BEFORE group g's MFMAs are issued -> ds_read latency sits under
  //     256+ cycles of matrix work instead of in front of it;
  //   * the next tile's global loads are issued at the top and their LDS stores placed MID-tile (other buffer), so the
  //     store phase is covered by this wave's remaining MFMAs rather than serialised in front of the barrier.
  auto load_frags = [&](const float* as, const float* bs, int g, float (&af)[TM][4], float (&bf)[TN][4]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) af[i][t] = as[(g * 8 + khalf * 4 + t) * LDA_S + a_row + i * 32];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[j][t] = bs[(g * 8 + khalf * 4 + t) * LDB_S + b_frag + j * 32];
    }
  };
  auto mma_steps = [&](const float (&af)[TM][4], const float (&bf)[TN][4], int t0, int t1) {
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
  };
  auto mma_group = [&](const float (&af)[TM][4], const float (&bf)[TN][4]) { mma_steps(af, bf, 0, 4); };
  constexpr int NG = BK / 8;
  static_assert(NG == 2 || NG == 4, "BK must be 16 or 32");

  if constexpr (WAVES_K == 2) {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const bool more = kt + 1 < nk;
      if (more) load_tiles(kt + 1);
      float fa[TM][4], fb[TN][4];
      load_frags(As + buf * A_TILE, Bs + buf * B_TILE, wk, fa, fb);
      mma_steps(fa, fb, 0, 2);
      if (more) store_tiles(buf ^ 1);
      mma_steps(fa, fb, 2, 4);
      __syncthreads();
    }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tiles(kt + 1);
    const float* as = As + buf * A_TILE;
    const float* bs = Bs + buf * B_TILE;
    float fa0[TM][4], fb0[TN][4], fa1[TM][4], fb1[TN][4];
    load_frags(as, bs, 0, fa0, fb0);
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
      load_frags(as, bs, g + 1, fa1, fb1);
      mma_group(fa0, fb0);
      if (g + 2 < NG) load_frags(as, bs, g + 2, fa0, fb0);
      if (g == NG - 2 && more) store_tiles(buf ^ 1);      // before the last MFMA group of the tile (later placements measured the same)
      mma_group(fa1, fb1);
    }
    __syncthreads();
  }

  if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
  gemm_epilogue<TM, TN, EPI>(p, acc, m0, n0, wm, wn, lane);
  if (tr && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); tr[3] = wall_clock64(); }
}

}  // namespace mt
