// Weight gradient of a 1x1 convolution with few channels and very many rows (EfficientNet-B0 stages 1-4 at 112^2..28^2:
// up to 3.2 M rows per batch of 256 crops against a 16x32 ... 240x40 weight; reference efficientnet_pytorch/model.py:93-104
// driven backwards by train.py:371).
//
//   dW[co, ci] += sum_r dz[r, co] * a[r, ci]
//   dz[r, co]   = ka[co]*du[r, co] + kb[co]*z[r, co] + kc[co]                      (BatchNorm backward folded into the load)
//   a[r, ci]    = x[r, ci]                                                          (expand conv: materialised block input)
//               | swish(sc[ci]*x[r, ci] + sh[ci]) * gate[r / hw, ci]                (project conv: BN + swish + SE gate on load)
//
// The generic TN GEMM tiles the OUTPUT (128x64 and up), which for a 96x16 result wastes most of every MFMA and leaves the
// launch at 5-12x its HBM time.  Here the output (padded to 32x32 MFMA tiles) lives entirely in each wavefront's accumulators
// and the ROWS are what gets distributed: a block streams 32- or 64-row chunks of du / z / x through registers into LDS (loads
// of chunk i+1 in flight while chunk i is multiplied), each of its 4 wavefronts multiplies a quarter of the chunk's rows into its
// own accumulators (v_mfma_f32_32x32x2_f32, k = rows), and the partial results meet in LDS and then in global fp32 atomics.
// HBM-bound by design: algorithmic bytes = rows * (2*Cout + Cin) * 4.
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
  const float* du; const float* z; const float* kabc;      // [rows, Cout] x2, [3, Cout]
  const float* x;                                            // [rows, Cin]
  const float* sc; const float* sh; const float* gate;       // GATE mode: [Cin], [Cin], [rows / hw, Cin]
  float* dw;                                                 // [Cout, Cin], accumulated with atomics
  int64_t rows; int Cout, Cin, hw;
  // STEM mode (3x3 stride-2 "same" conv on 3-channel crops, a = im2col(x) gathered on the fly; one chunk = one output row)
  const void* img; int x_u8, H, W, Ho, Wo, pad0;
};

enum { A_PLAIN = 0, A_GATE = 1, A_STEM = 2 };

__device__ __forceinline__ float swish_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }   /* v_rcp_f32 (1 ulp), see effnet_fwd.hip sigmoidf_ */
// component-wise on purpose: `c ? a : zero4` on the structs makes the compiler select between two stack slots
__device__ __forceinline__ float4 keep_if(bool c, const float4& a) { return make_float4(c ? a.x : 0.f, c ? a.y : 0.f, c ? a.z : 0.f, c ? a.w : 0.f); }

constexpr int ld_for(int tiles) { return tiles * 32 + ((tiles & 1) ? 0 : 32); }   // ld % 64 == 32: the two k-rows of a fragment hit disjoint banks

// MT x NT 32x32 output tiles, R rows per chunk, AMODE selects the `a` operand (plain / project-conv transform / stem im2col).  The block's 4 wavefronts are
// arranged as WM x WN groups over the output tiles times WR = 4 / (WM*WN) groups over the chunk's rows.
template <int MT, int NT, int R, int AMODE, int WM, int WN, bool KREG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1x1_wgrad_kernel(WgradArgs p) {
  constexpr int LDZ = ld_for(MT), LDA = ld_for(NT);
  constexpr int VZ = (R * MT * 32 / 4 + 255) / 256;      // float4 slots per thread covering [R, Cout]
  constexpr int VA = (R * NT * 32 / 4 + 255) / 256;
  constexpr int WT = WM * WN, WR = 4 / WT, MTW = MT / WM, NTW = NT / WN;
  static_assert(MT % WM == 0 && NT % WN == 0 && 4 % WT == 0 && R % (8 * WR) == 0, "wave arrangement");
  extern __shared__ float smem[];
  float* dzs = smem;                       // [R][LDZ]
  float* as = dzs + R * LDZ;               // [R][LDA]
  float* kab = as + R * LDA;               // [3][MT*32]  ka | kb | kc
  float* ssh = kab + 3 * MT * 32;          // [2][NT*32]  sc | sh   (GATE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = p.Cout >> 2, aq = p.Cin >> 2;

  for (int i = tid; i < R * (LDZ + LDA); i += 256) smem[i] = 0.f;   // padding columns stay zero for the whole kernel
  for (int i = tid; i < MT * 32; i += 256) {
    const bool ok = i < p.Cout;
    kab[i] = ok ? p.kabc[i] : 0.f;
    kab[MT * 32 + i] = ok ? p.kabc[p.Cout + i] : 0.f;
    kab[2 * MT * 32 + i] = ok ? p.kabc[2 * p.Cout + i] : 0.f;
  }
  constexpr bool GATE = AMODE == A_GATE, STEM = AMODE == A_STEM;
  constexpr int SV = STEM ? R / 8 : 1;       // STEM: scalar gather slots per thread (lane = tap, 8 pixels per pass)
  static_assert(!STEM || (MT == 1 && NT == 1 && WM * WN == 1), "stem mode is a 32 x 27 result");
  if constexpr (GATE)
    for (int i = tid; i < NT * 32; i += 256) {
      const bool ok = i < p.Cin;
      ssh[i] = ok ? p.sc[i] : 0.f;
      ssh[NT * 32 + i] = ok ? p.sh[i] : 0.f;
    }

  // loop-invariant placement of this thread's float4 slots inside a chunk: row (or -1) and column
  int zr[VZ], zc[VZ], ar[VA], ac[VA];
  float4 kar[KREG ? VZ : 1], kbr[KREG ? VZ : 1], kcr[KREG ? VZ : 1];
#pragma unroll
  for (int i = 0; i < VZ; ++i) {
    const int idx = tid + 256 * i;
    zr[i] = idx / cq;
    zc[i] = (idx - zr[i] * cq) * 4;
    if (zr[i] >= R) { zr[i] = -1; zc[i] = 0; }
    if constexpr (KREG) {
      kar[i] = *reinterpret_cast<const float4*>(p.kabc + zc[i]);
      kbr[i] = *reinterpret_cast<const float4*>(p.kabc + p.Cout + zc[i]);
      kcr[i] = *reinterpret_cast<const float4*>(p.kabc + 2 * p.Cout + zc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < VA; ++i) {
    const int idx = tid + 256 * i;
    ar[i] = idx / aq;
    ac[i] = (idx - ar[i] * aq) * 4;
    if (ar[i] >= R) { ar[i] = -1; ac[i] = 0; }
  }
  // STEM: this thread's tap (kh, kw, ci) is fixed; it gathers it for pixels (tid >> 5) + 8 i of the output row
  const int tap = tid & 31, s_kh = tap / 9, s_kw = (tap % 9) / 3, s_ci = tap % 3;
  float sx[SV];
  unsigned sx_ok = 0;

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int CR = STEM ? p.Wo : R;          // rows a chunk advances by
  const int64_t nchunks = (p.rows + CR - 1) / CR;
  float4 rdu[VZ], rz[VZ], rx[VA];

  // Loads are unconditional (row clamped into the chunk): a `cond ? *p : 0` form makes the compiler select between the global
  // address and a stack slot and emit flat loads, which serialises the whole prefetch.  Rows past the end are zeroed in stage().
  auto fetch = [&](int64_t chunk) {
    const int64_t r0 = chunk * CR;
    const int last = (int)((p.rows - r0) < CR ? (p.rows - r0) : CR) - 1;
    const float* du_c = p.du + r0 * p.Cout;
    const float* z_c = p.z + r0 * p.Cout;
#pragma unroll
    for (int i = 0; i < VZ; ++i) {
      const int off = min(max(zr[i], 0), last) * p.Cout + zc[i];
      rdu[i] = *reinterpret_cast<const float4*>(du_c + off);
      rz[i] = *reinterpret_cast<const float4*>(z_c + off);
    }
    if constexpr (STEM) {
      const int n = (int)(chunk / p.Ho), oh = (int)(chunk - (int64_t)n * p.Ho);
      const int ih = 2 * oh + s_kh - p.pad0;
      const bool row_ok = tap < 27 && ih >= 0 && ih < p.H;
      const int64_t base = ((int64_t)n * p.H + min(max(ih, 0), p.H - 1)) * p.W * 3 + s_ci;
      sx_ok = 0;
#pragma unroll
      for (int i = 0; i < SV; ++i) {
        const int ow = (tid >> 5) + 8 * i;
        const int iw = 2 * ow + s_kw - p.pad0;
        if (row_ok && ow < p.Wo && iw >= 0 && iw < p.W) sx_ok |= 1u << i;
        const int64_t off = base + (int64_t)min(max(iw, 0), p.W - 1) * 3;
        if (p.x_u8) sx[i] = (float)reinterpret_cast<const uint8_t*>(p.img)[off];
        else sx[i] = reinterpret_cast<const float*>(p.img)[off];
      }
    } else {
      const float* x_c = p.x + r0 * p.Cin;
#pragma unroll
      for (int i = 0; i < VA; ++i) rx[i] = *reinterpret_cast<const float4*>(x_c + min(max(ar[i], 0), last) * p.Cin + ac[i]);
    }
  };

  auto stage = [&](int64_t chunk) {   // registers -> LDS, operand transforms applied here
    const int64_t r0 = chunk * CR;
    const int left = (int)((p.rows - r0) < CR ? (p.rows - r0) : CR);
#pragma unroll
    for (int i = 0; i < VZ; ++i) {
      if (zr[i] >= 0) {
        float4 ka, kb, kc;
        if constexpr (KREG) {
          ka = kar[i]; kb = kbr[i]; kc = kcr[i];
        } else {
          ka = *reinterpret_cast<const float4*>(kab + zc[i]);
          kb = *reinterpret_cast<const float4*>(kab + MT * 32 + zc[i]);
          kc = *reinterpret_cast<const float4*>(kab + 2 * MT * 32 + zc[i]);
        }
        const bool ok = zr[i] < left;
        float4 v;
        v.x = ok ? fmaf(ka.x, rdu[i].x, fmaf(kb.x, rz[i].x, kc.x)) : 0.f;
        v.y = ok ? fmaf(ka.y, rdu[i].y, fmaf(kb.y, rz[i].y, kc.y)) : 0.f;
        v.z = ok ? fmaf(ka.z, rdu[i].z, fmaf(kb.z, rz[i].z, kc.z)) : 0.f;
        v.w = ok ? fmaf(ka.w, rdu[i].w, fmaf(kb.w, rz[i].w, kc.w)) : 0.f;
        *reinterpret_cast<float4*>(dzs + zr[i] * LDZ + zc[i]) = v;
      }
    }
    if constexpr (GATE) {
      const int64_t img0 = div_rows(r0, p.hw);                      // uniform: one division per chunk
      const int rem0 = (int)(r0 - img0 * p.hw);
#pragma unroll
      for (int i = 0; i < VA; ++i) {
        if (ar[i] >= 0) {
          float4 v = keep_if(ar[i] < left, rx[i]);
          if (ar[i] < left) {
            int t = rem0 + ar[i];
            int64_t img = img0;
            while (t >= p.hw) { t -= p.hw; ++img; }
            const float4 s = *reinterpret_cast<const float4*>(ssh + ac[i]);
            const float4 h = *reinterpret_cast<const float4*>(ssh + NT * 32 + ac[i]);
            const float4 g = *reinterpret_cast<const float4*>(p.gate + img * p.Cin + ac[i]);
            v.x = swish_f(fmaf(s.x, v.x, h.x)) * g.x;
            v.y = swish_f(fmaf(s.y, v.y, h.y)) * g.y;
            v.z = swish_f(fmaf(s.z, v.z, h.z)) * g.z;
            v.w = swish_f(fmaf(s.w, v.w, h.w)) * g.w;
          }
          *reinterpret_cast<float4*>(as + ar[i] * LDA + ac[i]) = v;
        }
      }
    } else if constexpr (STEM) {
#pragma unroll
      for (int i = 0; i < SV; ++i) as[((tid >> 5) + 8 * i) * LDA + tap] = ((sx_ok >> i) & 1u) ? sx[i] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < VA; ++i)
        if (ar[i] >= 0) *reinterpret_cast<float4*>(as + ar[i] * LDA + ac[i]) = keep_if(ar[i] < left, rx[i]);
    }
  };

  int64_t chunk = blockIdx.x;
  if (chunk < nchunks) fetch(chunk);
  __syncthreads();                        // zero fill and constants in place
  const int kh = lane >> 5, cl = lane & 31;
  const int tg = wave % WT, rg = wave / WT;
  const int tm = tg / WN, tn = tg % WN;
  const float* dz_w = dzs + tm * MTW * 32 + cl;
  const float* a_w = as + tn * NTW * 32 + cl;
  for (; chunk < nchunks; chunk += gridDim.x) {
    stage(chunk);
    __syncthreads();
    const int64_t nxt = chunk + gridDim.x;
    if (nxt < nchunks) fetch(nxt);        // in flight while this chunk is multiplied
#pragma unroll 2
    for (int ks = 0; ks < R / (2 * WR); ++ks) {
      const int r = rg * (R / WR) + ks * 2 + kh;
      float af[MTW], bf[NTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) af[i] = dz_w[r * LDZ + i * 32];
#pragma unroll
      for (int j = 0; j < NTW; ++j) bf[j] = a_w[r * LDA + j * 32];
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if constexpr (WR == 1) {
    // every wavefront owns its tiles: straight to the global accumulation
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (tm * MTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int n = (tn * NTW + j) * 32 + cl;
          if (m < p.Cout && n < p.Cin) atomicAdd(p.dw + m * p.Cin + n, acc[i][j][r]);
        }
  } else {
    // the row groups' partial results meet in LDS, then one global atomic per weight and block
    constexpr int LDR = NT * 32;
    float* red = smem;                       // [MT*32][LDR] (the host sizes smem for it)
    for (int i = tid; i < MT * 32 * LDR; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (tm * MTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          atomicAdd(red + m * LDR + (tn * NTW + j) * 32 + cl, acc[i][j][r]);
        }
    __syncthreads();
    for (int i = tid; i < p.Cout * p.Cin; i += 256) {
      const int co = i / p.Cin, ci = i - co * p.Cin;
      // STEM: column ci is the tap (kh*3 + kw)*3 + c; torch keeps the weight as [co][c][kh][kw]
      const int dst = STEM ? co * 27 + (ci % 3) * 9 + ci / 3 : i;
      atomicAdd(p.dw + dst, red[co * LDR + ci]);
    }
  }
}


template <int MT, int NT, int R, int WM, int WN, int BPC, bool KREG>
int launch(const WgradArgs& a, bool gate, hipStream_t st) {
  constexpr int LDZ = ld_for(MT), LDA = ld_for(NT);
  constexpr int WR = 4 / (WM * WN);
  const size_t tiles = ((size_t)R * (LDZ + LDA) + 3 * MT * 32 + 2 * NT * 32) * 4, red = WR == 1 ? 0 : (size_t)MT * 32 * NT * 32 * 4;
  const size_t smem = tiles > red ? tiles : red;
  const int64_t nchunks = (a.rows + R - 1) / R;
  const int64_t cap = 256 * BPC;                 // BPC resident blocks per CU (measured optimum: 2; 1 for the 10-tile shapes)
  const int blocks = (int)(nchunks < cap ? nchunks : cap);
  auto go = [&](auto k) {
    (void)ensure_dynamic_lds((const void*)k, smem);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), smem, st, a);
  };
  if (gate) go(conv1x1_wgrad_kernel<MT, NT, R, A_GATE, WM, WN, KREG>);
  else go(conv1x1_wgrad_kernel<MT, NT, R, A_PLAIN, WM, WN, KREG>);
  return check_launch("mt_conv1x1_wgrad");
}

}  // namespace

// _conv_stem weight gradient on the same kernel (one chunk = one output row of <= 128 pixels); called by mt_stem_conv_wgrad
int mt::stem_wgrad_mfma(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N, int H, int W,
                        int Ho, int Wo, int pad0, hipStream_t st) {
  constexpr int R = 128;
  WgradArgs a{du, z, kabc, nullptr, nullptr, nullptr, nullptr, dw, (int64_t)N * Ho * Wo, 32, 27, 1, x, x_is_u8, H, W, Ho, Wo, pad0};
  const size_t smem = ((size_t)R * (ld_for(1) + ld_for(1)) + 3 * 32 + 2 * 32) * 4;
  const int64_t nchunks = (int64_t)N * Ho;
  const int blocks = (int)(nchunks < 512 ? nchunks : 512);
  hipLaunchKernelGGL((conv1x1_wgrad_kernel<1, 1, R, A_STEM, 1, 1, true>), dim3(blocks), dim3(256), smem, st, a);
  return check_launch("mt_stem_conv_wgrad(mfma)");
}

// instances: (Cout tiles, Cin tiles, rows per chunk, tile groups along Cout, along Cin, resident blocks per CU to launch,
//             BatchNorm-backward coefficients in registers [1] or LDS [0])
#define MT_WGRAD_INSTANCES(X)                                                                                                   \
  X(1, 1, 128, 1, 1, 2, 1) X(1, 2, 64, 1, 1, 2, 1) X(2, 1, 64, 1, 1, 2, 1) X(2, 2, 64, 1, 1, 2, 1) X(3, 1, 64, 1, 1, 2, 1)         \
  X(1, 3, 64, 1, 1, 2, 1) X(4, 1, 64, 1, 1, 2, 1) X(1, 4, 64, 1, 1, 2, 1) X(5, 1, 32, 1, 1, 2, 1) X(1, 5, 32, 1, 1, 2, 1)         \
  X(2, 3, 32, 2, 1, 2, 1) X(3, 2, 32, 1, 2, 2, 1) X(2, 4, 32, 1, 4, 2, 0) X(4, 2, 32, 4, 1, 2, 0) X(2, 5, 32, 2, 1, 1, 0)         \
  X(5, 2, 32, 1, 2, 1, 0) X(2, 8, 32, 1, 4, 2, 0) X(8, 2, 32, 4, 1, 2, 0)

extern "C" int mt_conv1x1_wgrad_supported(int Cout, int Cin) {
  if ((Cout & 3) || (Cin & 3) || Cout <= 0 || Cin <= 0) return 0;
  const int mt_ = (Cout + 31) / 32, nt = (Cin + 31) / 32;
#define MT_CASE(M_, N_, R_, WM_, WN_, B_, K_) if (mt_ == M_ && nt == N_) return 1;
  MT_WGRAD_INSTANCES(MT_CASE)
#undef MT_CASE
  return 0;
}

extern "C" int mt_conv1x1_wgrad(const float* du, const float* z, const float* kabc, const float* x, const float* sc, const float* sh,
                                const float* gate, int hw, float* dw, int64_t rows, int Cout, int Cin, void* stream) {
  if (!du || !z || !kabc || !x || !dw) return fail(MT_ERR_ARG, "mt_conv1x1_wgrad: null pointer");
  if (!mt_conv1x1_wgrad_supported(Cout, Cin))
    return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_wgrad: %d x %d weights do not fit the accumulator-resident kernel", Cout, Cin);
  const bool g = gate != nullptr;
  if (g && (!sc || !sh || hw <= 0)) return fail(MT_ERR_ARG, "mt_conv1x1_wgrad: gate needs scale, shift and hw");
  if (((uintptr_t)du | (uintptr_t)z | (uintptr_t)x | (uintptr_t)kabc) & 15) return fail(MT_ERR_ARG, "mt_conv1x1_wgrad: 16-byte alignment");
  WgradArgs a{du, z, kabc, x, sc, sh, gate, dw, rows, Cout, Cin, hw, nullptr, 0, 0, 0, 0, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const int mt_ = (Cout + 31) / 32, nt = (Cin + 31) / 32;
#define MT_CASE(M_, N_, R_, WM_, WN_, B_, K_) if (mt_ == M_ && nt == N_) return launch<M_, N_, R_, WM_, WN_, B_, K_>(a, g, st);
  MT_WGRAD_INSTANCES(MT_CASE)
#undef MT_CASE
  return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_wgrad: no instance for %d x %d tiles", mt_, nt);
}
