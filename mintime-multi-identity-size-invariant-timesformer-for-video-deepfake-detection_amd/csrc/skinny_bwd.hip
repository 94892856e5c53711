// Data AND weight gradient of a 1x1 "expand" convolution with few input channels and very many rows in ONE streaming pass
// (EfficientNet-B0 blocks 1-3 at 112^2 / 56^2: 0.8-3.2 M rows, 96 / 144 expanded channels against 16 / 24 block channels;
// reference efficientnet_pytorch/model.py:96-99 driven backwards by train.py:371).
//
//   dz[r, co]  = ka[co]*du[r, co] + kb[co]*z[r, co] + kc[co]        BatchNorm backward of _bn0 (never stored), z = x . W^T
//   dx[r, ci]  = sum_co dz[r, co] * W[co, ci]  (+ res[r, ci])       data gradient  -> the block input's gradient
//   dW[co, ci] += sum_r dz[r, co] * x[r, ci]                        weight gradient
//
// As two launches (mt_conv1x1_rows mode 2 + mt_conv1x1_wgrad) each gradient streamed du AND z -- the two widest tensors of the
// backward pass (1.2 GB each for block 1 of a 256-crop batch): 4 passes over the expanded tensor.  This kernel makes ONE:
//
// (1) Both gradients come from the same dz tile, formed once per 64-row chunk.
// (2) z is not read at all.  dz is linear in z and z = x . W^T is linear in x, so
//       dx = (ka*du + kc) . W + x . G,                 G [Cin, Cin]  = W^T diag(kb) W      (built once per block)
//       dW = (ka*du + kc)^T . x + diag(kb) W S,        S [Cin, Cin]  = x^T x               (one more accumulator tile)
//     for Cin/2 more MFMA steps per data-gradient tile and one more tile per weight-gradient step.
// (3) One persistent 512-thread block per CU, split by ROLE and decoupled through double-buffered LDS tiles:
//       waves 4-7  PRODUCERS: stream 64-row chunks of du, x (res) from HBM -- two register sets, so the loads of chunks i+1 and
//                  i+2 are in flight while chunk i is multiplied -- apply the coefficients and stage ka*du + kc and x into LDS
//                  buffer i&1.  They also carry the data gradient OUT: the consumers leave it in an LDS tile (pre-loaded with the
//                  residual by the producers) and the producers write it back two chunks later with full-row float4 stores;
//       waves 0, 1 CONSUMERS, data gradient of a 32-row tile each (M = rows, K = Cout + Cin, W and G resident in LDS);
//       waves 2, 3 CONSUMERS, weight gradient of 32 of the chunk's rows each (K = rows; the [Cout + Cin, Cin] result stays in
//                  MFMA accumulators for the whole launch and the two halves meet in LDS at the end).
//     One barrier per chunk; the MFMA wavefronts touch no global memory and never stage.
// (4) N16 (Cin == 16, block 1): 16x16x4 MFMA tiles instead of 32x32x2 -- a 32-wide tile is half padding at 16 input channels
//     and the kernel is bound by the fp32 MFMA rate.
//
// History (256-crop batch, block 1 / blocks 2, 3, us per launch): two launches per gradient 1100 / 500; first fused version (every
// wavefront loads, stages, barrier, multiplies, barrier; du and z streamed) 675 / 330 -- knock-outs showed it at its HBM bound
// (z, du, x at 6.2 TB/s = 433 us) only because the phases serialised: compute alone took 518 us with the MFMA pipes 37% busy;
// producer / consumer split with z folded 715 / 270 -- the 16 scalar dx stores per lane of the data-gradient waves cost 165 us,
// hence the LDS hand-over; this version 365 / 240-335.
// Algorithmic bytes = rows * (Cout + 2*Cin [+ Cin for the residual]) * 4.
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FusedArgs {
  const float* du; const float* kabc;                      // [rows, Cout], [3, Cout]
  const float* x;                                            // [rows, Cin]
  const float* w;                                            // [Cout, Cin] (the forward weight)
  const float* res;                                          // optional [rows, Cin]
  float* dx;                                                 // [rows, Cin]
  float* dw;                                                 // [Cout, Cin], accumulated with atomics
  int64_t rows; int Cout, Cin;
};

constexpr int R = 64;                      // rows per chunk
constexpr int NTHR = 512, NPROD = 256;
constexpr int LDX = 33;                    // x tile [R][32]: read by column (weight gradient) and by row (fold term): odd pitch
constexpr int LDW = 33;                    // W tile [MT*32][33]: read as b[k = co][n = ci] with lanes over ci, k uniform per half-wave

template <int MT> struct Smem {
  static constexpr int LDZ = MT * 32 + 1;  // odd pitch: the data gradient reads dz by ROW (32 lanes = 32 rows -> 32 banks), the
                                           // weight gradient by column (consecutive lanes = consecutive floats): both conflict-free
  static constexpr int DZ = R * LDZ, XS = R * LDX, OUT = R * 32, WT = MT * 32 * LDW, G = 32 * 33, KAB = 3 * MT * 32;
  static constexpr int FLOATS = 2 * DZ + 2 * XS + 2 * OUT + WT + G + KAB;
};

// MT = 32-wide tiles of Cout (3: 96 channels, 5: 144 channels); Cin <= 32 (one tile)
template <int MT, bool N16>
__global__ __launch_bounds__(NTHR) void conv1x1_bwd_fused_kernel(FusedArgs p) {
  using S = Smem<MT>;
  constexpr int LDZ = S::LDZ;
  constexpr int VZ = R * MT * 32 / 4 / NPROD;          // float4 slots per producer thread covering [R, MT*32]
  static_assert(R * MT * 32 / 4 % NPROD == 0, "chunk must divide over the producers");
  constexpr int NACC = MT + 1;             // Cout tiles + S
  static_assert(2 * NACC * 32 * 32 <= 2 * S::DZ, "the end-of-launch reduction buffer reuses the dz tiles");
  extern __shared__ float smem[];
  float* dzs = smem;                       // [2][R][LDZ]
  float* xs = dzs + 2 * S::DZ;             // [2][R][LDX]
  float* outs = xs + 2 * S::XS;            // [2][R][32]  data gradient on its way out (+ residual)
  float* wt = outs + 2 * S::OUT;           // [MT*32][LDW]
  float* gm = wt + S::WT;                  // [32][33]   G = W^T diag(kb) W (fold)
  float* kab = gm + S::G;                  // [3][MT*32]  ka | kb | kc
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = lane >> 5, cl = lane & 31;

  for (int i = tid; i < S::FLOATS; i += NTHR) smem[i] = 0.f;       // padding rows / columns stay zero for the whole launch
  __syncthreads();
  for (int i = tid; i < p.Cout * p.Cin; i += NTHR) {
    const int co = i / p.Cin, ci = i - co * p.Cin;
    wt[co * LDW + ci] = p.w[i];
  }
  for (int i = tid; i < 3 * MT * 32; i += NTHR) {
    const int which = i / (MT * 32), c = i - which * MT * 32;
    kab[i] = c < p.Cout ? p.kabc[which * p.Cout + c] : 0.f;
  }
  __syncthreads();
  {
    for (int i = tid; i < p.Cin * p.Cin; i += NTHR) {
      const int a = i / p.Cin, b = i - a * p.Cin;
      float s = 0.f;
      for (int co = 0; co < p.Cout; ++co) s = fmaf(kab[MT * 32 + co] * wt[co * LDW + a], wt[co * LDW + b], s);
      gm[a * 33 + b] = s;
    }
  }

  const int64_t nchunks = (p.rows + R - 1) / R;
  const int n_it = blockIdx.x < nchunks ? (int)((nchunks - 1 - blockIdx.x) / gridDim.x) + 1 : 0;   // this block's chunks
  auto chunk_of = [&](int j) { return (int64_t)blockIdx.x + (int64_t)j * gridDim.x; };
  __syncthreads();

  if (wave >= 4) {
    // ================================================= PRODUCERS =================================================
    const int ptid = tid - NPROD;
    const int cq = p.Cout >> 2, aq = p.Cin >> 2;
    int zr[VZ], zc[VZ];                     // loop-invariant placement of this thread's slots: [R, Cout] as float4 along Cout
#pragma unroll
    for (int i = 0; i < VZ; ++i) {
      const int idx = ptid + NPROD * i;
      zr[i] = idx / cq;
      zc[i] = (idx - zr[i] * cq) * 4;
      if (zr[i] >= R) { zr[i] = -1; zc[i] = 0; }
    }
    int xr_[2], xc_[2];                     // two float4 slots per thread cover [R, Cin <= 32]
    bool x_on[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = ptid + NPROD * i;
      xr_[i] = idx / aq;
      xc_[i] = (idx - xr_[i] * aq) * 4;
      x_on[i] = xr_[i] < R;
      if (!x_on[i]) { xr_[i] = 0; xc_[i] = 0; }
    }
    float4 rdu[2][VZ];
    float4 rx0_0, rx0_1, rx1_0, rx1_1, rres0_0, rres0_1, rres1_0, rres1_1;     // named, not arrays: as [2][2] arrays one of them lands in scratch
    // the residual is loaded and staged unconditionally (without one: an L1 hit on x, overwritten by the consumers' plain store);
    // under `if (p.res)` the compiler keeps the register set in scratch memory
    const float* resp = p.res ? p.res : p.x;

    // unconditional loads on clamped rows (a predicated load de-pipelines: skinny_wgrad.hip)
#define MT_FETCH(SET, CHUNK)                                                                                        \
    {                                                                                                               \
      const int64_t r0 = (CHUNK) * R;                                                                               \
      const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;                                            \
      const float* du_c = p.du + r0 * p.Cout;                                                                       \
      _Pragma("unroll") for (int i = 0; i < VZ; ++i) {                                                              \
        const int off = min(max(zr[i], 0), last) * p.Cout + zc[i];                                                  \
        rdu[SET][i] = *reinterpret_cast<const float4*>(du_c + off);                                                 \
      }                                                                                                             \
      rx##SET##_0 = *reinterpret_cast<const float4*>(p.x + (r0 + min(xr_[0], last)) * p.Cin + xc_[0]);              \
      rx##SET##_1 = *reinterpret_cast<const float4*>(p.x + (r0 + min(xr_[1], last)) * p.Cin + xc_[1]);              \
      rres##SET##_0 = *reinterpret_cast<const float4*>(resp + (r0 + min(xr_[0], last)) * p.Cin + xc_[0]);           \
      rres##SET##_1 = *reinterpret_cast<const float4*>(resp + (r0 + min(xr_[1], last)) * p.Cin + xc_[1]);           \
    }
#define MT_STAGE_X(I, RX, RRES, BUF)                                                                                \
      if (x_on[I]) {                                                                                                \
        const bool ok = xr_[I] < left;                                                                              \
        float* dst = xb + xr_[I] * LDX + xc_[I];                                                                    \
        dst[0] = ok ? RX.x : 0.f; dst[1] = ok ? RX.y : 0.f; dst[2] = ok ? RX.z : 0.f; dst[3] = ok ? RX.w : 0.f;     \
        *reinterpret_cast<float4*>(outs + (BUF) * S::OUT + xr_[I] * 32 + xc_[I]) = RRES;                            \
      }
#define MT_STAGE(SET, CHUNK, BUF)                                                                                   \
    {                                                                                                               \
      const int64_t r0 = (CHUNK) * R;                                                                               \
      const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);                                                \
      float* dzb = dzs + (BUF) * S::DZ;                                                                             \
      float* xb = xs + (BUF) * S::XS;                                                                               \
      _Pragma("unroll") for (int i = 0; i < VZ; ++i) {                                                              \
        if (zr[i] >= 0) {                                                                                           \
          const bool ok = zr[i] < left;                                                                             \
          float* dst = dzb + zr[i] * LDZ + zc[i];                                                                   \
          const float4 ka = *reinterpret_cast<const float4*>(kab + zc[i]);                                          \
          const float4 kc = *reinterpret_cast<const float4*>(kab + 2 * MT * 32 + zc[i]);                            \
          float4 v = make_float4(fmaf(ka.x, rdu[SET][i].x, kc.x), fmaf(ka.y, rdu[SET][i].y, kc.y),                  \
                                 fmaf(ka.z, rdu[SET][i].z, kc.z), fmaf(ka.w, rdu[SET][i].w, kc.w));                 \
          dst[0] = ok ? v.x : 0.f; dst[1] = ok ? v.y : 0.f; dst[2] = ok ? v.z : 0.f; dst[3] = ok ? v.w : 0.f;       \
        }                                                                                                           \
      }                                                                                                             \
      MT_STAGE_X(0, rx##SET##_0, rres##SET##_0, BUF)                                                                \
      MT_STAGE_X(1, rx##SET##_1, rres##SET##_1, BUF)                                                                \
    }
    /* the data gradient of a finished chunk: LDS -> HBM, one float4 per slot (the chunk is one contiguous block of dx) */
#define MT_DRAIN(CHUNK, BUF)                                                                                        \
    {                                                                                                               \
      const int64_t r0 = (CHUNK) * R;                                                                               \
      const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);                                                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
        if (x_on[i] && xr_[i] < left)                                                                               \
          *reinterpret_cast<float4*>(p.dx + (r0 + xr_[i]) * p.Cin + xc_[i]) =                                       \
              *reinterpret_cast<const float4*>(outs + (BUF) * S::OUT + xr_[i] * 32 + xc_[i]);                       \
    }
    // memory operations retire in order: staging set 0 waits for ITS loads only and leaves set 1's (issued later) in flight
    if (n_it > 0) MT_FETCH(0, chunk_of(0));
    if (n_it > 1) MT_FETCH(1, chunk_of(1));
    // barrier k (k = 1, 2, ...) publishes chunk k - 1; the consumers reach it after finishing chunk k - 2, so past barrier k the
    // producers may refill buffer k & 1 and drain the data gradient of chunk k - 2 from the same side of the out tile
    for (int j = 0; j < n_it; j += 2) {
      if (j >= 2) MT_DRAIN(chunk_of(j - 2), 0);
      MT_STAGE(0, chunk_of(j), 0);
      if (j + 2 < n_it) MT_FETCH(0, chunk_of(j + 2));
      lds_barrier();
      if (j + 1 < n_it) {
        if (j >= 2) MT_DRAIN(chunk_of(j - 1), 1);
        MT_STAGE(1, chunk_of(j + 1), 1);
        if (j + 3 < n_it) MT_FETCH(1, chunk_of(j + 3));
        lds_barrier();
      }
    }
    __syncthreads();                        // every consumer is past its last chunk (pairs with their barrier before the reduction)
    if (n_it >= 2) MT_DRAIN(chunk_of(n_it - 2), (n_it - 2) & 1);
    if (n_it >= 1) MT_DRAIN(chunk_of(n_it - 1), (n_it - 1) & 1);
#undef MT_FETCH
#undef MT_STAGE
#undef MT_STAGE_X
#undef MT_DRAIN
  } else {
    // ================================================= CONSUMERS =================================================
    constexpr int NW = N16 ? 1 : NACC;      // 32x32 accumulator tiles (generic path)
    constexpr int NW4 = N16 ? 2 * MT + 1 : 1;          // 16x16 accumulator tiles (N16 path): 16-wide Cout tiles (+ S)
    f32x16 wacc[NW];                        // weight-gradient waves: dW tiles [32 co][32 ci] (+ S = x^T x), live for the whole launch
    f32x4 wacc4[NW4];
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NW4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) wacc4[i][r] = 0.f;
    const int l16 = lane & 15, q = lane >> 4;
    for (int j = 0; j < n_it; ++j) {
      const float* dzb = dzs + (j & 1) * S::DZ;
      const float* xb = xs + (j & 1) * S::XS;
      float* ob = outs + (j & 1) * S::OUT;
      lds_barrier();                        // buffer j & 1 staged
      if (wave < 2) {
        if constexpr (N16) {
          // ---- data gradient of rows [wave*32, +32): two 16x16 tiles (two independent chains), k = Cout in steps of 4
          f32x4 c0, c1;                       // start from the residual the producers left in the out tile
          float* o = ob + (wave * 32 + 4 * q) * 32 + l16;          // c[r] = out[4q + r][l16]
#pragma unroll
          for (int r = 0; r < 4; ++r) { c0[r] = p.res ? o[r * 32] : 0.f; c1[r] = p.res ? o[(16 + r) * 32] : 0.f; }
          const float* a_w = dzb + (wave * 32 + l16) * LDZ + q;
          const float* b_w = wt + q * LDW + l16;
          float a0[4], a1[4], b0[4], e0[4], e1[4], f0[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { a0[u] = a_w[4 * u]; a1[u] = a_w[16 * LDZ + 4 * u]; b0[u] = b_w[4 * u * LDW]; }
#pragma unroll 1
          for (int ks = 0; ks < MT * 8; ks += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              e0[u] = a_w[4 * (ks + 4 + u)]; e1[u] = a_w[16 * LDZ + 4 * (ks + 4 + u)]; f0[u] = b_w[4 * (ks + 4 + u) * LDW];
            }
            __builtin_amdgcn_sched_barrier(0);    // keep the requests ahead of the multiplies (the scheduler sinks them otherwise)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b0[u], c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kn = ks + 8 < MT * 8 ? ks + 8 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0[u] = a_w[4 * (kn + u)]; a1[u] = a_w[16 * LDZ + 4 * (kn + u)]; b0[u] = b_w[4 * (kn + u) * LDW]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[u], f0[u], c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[u], f0[u], c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          {
            const float* a_x = xb + (wave * 32 + l16) * LDX + q;
            const float* b_g = gm + q * 33 + l16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                       // Cin == 16
              const float g = b_g[4 * ks * 33];
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_x[4 * ks], g, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_x[16 * LDX + 4 * ks], g, c1, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { o[r * 32] = c0[r]; o[(16 + r) * 32] = c1[r]; }
        } else {
          // ---- data gradient of row tile `wave`: out[32 rows][32 ci] = dz[32 rows][Cout] . W[Cout][32 ci]  (+ x[32 rows][Cin] . G)
          // two accumulator chains (even / odd k-steps) so consecutive MFMAs do not depend on each other
          f32x16 acc, acc2;
          float* o = ob + (wave * 32 + 4 * kh) * 32 + cl;          // acc[r] = out[(r & 3) + 8 * (r >> 2) + 4 * kh][cl]
          float rcur[16];                     // the residual the producers left in the out tile: read now, added at the end
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; rcur[r] = p.res ? o[((r & 3) + 8 * (r >> 2)) * 32] : 0.f; }
          const float* a_w = dzb + (wave * 32 + cl) * LDZ + kh;
          const float* b_w = wt + kh * LDW + cl;
          // one wavefront per SIMD multiplies: nothing else hides the LDS latency, so the operands of the next four steps are
          // requested before the current four are multiplied (LDS returns in order: the wait is counted, not a drain)
          float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { a0[u] = a_w[2 * u]; b0[u] = b_w[2 * u * LDW]; }
#pragma unroll 1
          for (int ks = 0; ks < MT * 16; ks += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { a1[u] = a_w[2 * (ks + 4 + u)]; b1[u] = b_w[2 * (ks + 4 + u) * LDW]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u + 1], b0[u + 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kn = ks + 8 < MT * 16 ? ks + 8 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0[u] = a_w[2 * (kn + u)]; b0[u] = b_w[2 * (kn + u) * LDW]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u + 1], b1[u + 1], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          {
            const float* a_x = xb + (wave * 32 + cl) * LDX + kh;
            const float* b_g = gm + kh * 33 + cl;
#pragma unroll 2
            for (int ks = 0; ks < (p.Cin >> 1); ks += 2) {         // Cin % 4 == 0: an even number of steps
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_x[2 * ks], b_g[2 * ks * 33], acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_x[2 * ks + 2], b_g[(2 * ks + 2) * 33], acc2, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * 32] = acc[r] + acc2[r] + rcur[r];
        }
      } else {
        const int rb = (wave - 2) * 32;
        if constexpr (N16) {
          // ---- weight gradient over rows [rb, +32): 16x16 tiles dW[16 co][16 ci], k = rows in steps of 4
          const float* dz_w = dzb + (rb + q) * LDZ + l16;
          const float* x_w = xb + (rb + q) * LDX + l16;
          float d0[2 * MT], d1[2 * MT], x0, x1;
          x0 = x_w[0];
#pragma unroll
          for (int i = 0; i < 2 * MT; ++i) d0[i] = dz_w[i * 16];
#pragma unroll 1
          for (int ks = 0; ks < 8; ks += 2) {
            x1 = x_w[4 * (ks + 1) * LDX];
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) d1[i] = dz_w[4 * (ks + 1) * LDZ + i * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) wacc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0[i], x0, wacc4[i], 0, 0, 0);
            wacc4[2 * MT] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x0, wacc4[2 * MT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = ks + 2 < 8 ? ks + 2 : 0;
            x0 = x_w[4 * k2 * LDX];
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) d0[i] = dz_w[4 * k2 * LDZ + i * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) wacc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[i], x1, wacc4[i], 0, 0, 0);
            wacc4[2 * MT] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, x1, wacc4[2 * MT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // ---- weight gradient over rows [rb, +32) of the chunk, all Cout tiles: dW[co][ci] += dz[r][co] * x[r][ci], k = rows
          const float* dz_w = dzb + rb * LDZ + cl;
          const float* x_w = xb + rb * LDX + cl;
          float d0[MT], d1[MT], x0, x1;       // same one-step-ahead operand prefetch as the data-gradient waves
          x0 = x_w[kh * LDX];
#pragma unroll
          for (int i = 0; i < MT; ++i) d0[i] = dz_w[kh * LDZ + i * 32];
#pragma unroll 1
          for (int ks = 0; ks < 16; ks += 2) {
            const int r1 = 2 * (ks + 1) + kh;
            x1 = x_w[r1 * LDX];
#pragma unroll
            for (int i = 0; i < MT; ++i) d1[i] = dz_w[r1 * LDZ + i * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) wacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0[i], x0, wacc[i], 0, 0, 0);
            wacc[MT] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, x0, wacc[MT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int r2 = 2 * (ks + 2 < 16 ? ks + 2 : 0) + kh;
            x0 = x_w[r2 * LDX];
#pragma unroll
            for (int i = 0; i < MT; ++i) d0[i] = dz_w[r2 * LDZ + i * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) wacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[i], x1, wacc[i], 0, 0, 0);
            wacc[MT] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, x1, wacc[MT], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // the two weight-gradient waves meet in LDS (the dz tiles are free once every consumer is past its last chunk)
    __syncthreads();
    if (wave >= 2) {
      float* red_w = dzs + (wave - 2) * NACC * 32 * 32;            // [NACC*32 rows: co, then S's c][32 ci]
      if constexpr (N16) {
#pragma unroll
        for (int i = 0; i < NW4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) red_w[(i * 16 + 4 * q + r) * 32 + l16] = wacc4[i][r];      // tile 2*MT = S at row MT*32
      } else {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) red_w[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + cl] = wacc[i][r];
      }
    }
  }
  const float* red = dzs;                   // [2][NACC*32][32]; then one global atomic per weight and block

  __syncthreads();
  if (n_it > 0)
    for (int i = tid; i < p.Cout * p.Cin; i += NTHR) {
      const int co = i / p.Cin, ci = i - co * p.Cin;
      float v = red[co * 32 + ci] + red[(NACC * 32 + co) * 32 + ci];
      {                 // + kb[co] * sum_c W[co][c] * S[c][ci]
        float s = 0.f;
        for (int c = 0; c < p.Cin; ++c)
          s = fmaf(wt[co * LDW + c], red[(MT * 32 + c) * 32 + ci] + red[(NACC * 32 + MT * 32 + c) * 32 + ci], s);
        v = fmaf(kab[MT * 32 + co], s, v);
      }
      atomicAdd(p.dw + i, v);
    }
}

template <int MT, bool N16>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  const size_t smem = (size_t)Smem<MT>::FLOATS * 4;
  const int64_t nchunks = (a.rows + R - 1) / R;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int blocks = (int)(nchunks < cus ? nchunks : cus);          // one persistent block per CU
  auto k = conv1x1_bwd_fused_kernel<MT, N16>;
  hipError_t e = ensure_dynamic_lds((const void*)k, smem);
  if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_conv1x1_bwd_fused: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(NTHR), smem, st, a);
  return check_launch("mt_conv1x1_bwd_fused");
}

}  // namespace

extern "C" int mt_conv1x1_bwd_fused_supported(int Cout, int Cin) {
  if ((Cout & 3) || (Cin & 3) || Cin <= 0 || Cin > 32) return 0;
  const int mt_ = (Cout + 31) / 32;
  return Cout == mt_ * 32 ? (mt_ == 3) : (mt_ == 5 && Cout == 144);     // 96 -> <= 32 and 144 -> <= 32 channels
}

extern "C" int mt_conv1x1_bwd_fused(const float* du, const float* kabc, const float* x, const float* w, const float* res, float* dx,
                                    float* dw, int64_t rows, int Cout, int Cin, void* stream) {
  if (!du || !kabc || !x || !w || !dx || !dw) return fail(MT_ERR_ARG, "mt_conv1x1_bwd_fused: null pointer");
  if (!mt_conv1x1_bwd_fused_supported(Cout, Cin))
    return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_bwd_fused: no instance for %d -> %d channels", Cout, Cin);
  if (((uintptr_t)du | (uintptr_t)x | (uintptr_t)kabc | (uintptr_t)res | (uintptr_t)dx) & 15) return fail(MT_ERR_ARG, "mt_conv1x1_bwd_fused: 16-byte alignment");
  FusedArgs a{du, kabc, x, w, res, dx, dw, rows, Cout, Cin};
  hipStream_t st = (hipStream_t)stream;
  if ((Cout + 31) / 32 == 3) return Cin == 16 ? launch_fused<3, true>(a, st) : launch_fused<3, false>(a, st);
  return launch_fused<5, false>(a, st);
}
