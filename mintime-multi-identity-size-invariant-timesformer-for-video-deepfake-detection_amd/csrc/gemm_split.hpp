// Split-operand GEMM main loop for gfx950:  fp32 in, fp32 out, fp32 accumulate, products on the bf16 matrix pipe.
//
// gfx950 has no TF32/xf32 and its fp32 MFMA runs at 1/16 of the bf16 rate (v_mfma_f32_32x32x2_f32: 64 cycles for K = 2;
// v_mfma_f32_32x32x16_bf16: 32 cycles for K = 16).  This loop keeps fp32 semantics and moves the products to the bf16 pipe:
// every fp32 operand value is split EXACTLY into three bf16 pieces (round-to-nearest at each level)
//       x = x0 + x1 + x2,   |x1| <= 2^-9 |x|,  |x2| <= 2^-18 |x|
// and  x*y  is accumulated from the six piece products of weight >= 2^-18:
//       x0y0 + (x0y1 + x1y0) + (x0y2 + x1y1 + x2y0)                 (dropped: x1y2 + x2y1 + x2y2 <= 2^-26 |x||y|)
// bf16 x bf16 products are exact in fp32 and land in the same fp32 accumulator registers the fp32 MFMA uses, so the result
// differs from the fp32 pipe's by less than that pipe's own rounding (tests/test_gpu_gemm.py compares both with fp64).
// Six bf16 MFMAs replace sixteen fp32 ones: 6/16 of the matrix-pipe time.
//
// BAL (balanced accumulators).  v_mfma_f32_32x32x16_bf16 does not round  C + sum_k a_k b_k  like an fmaf chain: measured on
// gfx950 (tools/lab/mfma_round_probe.hip; profiles/r02_mfma_bf16_rounding_probe.txt, r02_split_accuracy_probe.txt) its result carries a sign-INDEPENDENT bias of about -3e-9 x sum|a||b|
// per instruction (rms error equal to the fp32 pipe's, but a mean that the fp32 pipe does not have).  Invisible in one GEMM's
// max error, it adds up coherently over a deep network (Xception's 36 convolutions moved a logit by 1.6e-3 relative).  Because
// the bias does not depend on the sign of the products it cancels between two accumulators fed with opposite signs: even k-tiles
// accumulate  +x*y  into acc, odd k-tiles accumulate  (-x0)*y  into nacc (the leading A piece is stored negated), and the
// result is acc - nacc.  Same MFMA count; 16 x TM x TN more accumulator registers.
//
// The split costs ~4.6 VALU instructions per element, so it must run ONCE per element per block, not once per fragment read
// (gemm_dma.hpp's MMA_BF16X6 does the latter: VALU-bound at 1.3x).  Hence this loop stages through registers:
//   global fp32 -> VGPRs (8 consecutive k of one tile row/column per thread) -> split -> three bf16 "planes" in LDS
//   -> ds_read_b128 fragments (8 k of the lane's row: the 32x32x16 operand layout) -> 6 x TM x TN MFMAs per 16 k.
// k-major operands (dgrad's weight, both wgrad operands) are transposed for free by the load pattern: lane = column, eight
// strided dword loads (a wave covers 256 contiguous bytes per k row), so the LDS image is the same [rows][16 k] for every layout.
//
// LDS per stage: (BM + BN) rows x 16 k x 2 B x 3 planes; two stages (one barrier per 16 k).  Row = 32 B = two 16-byte slots
// (k 0-7, k 8-15); slot ^= (row >> 3) & 1 makes the 16 rows a ds_read_b128 serves per cycle cover all 64 banks.
#pragma once
#include "gemm_dma.hpp"
#include <type_traits>

#ifndef MT_SPLIT_A_DEPTH       // B-planes loop: A register sets in flight (2 = one k-step of prefetch distance, 3 = two).  Measured equal
#define MT_SPLIT_A_DEPTH 2     // (profiles/r03_split_planes_manual_waits.txt): the loop is not waiting for its operands.
#endif
#ifndef MT_SPLIT_ABLATE        // tuning lab only: 1 no global loads, 2 no split + LDS writes, 4 no barrier, 8 no fragment reads, 16 no epilogue,
                               // 32 B operand neither loaded nor split after the first tile (upper bound of pre-split weight planes)
#define MT_SPLIT_ABLATE 0
#endif

namespace mt {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// BPL (B planes): the B operand arrives already split -- three bf16 planes [N][K] in memory (weights: split once per step by
// mt_split_planes) -- and goes global -> LDS by DMA (global_load_lds_dwordx4, no staging VGPRs, no VALU, no ds_write) through a
// ring of three B stages: tile kt+2's planes are issued at the top of step kt, so they have two steps to land.
template <int WAVES_M, int WAVES_N, int TM, int TN, int AL, int BL, int EPI, int MINW, bool X6 = true, int PIPE = 2, bool BAL = false,
          int PRO = PRO_NONE, bool BPL = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINW)
void gemm_split_kernel(const GemmArgs p) {
  static_assert(!BPL || (PIPE == 2 && X6 && AL == LAYOUT_KCONTIG && BL == LAYOUT_KCONTIG && PRO == PRO_NONE), "B planes: NT form only");
  static_assert(!BAL || PIPE == 2, "the balanced accumulators need the stage = tile parity of the two-register-set loop");
  // PRO_BN_SWISH_GATE (k-contiguous A, two-register-set loop): a = swish(z*scale[k]+shift[k]) * gate[(m/hw)*K+k], applied to the
  // staged registers right before the split -- VALU work that rides in the MFMA shadow like the split itself.  The per-k vectors
  // and the gate rows of the images this row tile touches are cached in LDS behind the two plane stages.
  // PRO_BN_BWD (either A layout, two-register-set loop): a = ka[c]*A + kb[c]*A2 + kc[c], c = the operand's channel index (k for a
  // k-contiguous A: data gradients; the output row m for a k-major A: weight gradients).  A2's granules ride in a second half of
  // the staging registers; the coefficient vectors are cached in LDS (k-contiguous) or held per granule (k-major).
  static_assert(PRO == PRO_NONE || (PRO == PRO_BN_SWISH_GATE && AL == LAYOUT_KCONTIG && PIPE == 2) || (PRO == PRO_BN_BWD && PIPE == 2 && !BPL),
                "unsupported prologue");
  constexpr bool BNB = PRO == PRO_BN_BWD;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = NW * 64;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int BK = 16;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;           // bytes
  constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
  constexpr int AG = (BM * 2 + NT - 1) / NT, BG = (BN * 2 + NT - 1) / NT;   // 8-k granules per thread
  constexpr bool A_ALL = (BM * 2) % NT == 0, B_ALL = (BN * 2) % NT == 0;   // every thread stages AG / BG granules (no predicate)
  constexpr int AG2 = BNB ? 2 * AG : AG;            // staging granules for A (+ A2)
  // LDS map.  Staged B: two stages of {A planes, B planes}.  B planes by DMA: two A stages, then a ring of three B stages.
  constexpr int A_STAGE = BPL ? 3 * A_PLANE : STAGE;             // byte distance between the two A stages
  constexpr int B_BASE = BPL ? 6 * A_PLANE : 3 * A_PLANE;        // first B stage
  constexpr int B_STAGE = BPL ? 3 * B_PLANE : STAGE;
  constexpr int LDS_END = BPL ? 6 * A_PLANE + 9 * B_PLANE : 2 * STAGE;
  constexpr int BI = BPL ? (6 * BN / 64) / NW : 1;               // DMA wave-instructions per wave per B tile (3 planes x BN rows x 2 slots)
  static_assert(!BPL || (6 * BN / 64) % NW == 0, "B plane tile must divide into whole wave-instructions per wave");
  static_assert((BM * 2) % NT == 0 || BM * 2 < NT, "A granules must divide over the block");
  static_assert((BN * 2) % NT == 0 || BN * 2 < NT, "B granules must divide over the block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  int mt_, nt_;
  int k_begin = 0, k_end = p.K;
  int split_idx = blockIdx.y;
  if (p.xcd_k) {
    // Split-K weight gradient, K-range-major over the XCDs.  Workgroups are dealt round-robin to the 8 XCDs in flattened
    // (y, x) order; with the split index on blockIdx.y every XCD ran a slice of the TILES for ALL K-ranges, so each XCD's L2
    // fetched the whole second operand (and most of the first) for itself: 3.5x the algorithmic bytes per launch.  Here XCD c owns
    // the K-ranges c, c + 8, ...: inside one range the operand panels are shared by tiles that run back to back on the SAME L2
    // (the larger operand's panel index is the slow one), and nothing is read by two XCDs.
    const int tiles = gridDim.x, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int split = xcd + 8 * (idx / tiles), t = idx % tiles;
    const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
    if (m_tiles >= n_tiles) { mt_ = t / n_tiles; nt_ = t - mt_ * n_tiles; }
    else { nt_ = t / m_tiles; mt_ = t - nt_ * m_tiles; }
    k_begin = split * p.k_chunk;
    k_end = min(p.K, k_begin + p.k_chunk);
    if (k_begin >= k_end) return;
    split_idx = split;
  } else {
    if (!tile_coords<BM, BN>(p, mt_, nt_)) return;
    if (p.k_chunk > 0) {
      k_begin = blockIdx.y * p.k_chunk;
      k_end = min(p.K, k_begin + p.k_chunk);
      if (k_begin >= k_end) return;
    }
  }
  const int m0 = mt_ * BM;
  const int n0 = nt_ * BN;
  const int nk = (k_end - k_begin + BK - 1) / BK;   // host guarantees (k_end - k_begin) % 8 == 0: the last tile may hold one 8-k granule only
  const bool k_tail = ((k_end - k_begin) & 15) != 0;

  // ---- the granules this thread stages: (tile row, k half) -> global source, LDS byte offset inside a plane
  const float* a_src[AG];
  const float* a2_src[BNB ? AG : 1];
  float bn_ka[BNB && AL == LAYOUT_KMAJOR ? AG : 1], bn_kb[BNB && AL == LAYOUT_KMAJOR ? AG : 1], bn_kc[BNB && AL == LAYOUT_KMAJOR ? AG : 1];
  const float* b_src[BG];
  int a_dst[AG], b_dst[BG];
  int a_kh[AG], b_kh[BG];                           // k-major: the granule's k half
  bool a_on[AG], b_on[BG];
#pragma unroll
  for (int j = 0; j < AG; ++j) {
    const int gi = tid + j * NT;
    a_on[j] = A_ALL || gi < BM * 2;
    int row, kh;
    if constexpr (AL == LAYOUT_KCONTIG) { row = gi >> 1; kh = gi & 1; } else { row = gi % BM; kh = gi / BM; }
    row = a_on[j] ? row : 0; kh = a_on[j] ? kh : 0;
    int m = m0 + row;
    m = m < p.M ? m : p.M - 1;                      // rows past the end are never stored: any valid row will do
    if constexpr (AL == LAYOUT_KCONTIG) a_src[j] = p.A + map_row(p.a_map, m) * p.lda + k_begin + kh * 8;
    else a_src[j] = p.A + m;
    if constexpr (BNB) {
      a2_src[j] = p.A2 + (a_src[j] - p.A);
      if constexpr (AL == LAYOUT_KMAJOR) { bn_ka[j] = p.scale[m]; bn_kb[j] = p.shift[m]; bn_kc[j] = p.gate[m]; }
    }
    a_kh[j] = kh;
    a_dst[j] = row * 32 + ((kh ^ ((row >> 3) & 1)) << 4);
  }
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int gi = tid + j * NT;
    b_on[j] = B_ALL || gi < BN * 2;
    int row, kh;
    if constexpr (BL == LAYOUT_KCONTIG) { row = gi >> 1; kh = gi & 1; } else { row = gi % BN; kh = gi / BN; }
    row = b_on[j] ? row : 0; kh = b_on[j] ? kh : 0;
    int n;
    if constexpr (EPI == EPI_GEGLU) {               // tile row -> weight row: 'a' and 'gate' halves interleaved per 32 columns
      static_assert(EPI != EPI_GEGLU || TN == 2, "GEGLU wants TN == 2");
      const int w = row / 64, sel = (row >> 5) & 1, c = row & 31;
      const int jj = (n0 >> 1) + w * 32 + c;
      n = jj < p.n_half ? sel * p.n_half + jj : 0;
    } else {
      n = n0 + row;
      n = n < p.N ? n : p.N - 1;
    }
    if constexpr (BL == LAYOUT_KCONTIG) b_src[j] = p.B + (int64_t)n * p.ldb + k_begin + kh * 8;
    else b_src[j] = p.B + n;
    b_kh[j] = kh;
    b_dst[j] = row * 32 + ((kh ^ ((row >> 3) & 1)) << 4);
  }

  float* pv = reinterpret_cast<float*>(smem_split + LDS_END);      // PRO: [scale K | shift K | gate rows of the tile's images]
  int g_off[AG];
  if constexpr (PRO == PRO_BN_SWISH_GATE) {
    const int K = p.K;
    const int img_lo = m0 / p.hw;
    const int img_hi = min(m0 + BM - 1, p.M - 1) / p.hw;
    for (int i = tid; i < K; i += NT) { pv[i] = p.scale[i]; pv[K + i] = p.shift[i]; }
    for (int i = tid; i < (img_hi - img_lo + 1) * K; i += NT) pv[2 * K + i] = p.gate[(int64_t)img_lo * K + i];
#pragma unroll
    for (int j = 0; j < AG; ++j) {
      const int row = (tid + j * NT) >> 1;
      g_off[j] = (2 + min(m0 + row, p.M - 1) / p.hw - img_lo) * K;
    }
    __syncthreads();
  }

  if constexpr (BNB && AL == LAYOUT_KCONTIG) {
    const int K = p.K;
    for (int i = tid; i < K; i += NT) { pv[i] = p.scale[i]; pv[K + i] = p.shift[i]; pv[2 * K + i] = p.gate[i]; }
    __syncthreads();
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_split;
  const __bf16* bp_src[BI];
  if constexpr (BPL) {
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int gi = (wave * BI + j) * 64 + lane;                // granule of the stage image: plane, row, slot
      const int plane = gi / (2 * BN), within = gi % (2 * BN);
      const int row = within >> 1, slot = within & 1;
      const int kh = slot ^ ((row >> 3) & 1);
      int n;
      if constexpr (EPI == EPI_GEGLU) {
        const int w = row / 64, sel = (row >> 5) & 1, c = row & 31;
        const int jj = (n0 >> 1) + w * 32 + c;
        n = jj < p.n_half ? sel * p.n_half + jj : 0;
      } else {
        n = n0 + row;
        n = n < p.N ? n : p.N - 1;
      }
      bp_src[j] = reinterpret_cast<const __bf16*>(p.b_planes) + plane * p.b_pstride + (int64_t)n * p.ldb + k_begin + kh * 8;
    }
  }
  auto dma_b = [&](int kt, int ring) {                           // B planes of k-tile kt -> ring slot
    if constexpr (BPL) {
      const unsigned dst = lds_base + (unsigned)(B_BASE + ring * B_STAGE);
#pragma unroll
      for (int j = 0; j < BI; ++j)
        lds_dma16(reinterpret_cast<const float*>(bp_src[j] + kt * BK), dst + (unsigned)((wave * BI + j) * 1024));
    }
  };

  auto gload = [&](int kt, float (&ga)[AG2][8], float (&gb)[BG][8]) {
    if (MT_SPLIT_ABLATE & 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int j = 0; j < AG; ++j) ga[j][e] = (float)(lane + e + kt);
#pragma unroll
        for (int j = 0; j < BG; ++j) gb[j][e] = (float)(lane - e + kt);
      }
      return;
    }
    const int k0 = k_begin + kt * BK;
#pragma unroll
    for (int j = 0; j < AG; ++j) {
      if (!A_ALL && !a_on[j]) continue;
      if (k_tail && kt == nk - 1 && a_kh[j]) {         // second granule of a half-filled last tile: zeros (sstore skips the prologue there)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ga[j][e] = 0.f; if constexpr (BNB) ga[AG + j][e] = 0.f; }
        continue;
      }
      if constexpr (AL == LAYOUT_KCONTIG) {
        const float4 u = *reinterpret_cast<const float4*>(a_src[j] + kt * BK), v = *reinterpret_cast<const float4*>(a_src[j] + kt * BK + 4);
        ga[j][0] = u.x; ga[j][1] = u.y; ga[j][2] = u.z; ga[j][3] = u.w; ga[j][4] = v.x; ga[j][5] = v.y; ga[j][6] = v.z; ga[j][7] = v.w;
        if constexpr (BNB) {
          const float4 u2 = *reinterpret_cast<const float4*>(a2_src[j] + kt * BK), v2 = *reinterpret_cast<const float4*>(a2_src[j] + kt * BK + 4);
          ga[AG + j][0] = u2.x; ga[AG + j][1] = u2.y; ga[AG + j][2] = u2.z; ga[AG + j][3] = u2.w;
          ga[AG + j][4] = v2.x; ga[AG + j][5] = v2.y; ga[AG + j][6] = v2.z; ga[AG + j][7] = v2.w;
        }
      } else {
        const int kr = k0 + a_kh[j] * 8;
        if (p.a_map.gin == 0) {
          const float* s = a_src[j] + (int64_t)kr * p.lda;
#pragma unroll
          for (int e = 0; e < 8; ++e) ga[j][e] = s[(int64_t)e * p.lda];
          if constexpr (BNB) {
            const float* s2 = a2_src[j] + (int64_t)kr * p.lda;
#pragma unroll
            for (int e = 0; e < 8; ++e) ga[AG + j][e] = s2[(int64_t)e * p.lda];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) ga[j][e] = a_src[j][map_row(p.a_map, kr + e) * p.lda];
          if constexpr (BNB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ga[AG + j][e] = a2_src[j][map_row(p.a_map, kr + e) * p.lda];
          }
        }
      }
    }
    if ((MT_SPLIT_ABLATE & 32) && kt > 0) return;
    if constexpr (BPL) return;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      if (!B_ALL && !b_on[j]) continue;
      if (k_tail && kt == nk - 1 && b_kh[j]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gb[j][e] = 0.f;
        continue;
      }
      if constexpr (BL == LAYOUT_KCONTIG) {
        const float4 u = *reinterpret_cast<const float4*>(b_src[j] + kt * BK), v = *reinterpret_cast<const float4*>(b_src[j] + kt * BK + 4);
        gb[j][0] = u.x; gb[j][1] = u.y; gb[j][2] = u.z; gb[j][3] = u.w; gb[j][4] = v.x; gb[j][5] = v.y; gb[j][6] = v.z; gb[j][7] = v.w;
      } else {
        const int kr = k0 + b_kh[j] * 8;
        if (p.b_map.gin == 0) {
          const float* s = b_src[j] + (int64_t)kr * p.ldb;
#pragma unroll
          for (int e = 0; e < 8; ++e) gb[j][e] = s[(int64_t)e * p.ldb];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) gb[j][e] = b_src[j][map_row(p.b_map, kr + e) * p.ldb];
        }
      }
    }
  };
  bool first_store = true;
  auto sstore = [&](int stage, const float (&ga_in)[AG2][8], const float (&gb)[BG][8], bool negate_a0 = false, int ktile = 0) {
    if (MT_SPLIT_ABLATE & 2) {
      float z = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int j = 0; j < AG; ++j) z += ga_in[j][e];
#pragma unroll
        for (int j = 0; j < BG; ++j) z += gb[j][e];
      }
      if (z == 123.456f) smem_split[tid] = 1;        // keeps the loads live
      return;
    }
    unsigned char* st = smem_split + stage * A_STAGE;
#pragma unroll
    for (int j = 0; j < AG; ++j) {
      if (!A_ALL && !a_on[j]) continue;
      float ga[1][8];                                  // (indexed [0] below; keeps the non-prologue path a plain copy)
#pragma unroll
      for (int e = 0; e < 8; ++e) ga[0][e] = ga_in[j][e];
      if constexpr (PRO == PRO_BN_SWISH_GATE) {
        const int k = k_begin + ktile * BK + a_kh[j] * 8;
        const float* sc = pv + k;
        const float* sh = pv + p.K + k;
        const float* gt = pv + g_off[j] + k;
#pragma unroll
        for (int e = 0; e < 8; ++e) ga[0][e] = swishf_(fmaf(ga[0][e], sc[e], sh[e])) * gt[e];
      }
      if constexpr (BNB) {
        if (!(k_tail && ktile == nk - 1 && a_kh[j])) {            // (the zero granule of a half-filled last tile stays zero)
          if constexpr (AL == LAYOUT_KCONTIG) {
            const int k = k_begin + ktile * BK + a_kh[j] * 8;
            const float* ka = pv + k;
            const float* kb = pv + p.K + k;
            const float* kc = pv + 2 * p.K + k;
#pragma unroll
            for (int e = 0; e < 8; ++e) ga[0][e] = fmaf(ka[e], ga[0][e], fmaf(kb[e], ga_in[AG + j][e], kc[e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ga[0][e] = fmaf(bn_ka[j], ga[0][e], fmaf(bn_kb[j], ga_in[AG + j][e], bn_kc[j]));
          }
        }
      }
      bf16x8_t x0, x1, x2;
      split_bf16<X6>(reinterpret_cast<const float(&)[4]>(ga[0][0]), reinterpret_cast<const float(&)[4]>(ga[0][4]), x0, x1, x2);
      if (negate_a0) {                                // BAL: the odd tiles' leading A piece goes in negated (see the header comment)
        uint4 u = *reinterpret_cast<uint4*>(&x0);
        u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
        x0 = *reinterpret_cast<bf16x8_t*>(&u);
      }
      *reinterpret_cast<bf16x8_t*>(st + a_dst[j]) = x0;
      *reinterpret_cast<bf16x8_t*>(st + A_PLANE + a_dst[j]) = x1;
      if constexpr (X6) *reinterpret_cast<bf16x8_t*>(st + 2 * A_PLANE + a_dst[j]) = x2;
    }
    if ((MT_SPLIT_ABLATE & 32) && !first_store) return;
    if constexpr (BPL) return;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      if (!B_ALL && !b_on[j]) continue;
      bf16x8_t x0, x1, x2;
      split_bf16<X6>(reinterpret_cast<const float(&)[4]>(gb[j][0]), reinterpret_cast<const float(&)[4]>(gb[j][4]), x0, x1, x2);
      *reinterpret_cast<bf16x8_t*>(st + 3 * A_PLANE + b_dst[j]) = x0;
      *reinterpret_cast<bf16x8_t*>(st + 3 * A_PLANE + B_PLANE + b_dst[j]) = x1;
      if constexpr (X6) *reinterpret_cast<bf16x8_t*>(st + 3 * A_PLANE + 2 * B_PLANE + b_dst[j]) = x2;
    }
  };

  f32x16 acc[TM][TN];
  f32x16 nacc[BAL ? TM : 1][BAL ? TN : 1];           // BAL: minus the sum of the odd tiles' a0-products
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if constexpr (BAL) nacc[i][j][r] = 0.f; }

  // ---- fragment addressing: lane = (row l & 31, k half l >> 5); tile i / j adds 32 rows (same swizzle)
  const int kh = lane >> 5;
  const int a_row = wm * TM * 32 + (lane & 31);
  const int b_row = wn * TN * 32 + (lane & 31);
  const int a_frag = a_row * 32 + ((kh ^ ((a_row >> 3) & 1)) << 4);
  const int b_frag = b_row * 32 + ((kh ^ ((b_row >> 3) & 1)) << 4);

  auto compute = [&](int stage, int kt, auto odd_c, int b_ring = 0) {
    constexpr bool ODD = BAL && decltype(odd_c)::value;
    const unsigned char* st = smem_split + stage * A_STAGE;
    const unsigned char* sb = smem_split + B_BASE + (BPL ? b_ring : stage) * B_STAGE;
    bf16x8_t a0[TM], a1[TM], a2[TM], b0[TN], b1[TN], b2[TN];
    if (MT_SPLIT_ABLATE & 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { a0[i][e] = (__bf16)(float)(lane + e + i); a1[i][e] = (__bf16)(float)(lane - e); a2[i][e] = (__bf16)(float)(lane ^ e); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { b0[j][e] = (__bf16)(float)(lane * 2 + e + j); b1[j][e] = (__bf16)(float)(lane + 3 * e); b2[j][e] = (__bf16)(float)(3 + e); }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a0[i] = *reinterpret_cast<const bf16x8_t*>(st + a_frag + i * 1024);
        a1[i] = *reinterpret_cast<const bf16x8_t*>(st + A_PLANE + a_frag + i * 1024);
        if constexpr (X6) a2[i] = *reinterpret_cast<const bf16x8_t*>(st + 2 * A_PLANE + a_frag + i * 1024);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b0[j] = *reinterpret_cast<const bf16x8_t*>(sb + b_frag + j * 1024);
        b1[j] = *reinterpret_cast<const bf16x8_t*>(sb + B_PLANE + b_frag + j * 1024);
        if constexpr (X6) b2[j] = *reinterpret_cast<const bf16x8_t*>(sb + 2 * B_PLANE + b_frag + j * 1024);
      }
    }
#define MT_TERM(X, Y)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[i], Y[j], acc[i][j], 0, 0, 0);
#define MT_NTERM(X, Y)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      nacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[i], Y[j], nacc[i][j], 0, 0, 0);
    if constexpr (ODD) {                              // a0 holds -x0 here: its products build the negated sum
      if constexpr (X6) { MT_TERM(a2, b0) MT_TERM(a1, b1) MT_NTERM(a0, b2) }
      MT_TERM(a1, b0) MT_NTERM(a0, b1) MT_NTERM(a0, b0)
    } else {
      if constexpr (X6) { MT_TERM(a2, b0) MT_TERM(a1, b1) MT_TERM(a0, b2) }
      MT_TERM(a1, b0) MT_TERM(a0, b1) MT_TERM(a0, b0)
    }
#undef MT_TERM
#undef MT_NTERM
  };
  // scheduling directive for the basic block {fragment reads, MFMAs, split, LDS writes}: VALU_PER MFMA-shadow slots of the split
  // behind every MFMA (an MFMA holds the matrix pipe for 32 cycles; the wave's next 4-5 VALU issues are free in that shadow)
  auto interleave = [&]() {
    constexpr int N_MFMA = (X6 ? 6 : 3) * TM * TN;
    constexpr int N_VALU = (AG + (BPL ? 0 : BG)) * (X6 ? 36 : 22) + (PRO == PRO_BN_SWISH_GATE ? AG * 8 * 14 : 0) + (BNB ? AG * 8 * 2 : 0);
    constexpr int VALU_PER = (N_VALU + N_MFMA - 1) / N_MFMA;
    constexpr int N_VMEM = AG * (AL == LAYOUT_KCONTIG ? 2 : 8) + BG * (BL == LAYOUT_KCONTIG ? 2 : 8);
    (void)N_VMEM;   // pinning the VMEM group first was tried: the compiler then drains the previous step's loads at the top (722 vs 669 us at 4096^3)
#pragma unroll
    for (int q = 0; q < N_MFMA; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);   // then VALU
    }
  };
#define MT_SPLIT_SYNC() do { if (!(MT_SPLIT_ABLATE & 4)) __syncthreads(); } while (0)

  if (nk <= 0) return;
  if constexpr (PIPE == 1) {
    // one register set: tile kt+1's loads are issued at the top of step kt and consumed (split + LDS write) at its end
    float ga[AG2][8], gb[BG][8];
    gload(0, ga, gb); sstore(0, ga, gb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload(kt + 1, ga, gb);
      compute(kt & 1, kt, std::false_type{});
      if (kt + 1 < nk) sstore((kt + 1) & 1, ga, gb);
      MT_SPLIT_SYNC();
    }
  } else if constexpr (BPL) {
    // A through two register sets as below, B planes by DMA two tiles ahead through the three-slot ring -- and NO compiler-visible
    // VMEM operation in the loop: the A loads are inline asm too (aload), so the only wait is the counted one written here.
    // (With the A loads as ordinary C++ loads hipcc inserted its own vmcnt for them, counting only its own loads: that wait also
    // drained the younger DMA of the same step and the variant measured 7-12 % SLOWER than splitting B in the kernel.)
    // VMEM queue at the wait of step kt, oldest first: DMA(kt+1) | A(kt+1) | DMA(kt+2) | A(kt+2); memory operations retire in
    // order, so vmcnt <= BI + 2 AG means tile kt+1's planes are in LDS and its A granules in their registers.
    constexpr int YOUNGER = BI + AG * 2;
    static_assert(A_ALL && AL == LAYOUT_KCONTIG, "asm A loads: every thread stages AG full granules");
    f32x4_t ra0[AG][2], ra1[AG][2];
    float gb_none[BG][8];                           // sstore's B argument is not read with B planes
    auto aload = [&](int kt, f32x4_t (&r)[AG][2]) {
#pragma unroll
      for (int j = 0; j < AG; ++j) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j][0]) : "v"(a_src[j] + kt * BK));
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r[j][1]) : "v"(a_src[j] + kt * BK));
      }
    };
    auto astore = [&](int stage, f32x4_t (&r)[AG][2], bool negate_a0, int ktile) {
      float ga[AG][8];
#pragma unroll
      for (int j = 0; j < AG; ++j) {
        asm volatile("" : "+v"(r[j][0]), "+v"(r[j][1]));       // uses of r stay behind the counted wait that precedes this call
#pragma unroll
        for (int e = 0; e < 4; ++e) { ga[j][e] = r[j][0][e]; ga[j][4 + e] = r[j][1][e]; }
      }
      sstore(stage, ga, gb_none, negate_a0, ktile);
    };
    const int last = nk - 1;
#if MT_SPLIT_A_DEPTH == 3
    // Three A register sets: tile t lives in set t % 3 and is loaded THREE half-steps before it is split (a k-step of this tile
    // is ~0.35 us of matrix time, an L2 / HBM round trip 0.5-1.5 us: one step of prefetch distance is not enough).  The phase of
    // (stage parity, ring slot, register set) repeats every 6 tiles: the loop body is six half-steps with constant indices.
    f32x4_t ra2[AG][2];
    auto rset = [&](auto idx) -> f32x4_t (&)[AG][2] {
      if constexpr (decltype(idx)::value == 0) return ra0; else if constexpr (decltype(idx)::value == 1) return ra1; else return ra2;
    };
    constexpr int YOUNGER3 = BI + AG * 4;          // A(kt+2), DMA(kt+2), A(kt+3) may still be in flight at the wait of step kt
    dma_b(0, 0);
    dma_b(min(1, last), 1);
    aload(0, ra0);
    wait_vmcnt<0>();
    astore(0, ra0, false, 0);
    aload(min(1, last), ra1);
    aload(min(2, last), ra2);
    __syncthreads();
    auto half = [&](int kt, auto ph) {
      constexpr int PH = decltype(ph)::value;                    // kt % 6
      dma_b(min(kt + 2, last), (PH + 2) % 3);
      aload(min(kt + 3, last), rset(std::integral_constant<int, PH % 3>{}));        // (kt + 3) % 3 == kt % 3: the set tile kt was split from
      compute(PH & 1, kt, std::integral_constant<bool, (PH & 1) != 0>{}, PH % 3);
      wait_vmcnt<YOUNGER3>();
      astore((PH + 1) & 1, rset(std::integral_constant<int, (PH + 1) % 3>{}), BAL && ((PH + 1) & 1), min(kt + 1, last));
      interleave();
      MT_SPLIT_SYNC();
    };
    for (int kt = 0; kt < nk; kt += 6) {
      half(kt, std::integral_constant<int, 0>{});
      if (kt + 1 >= nk) break;
      half(kt + 1, std::integral_constant<int, 1>{});
      if (kt + 2 >= nk) break;
      half(kt + 2, std::integral_constant<int, 2>{});
      if (kt + 3 >= nk) break;
      half(kt + 3, std::integral_constant<int, 3>{});
      if (kt + 4 >= nk) break;
      half(kt + 4, std::integral_constant<int, 4>{});
      if (kt + 5 >= nk) break;
      half(kt + 5, std::integral_constant<int, 5>{});
    }
#else
    dma_b(0, 0);
    dma_b(min(1, last), 1);
    aload(0, ra0);
    wait_vmcnt<0>();
    astore(0, ra0, false, 0);
    aload(min(1, last), ra0);                       // stays in flight across the barrier
    __syncthreads();
    int ring = 0;                                   // ring slot of tile kt
    for (int kt = 0; kt < nk; kt += 2) {
      int r2 = ring + 2; r2 = r2 >= 3 ? r2 - 3 : r2;
      dma_b(min(kt + 2, last), r2);
      aload(min(kt + 2, last), ra1);
      compute(0, kt, std::false_type{}, ring);
      wait_vmcnt<YOUNGER>();
      astore(1, ra0, BAL, min(kt + 1, last));
      interleave();
      MT_SPLIT_SYNC();
      if (kt + 1 >= nk) break;
      ring = ring + 1 >= 3 ? 0 : ring + 1;
      r2 = ring + 2; r2 = r2 >= 3 ? r2 - 3 : r2;
      dma_b(min(kt + 3, last), r2);
      aload(min(kt + 3, last), ra0);
      compute(1, kt + 1, std::true_type{}, ring);
      wait_vmcnt<YOUNGER>();
      astore(0, ra1, false, min(kt + 2, last));
      interleave();
      MT_SPLIT_SYNC();
      ring = ring + 1 >= 3 ? 0 : ring + 1;
    }
#endif
    wait_vmcnt<0>();                                // the clamped re-loads of the last tile must not outlive their registers
  } else {
    // two register sets: tile kt+2's loads are issued at the top of step kt; tile kt+1 (loaded a full step ago) is split and
    // written to the other LDS stage in the shadow of tile kt's MFMAs.  Branch-free body (the scheduler interleaves the VALU
    // split with the MFMA stream inside one basic block): past the end the loads re-read the last tile and the stores fill
    // a stage nobody reads.
    float ga0[AG2][8], gb0[BG][8], ga1[AG2][8], gb1[BG][8];
    const int last = nk - 1;
    gload(0, ga0, gb0); sstore(0, ga0, gb0, false, 0);
    if (MT_SPLIT_ABLATE & 32) { sstore(1, ga0, gb0); first_store = false; }
    gload(min(1, last), ga0, gb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      gload(min(kt + 2, last), ga1, gb1);
      compute(0, kt, std::false_type{});
      sstore(1, ga0, gb0, BAL, min(kt + 1, last));
      interleave();
      MT_SPLIT_SYNC();
      if (kt + 1 >= nk) break;
      gload(min(kt + 3, last), ga0, gb0);
      compute(1, kt + 1, std::true_type{});
      sstore(0, ga1, gb1, false, min(kt + 2, last));
      interleave();
      MT_SPLIT_SYNC();
    }
  }

  if constexpr (BAL) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] -= nacc[i][j][r];
    // finish the subtraction here (nacc dead) instead of letting it sink into the epilogue's branches next to their address registers
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
  }
  if (MT_SPLIT_ABLATE & 16) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 123.456f) p.C[0] = sacc;
    return;
  }
  // the epilogue's address arithmetic depends only on the tile coordinates: without this fence the compiler computes it before
  // the main loop and carries it through (60-100 VGPRs: spills in the balanced variants)
  int m0e = m0, n0e = n0, lane_e = lane;
  asm volatile("" : "+s"(m0e), "+s"(n0e), "+v"(lane_e));
  gemm_epilogue<TM, TN, EPI>(p, acc, m0e, n0e, wm, wn, lane_e, split_idx);
}

}  // namespace mt
