// Plane-operand GEMM main loop for gfx950: fp32 semantics on the bf16 matrix pipe with BOTH operands arriving pre-split.
//
// gemm_split.hpp splits every fp32 operand value exactly into three bf16 pieces INSIDE the GEMM (global fp32 -> VGPRs -> ~4.6 VALU
// per element -> three ds_write_b128 per granule) and does so once per tile that reads the element: every activation and gradient
// of the TimeSformer is split again by every output-tile column that consumes it and a second time by its weight-gradient GEMM.
// Here the PRODUCER of a tensor (LayerNorm, the attention kernels, the GEGLU epilogues, the LayerNorm backward; mt_split_planes_blk
// for the weights) writes the three planes  x = x0 + x1 + x2  once, 6 B per element, and every consumer -- forward, data-gradient
// AND weight-gradient GEMM -- moves them global -> LDS by DMA (global_load_lds_dwordx4): no VALU, no staging VGPRs and no ds_write
// in any MFMA wavefront.  The arithmetic is the one of gemm_split.hpp: six piece products of weight >= 2^-18 per fp32 product,
// fp32 accumulators.
//
// PLANE LAYOUT ("blk"): planes[3][R/32][C/16][32][16] bf16 -- 1 KB blocks of 32 rows x 16 columns, R padded to 32 and C to 16 with
// zeros.  A row-major plane serves a 16-k step of a k-contiguous operand with 32 B per row (a quarter of each 128-byte line; the
// other three quarters come back from L2 for the next three k-steps: measured 160 vs 191 TF-eq for the in-kernel split on 4096^3),
// the blocked layout serves it with whole 1 KB blocks -- and the SAME tensor serves the other contraction direction in 128-byte runs:
//   * k-contiguous use (contraction along the columns: forward GEMMs, the A side of data gradients).  LDS image per plane
//     [rows][16 k] (32 B per row, 16-byte slot ^= (row >> 3) & 1); one DMA piece = one block; fragment = one ds_read_b128.
//   * k-major use (contraction along the ROWS: both operands of a weight gradient dW = dY^T X, and the weight of a data gradient
//     dX = dY W -- no transposed weight copies).  LDS image per plane [16 k][128 cols] (256 B per k-row, 32-byte segment
//     ^= 2 (k & 3)); one DMA piece = 4 k-rows x 8 column blocks; fragment = two ds_read_b64_tr_b16: the LDS transpose-read hands
//     lane i of a 16-lane group column i of a [4 k][16 col] block, i.e. four consecutive k of one output row -- the 32x32x16
//     operand layout without any register transposition; the eight (k, segment) pairs a 32-lane half touches cover all 64 banks.
//
// Pipeline: a ring of STAGES LDS stages of one 16-k step each; tile kt + STAGES - 1 is issued right behind the barrier of step
// kt, the wait is a counted vmcnt (the younger tiles stay in flight across the barrier).  The DMA uses the saddr form: one
// uniform 64-bit base per operand and step (scalar arithmetic), one constant 32-bit offset VGPR per 1 KB piece.
//
// Rounding bias of the bf16 pipe (gemm_split.hpp: BAL).  v_mfma_f32_32x32x16_bf16 truncates its aligned addends: a
// sign-INDEPENDENT bias of about -3e-9 x sum|a||b| per instruction that adds up coherently through a deep network.
//   BAL_PAIR   two accumulators: odd k-steps feed -x0 (negated in registers here: 4 v_xor per tile) into a second accumulator
//              and the result is acc - nacc.  Bit-identical to gemm_split.hpp's loop; 16 x TM x TN more registers.
//   BAL_PHASE  one accumulator, sign phases + - - + over the block's k-range: during the middle half every A fragment is negated
//              (12 v_xor per step) and the accumulator holds the NEGATED sum (sign flipped at the two phase boundaries, exact).
//              The bias enters with opposite signs in the two halves; the + - - + order also cancels its linear growth with |C|.
#pragma once
#include "gemm_split.hpp"
#include "planes.hpp"

#ifndef MT_PLANES_ABLATE      // tuning lab only: 1 no DMA, 2 no barrier, 4 no fragment reads, 8 no epilogue, 16 no MFMA
#define MT_PLANES_ABLATE 0
#endif
#ifndef MT_PLANES_EPI_ABLATE  // tuning lab only (GEGLU-backward plane epilogue): 1 no pre-activation loads, 2 no gelu math, 4 no LDS transposition + plane stores,
#define MT_PLANES_EPI_ABLATE 0 // 8 no plane stores (transposition kept), 16 no column-sum atomics
#endif
#ifndef MT_PLANES_PRIO        // tuning lab only: 1 = static priority 1 for waves in odd hardware wave slots (breaks the lockstep of the
#define MT_PLANES_PRIO 0      // two co-resident blocks' waves on a SIMD), 2 = for odd blocks of 256 in launch order
#endif

namespace mt {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
enum : int { BAL_NONE = 0, BAL_PAIR = 1, BAL_PHASE = 2 };

// one 1 KB piece: lane l's 16 bytes from (sbase + voff) land at lds_byte_addr + 16 l
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// ---- epilogues that emit the result as blocked planes (the GEGLU pair: h feeds FF2 and its weight gradient, du feeds FF1's data and
// weight gradients -- neither is read as fp32 by anything).  A lane of the 32x32 accumulator tile owns ONE column and 16 rows,
// the plane format wants 8 consecutive columns of one row per 16-byte store: the tile goes through a per-wave [32][36] fp32 LDS
// patch (main-loop LDS, free after the loop), comes back row-wise, is split once and leaves as whole 1 KB blocks.
__device__ __forceinline__ void planes_emit_tile(const float (&v)[16], float* wl, const PlaneRef& o, int row0, int col0, int M, int lane) {
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + row_h) * 36 + col_l] = v[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS stores are complete (no other wave touches this patch)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int g = pass * 64 + lane, row = g >> 2, cg = g & 3;
    const float4 lo = *reinterpret_cast<const float4*>(wl + row * 36 + cg * 8);
    const float4 hi = *reinterpret_cast<const float4*>(wl + row * 36 + cg * 8 + 4);
    const int rg = row0 + row, cgl = col0 + cg * 8;
    const bool live = rg < M;                               // padding rows of the last row block are written as zeros
    const float x[8] = {live ? lo.x : 0.f, live ? lo.y : 0.f, live ? lo.z : 0.f, live ? lo.w : 0.f,
                        live ? hi.x : 0.f, live ? hi.y : 0.f, live ? hi.z : 0.f, live ? hi.w : 0.f};
    if (rg < o.rows_pad && cgl < o.cb16 * 16) { if (MT_PLANES_EPI_ABLATE & 8) { if (x[0] + x[7] == 123.456f) o.p[0] = (__bf16)1.f; } else planes_store8(o, rg, cgl, x); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next tile overwrites the patch
}

template <int TM, int TN, int EPI>
__device__ __forceinline__ void planes_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, float* wl) {
  static_assert(EPI == EPI_GEGLU || EPI == EPI_GEGLU_BWD, "plane output: the GEGLU pair");
  const PlaneRef o{reinterpret_cast<__bf16*>(p.c_planes), p.c_pstride, (int)p.ldcp, (p.M + 31) & ~31};
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  const int mw = m0 + wm * TM * 32;
  if constexpr (EPI == EPI_GEGLU) {
    const int j = (n0 >> 1) + wn * 32 + col_l;
    const bool jok = j < p.n_half;
    const float ba = (jok && p.bias) ? p.bias[j] : 0.f;
    const float bg = (jok && p.bias) ? p.bias[p.n_half + j] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float hv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + i * 32 + row_h + (r & 3) + 8 * (r >> 2);
        const float a = acc[i][0][r] + ba, g = acc[i][1][r] + bg;
        hv[r] = a * gelu_erf(g);
        if (m < p.M && jok) {
          if (p.C2) *reinterpret_cast<float2*>(p.C2 + (int64_t)m * p.ldc2 + 2 * j) = make_float2(a, g);
          if (p.C) p.C[(int64_t)m * p.ldc + j] = hv[r];
        }
      }
      planes_emit_tile(hv, wl, o, mw + i * 32, (n0 >> 1) + wn * 32, p.M, lane);
    }
  } else {
    // Two phases, because loads and stores share the wave's vmcnt queue (gfx9 counts store acknowledgements there too): a tile-by-tile
    // "load pre-activations, compute, store planes" sequence waits for the PREVIOUS tile's stores to be acknowledged before it sees
    // its own loads (measured: 148 of 328 us per launch went to the 206 MB of pre-activation loads, 1.4 TB/s).  Phase 1 loads every
    // pre-activation of the wave's tiles (one MFMA tile per batch) and turns the accumulators into both gradients in place -- da
    // over acc, dg into `dgs` (the registers of the dead second accumulator); phase 2 only transposes through LDS and stores.
    f32x16 dgs[TM][TN];
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + col_l;
      const bool nok = n < p.N;
      s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float2 ag[16];                                      // one MFMA tile's pre-activations per batch: 32 registers next to acc + dgs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + row_h + (r & 3) + 8 * (r >> 2);
          ag[r] = ((MT_PLANES_EPI_ABLATE & 1) || !(m < p.M && nok)) ? make_float2(0.3f, 0.1f)
                                                                     : *reinterpret_cast<const float2*>(p.C2 + (int64_t)m * p.ldc2 + 2 * n);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + i * 32 + row_h + (r & 3) + 8 * (r >> 2);
          const float v = acc[i][j][r];
          float gl, gr;
          if (MT_PLANES_EPI_ABLATE & 2) { gl = ag[r].y; gr = ag[r].y + 1.f; } else gelu_erf_both(ag[r].y, gl, gr);
          const bool live = m < p.M && nok;
          const float da = live ? v * gl : 0.f, dg = live ? v * ag[r].x * gr : 0.f;
          acc[i][j][r] = da;
          dgs[i][j][r] = dg;
          s1[j] += da; s2[j] += dg;
        }
        __builtin_amdgcn_sched_barrier(0);                  // the next tile's 16 loads stay behind this tile's arithmetic (registers)
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + col_l;
      const bool nok = n < p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float da[16], dg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          da[r] = acc[i][j][r]; dg[r] = dgs[i][j][r];
          const int m = mw + i * 32 + row_h + (r & 3) + 8 * (r >> 2);
          if (p.C && m < p.M && nok) { p.C[(int64_t)m * p.ldc + n] = da[r]; p.C[(int64_t)m * p.ldc + p.n_half + n] = dg[r]; }
        }
        if (MT_PLANES_EPI_ABLATE & 4) { if (da[3] + dg[5] == 123.456f) p.col_sum[0] = 1.f; }
        else if (n0 + wn * TN * 32 + j * 32 < p.N) {
          // (a 32-column tile past N = n_half belongs to nobody: with n_half % 128 != 0 the last block tile's upper waves would
          // otherwise write zeros over dg columns [n_half, ...) that another block owns; n_half % 32 == 0, so tiles are whole)
          planes_emit_tile(da, wl, o, mw + i * 32, n0 + wn * TN * 32 + j * 32, p.M, lane);
          planes_emit_tile(dg, wl, o, mw + i * 32, p.n_half + n0 + wn * TN * 32 + j * 32, p.M, lane);
        }
      }
      if (p.col_sum && !(MT_PLANES_EPI_ABLATE & 16)) {     // bias gradient of the first feed-forward Linear: column sums of du
        float t1 = s1[j], t2 = s2[j];
        t1 += __shfl_xor(t1, 32);
        t2 += __shfl_xor(t2, 32);
        if (lane < 32 && nok) {
          if (p.det.vals) {                 // deterministic mode: group = 32-column block, rank = 32-row block of the wave's first row (det.hpp)
            const int rk = mw >> 5;
            det_put(p.det, n >> 5, rk, n & 31, t1);
            det_put(p.det, (p.n_half + n) >> 5, rk, n & 31, t2);
            if (rk == 0 && (n & 31) == 0) { det_base(p.det, n >> 5, n); det_base(p.det, (p.n_half + n) >> 5, p.n_half + n); }
          } else {
            atomicAdd(p.col_sum + n, t1);
            atomicAdd(p.col_sum + p.n_half + n, t2);
          }
        }
      }
    }
  }
}

// ---- stream-K (SK): a persistent grid splits the linearised (tile, k-step) space evenly instead of handing out whole tiles.
// 396 tiles of a 12576 x 512 output fill 77 % of 2 x 256 resident blocks and 1188 tiles of the QKV layer take 3 rounds for 2.3
// rounds of work; here every block gets total / grid k-steps: a contiguous run of the XCD-ordered tile list -- the tail of one tile
// (stored as an fp32 slab + flag for the block that owns that tile), whole tiles, and the head of a last tile, which it OWNS: it adds
// the slabs of the blocks that follow it (they produced them at the START of their runs) in a fixed order and runs the epilogue.
// Deterministic; a block only ever waits for work its successors do first, so any residency >= 3 blocks makes progress.
// Hand-off = the agent-scope release / acquire recipe of the CDNA4 guide (plain slab stores, every wave drains, barrier, lane 0
// release fence + drain, relaxed flag store | relaxed poll, ONE acquire fence, barrier, plain loads); the owner clears the flag.
template <int BM, int BN>
__device__ __forceinline__ void sk_tile_coords(const GemmArgs& p, int t, int& mt_, int& nt_) {
  const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
  if (p.group_n > 0) {
    // the L2-blocked order of tile_coords(): XCD x sweeps its band of tile rows one column group at a time; bands are listed one
    // after the other, so a block's contiguous run stays inside one band (except at a band seam)
    const int q = m_tiles >> 3, r = m_tiles & 7;
    int xcd = 0, base = 0;
    for (; xcd < 7; ++xcd) {
      const int cnt = (q + (xcd < r ? 1 : 0)) * n_tiles;
      if (t < base + cnt) break;
      base += cnt;
    }
    const int rows = q + (xcd < r ? 1 : 0);
    const int row0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int idx = t - base;
    const int per_group = rows * p.group_n;
    const int g = idx / per_group;
    const int rem = idx - g * per_group;
    const int groups = (n_tiles + p.group_n - 1) / p.group_n;
    const int gw = (g == groups - 1) ? n_tiles - g * p.group_n : p.group_n;
    // (the last group is narrower: its per-group count is rows * gw, and it is the last one, so idx / per_group is still right)
    const int ml = rem / gw;
    mt_ = row0 + ml;
    nt_ = g * p.group_n + (rem - ml * gw);
  } else {
    mt_ = t / n_tiles;
    nt_ = t - mt_ * n_tiles;
  }
}

// GemmArgs for this loop: a_planes / b_planes with a_pstride / b_pstride (elements between planes), lda / ldb = the operand's
// number of 16-column blocks (C / 16); M, N, K logical; rows past the end are clamped to the last row block (whose padding rows are
// zeros).
// SKM: 0 = one block per tile, 1 = stream-K (below), 2 = persistent blocks that walk their XCD's share of the tile list and issue
// the NEXT tile's first stage before the current tile's epilogue (the per-tile pipeline fill that K = 512 problems pay on every tile)
template <int WAVES_M, int WAVES_N, int TM, int TN, bool AKM, bool BKM, int EPI, int STAGES, int MINW, int BAL, bool CPL = false, int SKM = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINW)
void gemm_planes_kernel(const GemmArgs p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int BK = 16;
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;            // bytes per plane per stage
  constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
  constexpr int PPA = A_PLANE / 1024, PPB = B_PLANE / 1024;       // 1 KB pieces per plane
  constexpr int PA = 3 * PPA, PB = 3 * PPB;
  static_assert((PA + PB) % NW == 0, "every wavefront must issue the same number of DMA pieces");
  constexpr int IPW = (PA + PB) / NW;
  static_assert(STAGES >= 2 && STAGES <= 4, "2..4 stages");
  static_assert((!AKM || BM == 128) && (!BKM || BN == 128), "k-major images are laid out for 128-column tiles");
  constexpr bool SK = SKM == 1, PS = SKM == 2;
  static_assert(!SK || !AKM, "stream-K: forward / data-gradient forms (weight gradients split K over the grid already)");
  static_assert(!PS || (!AKM && !CPL && STAGES == 2 && EPI != EPI_GEGLU && BAL != BAL_PHASE), "persistent form: NT / NN, fp32 output, ring of two");
  constexpr bool PAIR = BAL == BAL_PAIR, PHASE = BAL == BAL_PHASE;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_pl[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_pl;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  const __bf16* a_base = reinterpret_cast<const __bf16*>(p.a_planes);
  const __bf16* b_base = reinterpret_cast<const __bf16*>(p.b_planes);

  // ---- fragment addressing (byte offsets inside a plane image; independent of the tile)
  int a_frag[TM], b_frag[TN];
  {
    const int kh = lane >> 5;
    const int g = lane >> 4, i16 = lane & 15;
    const int ktr = 8 * (g >> 1) + (i16 >> 2);       // transpose-read: 16-lane group g reads the [4 k][16 col] block k = 8 (g >> 1) + 4 r ..
    const int swz = 2 * (i16 >> 2);                  // (r = 0, 1: the second read sits 4 k-rows = 1024 B further), columns = segment S + (g & 1);
#pragma unroll                                       // lane i supplies k-row i >> 2, 4 columns from (i & 3) * 4, and receives column i
    for (int i = 0; i < TM; ++i) {
      if constexpr (!AKM) {
        const int row = wm * TM * 32 + i * 32 + (lane & 31);
        a_frag[i] = row * 32 + ((kh ^ ((row >> 3) & 1)) << 4);
      } else {
        const int seg = (wm * TM * 32 + i * 32) / 16 + (g & 1);
        a_frag[i] = ktr * 256 + ((seg ^ swz) * 32) + (i16 & 3) * 8;
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (!BKM) {
        const int row = wn * TN * 32 + j * 32 + (lane & 31);
        b_frag[j] = row * 32 + ((kh ^ ((row >> 3) & 1)) << 4);
      } else {
        const int seg = (wn * TN * 32 + j * 32) / 16 + (g & 1);
        b_frag[j] = ktr * 256 + ((seg ^ swz) * 32) + (i16 & 3) * 8;
      }
    }
  }

  struct Frags { bf16x8_t a0[TM], a1[TM], a2[TM], b0[TN], b1[TN], b2[TN]; };
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_p;
  typedef __attribute__((address_space(3))) const unsigned char* lds_cp;
  auto tr_read = [&](const unsigned char* base) -> bf16x8_t {
    lds_cp a = (lds_cp)base;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)a);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(a + 1024));
    union { s16x4_t h[2]; bf16x8_t v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
  };

  auto read_frags = [&](int slot, Frags& f) {
    const unsigned char* sa = smem_pl + slot * STAGE;
    const unsigned char* sb = sa + 3 * A_PLANE;
    if (MT_PLANES_ABLATE & 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { f.a0[i][e] = (__bf16)(float)(lane + e + i); f.a1[i][e] = (__bf16)(float)(lane - e); f.a2[i][e] = (__bf16)(float)(lane ^ e); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { f.b0[j][e] = (__bf16)(float)(lane * 2 + e + j); f.b1[j][e] = (__bf16)(float)(lane + 3 * e); f.b2[j][e] = (__bf16)(float)(3 + e); }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (!AKM) {
        f.a0[i] = *reinterpret_cast<const bf16x8_t*>(sa + a_frag[i]);
        f.a1[i] = *reinterpret_cast<const bf16x8_t*>(sa + A_PLANE + a_frag[i]);
        f.a2[i] = *reinterpret_cast<const bf16x8_t*>(sa + 2 * A_PLANE + a_frag[i]);
      } else {
        f.a0[i] = tr_read(sa + a_frag[i]);
        f.a1[i] = tr_read(sa + A_PLANE + a_frag[i]);
        f.a2[i] = tr_read(sa + 2 * A_PLANE + a_frag[i]);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (!BKM) {
        f.b0[j] = *reinterpret_cast<const bf16x8_t*>(sb + b_frag[j]);
        f.b1[j] = *reinterpret_cast<const bf16x8_t*>(sb + B_PLANE + b_frag[j]);
        f.b2[j] = *reinterpret_cast<const bf16x8_t*>(sb + 2 * B_PLANE + b_frag[j]);
      } else {
        f.b0[j] = tr_read(sb + b_frag[j]);
        f.b1[j] = tr_read(sb + B_PLANE + b_frag[j]);
        f.b2[j] = tr_read(sb + 2 * B_PLANE + b_frag[j]);
      }
    }
  };

  auto negate = [&](bf16x8_t& v, unsigned mask) {
    uint4 u = *reinterpret_cast<uint4*>(&v);
    u.x ^= mask; u.y ^= mask; u.z ^= mask; u.w ^= mask;
    v = *reinterpret_cast<bf16x8_t*>(&u);
  };

  f32x16 acc[TM][TN];
  f32x16 nacc[PAIR ? TM : 1][PAIR ? TN : 1];

  auto mma = [&](Frags& f, auto odd_c, unsigned phase_mask) {
    constexpr bool ODD = PAIR && decltype(odd_c)::value;
    if (MT_PLANES_ABLATE & 16) {                     // keep the fragments live without the matrix pipe
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) z += (float)f.a0[i][0] + (float)f.a1[i][1] + (float)f.a2[i][2];
#pragma unroll
      for (int j = 0; j < TN; ++j) z += (float)f.b0[j][0] + (float)f.b1[j][1] + (float)f.b2[j][2];
      acc[0][0][0] += z;
      return;
    }
#define MT_TERM(X, Y)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.X[i], f.Y[j], acc[i][j], 0, 0, 0);
#define MT_NTERM(X, Y)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      nacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.X[i], f.Y[j], nacc[i][j], 0, 0, 0);
    if constexpr (PHASE) {
#pragma unroll
      for (int i = 0; i < TM; ++i) { negate(f.a0[i], phase_mask); negate(f.a1[i], phase_mask); negate(f.a2[i], phase_mask); }
    }
    if constexpr (ODD) {                              // odd k-steps feed -x0: their a0 products build the negated sum (gemm_split.hpp: BAL)
#pragma unroll
      for (int i = 0; i < TM; ++i) negate(f.a0[i], 0x80008000u);
      MT_TERM(a2, b0) MT_TERM(a1, b1) MT_NTERM(a0, b2)
      MT_TERM(a1, b0) MT_NTERM(a0, b1) MT_NTERM(a0, b0)
    } else {
      MT_TERM(a2, b0) MT_TERM(a1, b1) MT_TERM(a0, b2)
      MT_TERM(a1, b0) MT_TERM(a0, b1) MT_TERM(a0, b0)
    }
#undef MT_TERM
#undef MT_NTERM
  };
  auto flip_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
  };
#define MT_PL_BARRIER() do { if (!(MT_PLANES_ABLATE & 2)) __builtin_amdgcn_s_barrier(); } while (0)

  // ---- one piece of work: k-steps [kt_lo, kt_lo + nk) (absolute step index counted from k_begin) of tile (mt_, nt_) into acc
  auto run_piece = [&](int mt_, int nt_, int k_begin, int kt_lo, int nk) {
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    // DMA sources.  Piece q of a stage (q = wave * IPW + j): q < PA -> A plane q / PPA, else B plane; inside a plane the piece
    // index selects a 32-row block (k-contiguous) or 4 k-rows (k-major).  Per piece a constant 32-bit byte offset from a uniform
    // base that depends on the operand and the k-step only.
    unsigned voff[IPW];
    bool is_a[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int q = wave * IPW + j;
      is_a[j] = q < PA;
      const int qq = is_a[j] ? q : q - PA;
      const int plane = is_a[j] ? qq / PPA : qq / PPB;
      const int piece = is_a[j] ? qq % PPA : qq % PPB;
      const int64_t pstride = is_a[j] ? p.a_pstride : p.b_pstride;
      const int64_t cb16 = is_a[j] ? p.lda : p.ldb;
      const bool kmaj = is_a[j] ? AKM : BKM;
      int64_t off;                                     // elements, without the k-step term
      if (!kmaj) {
        // rows = output rows, 16 k = one column block: the piece is row block (g0 / 32 + piece), the lane permutes inside it
        const int row = lane >> 1, slot = lane & 1;
        const int kh = slot ^ ((row >> 3) & 1);
        int g0;                                        // first global row of the piece's 32 tile rows
        int lim;
        if (is_a[j]) { g0 = m0 + piece * 32; lim = p.M; }
        else if constexpr (EPI == EPI_GEGLU) {         // tile row -> weight row: 'a' and 'gate' halves interleaved per 32 columns
          static_assert(EPI != EPI_GEGLU || TN == 2, "GEGLU wants TN == 2");
          const int w = piece >> 1, sel = piece & 1;
          g0 = sel * p.n_half + (n0 >> 1) + w * 32; lim = 2 * p.n_half;
          if ((n0 >> 1) + w * 32 >= p.n_half) g0 = 0;
        } else { g0 = n0 + piece * 32; lim = p.N; }
        if (g0 >= lim) g0 = (lim - 1) & ~31;           // row blocks past the end are never stored: any valid block will do
        off = plane * pstride + (int64_t)(g0 >> 5) * cb16 * 512 + row * 16 + kh * 8;
      } else {
        // rows = k, 128 columns = 8 column blocks: 4 k-rows per piece
        const int kl = piece * 4 + (lane >> 4), s16 = lane & 15;
        const int seg = (s16 >> 1) ^ (2 * (kl & 3)), half = s16 & 1;
        int cb = ((is_a[j] ? m0 : n0) >> 4) + seg;
        const int cbmax = (int)cb16 - 1;
        cb = cb < cbmax ? cb : cbmax;                  // column blocks past the end are never stored
        off = plane * pstride + (int64_t)cb * 512 + kl * 16 + half * 8;
      }
      voff[j] = (unsigned)(off * 2);
    }

    auto issue = [&](int kt, int slot) {               // DMA of k-tile kt (counted from kt_lo) into ring slot `slot`
      if (MT_PLANES_ABLATE & 1) return;
      const int kb = k_begin + (kt_lo + kt) * BK;
      // k-contiguous: column block kb / 16.  k-major: row block kb / 32 (all of its column blocks), row kb % 32 inside it.
      const __bf16* as = a_base + (AKM ? ((int64_t)(kb >> 5) * p.lda * 512 + (kb & 31) * 16) : (int64_t)(kb >> 4) * 512);
      const __bf16* bs = b_base + (BKM ? ((int64_t)(kb >> 5) * p.ldb * 512 + (kb & 31) * 16) : (int64_t)(kb >> 4) * 512);
      const unsigned dst = lds_base + (unsigned)(slot * STAGE);
#pragma unroll
      for (int j = 0; j < IPW; ++j)
        lds_dma16_s(is_a[j] ? as : bs, voff[j], dst + (unsigned)((wave * IPW + j) * 1024));
    };

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if constexpr (PAIR) nacc[i][j][r] = 0.f; }

    // sign phases + - - + : negated operands (and a negated accumulator) for k-steps [q1, q2)
    const int q1 = PHASE ? nk >> 2 : nk, q2 = PHASE ? nk - (nk >> 2) : nk;

    // tiles kt+1 .. kt+STAGES-1 in flight while tile kt is multiplied
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) issue(s, s);
    int slot = 0;                                      // ring slot of tile kt
    auto step = [&](int kt, auto odd_c) {
      const int later = min(STAGES - 2, nk - 1 - kt);
      if (later >= 2) wait_vmcnt<2 * IPW>();
      else if (later == 1) wait_vmcnt<IPW>();
      else wait_vmcnt<0>();
      MT_PL_BARRIER();                                 // every wave's share of tile kt is visible; the slot of tile kt-1 is free
      int nslot = slot + STAGES - 1; nslot = nslot >= STAGES ? nslot - STAGES : nslot;
      if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, nslot);
      Frags f;
      read_frags(slot, f);
      unsigned mask = 0;
      if constexpr (PHASE) {
        if (kt == q1 || kt == q2) flip_acc();
        mask = (kt >= q1 && kt < q2) ? 0x80008000u : 0u;
      }
      mma(f, odd_c, mask);
      slot = slot + 1 >= STAGES ? 0 : slot + 1;
    };
    // the accumulator pair alternates with the ABSOLUTE k-step: pieces start on even steps (stream-K hands out pairs of k-steps)
    for (int kt = 0; kt < nk; kt += 2) {
      step(kt, std::false_type{});
      if (kt + 1 >= nk) break;
      step(kt + 1, std::true_type{});
    }
    if constexpr (PHASE) {
      if (q2 >= nk && q1 < nk) flip_acc();             // (only when the last phase is empty: nk < 4)
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] -= nacc[i][j][r];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
  };

  auto epilogue = [&](int mt_, int nt_, int split_idx = -1) {
    if (MT_PLANES_ABLATE & 8) {
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
    int m0e = mt_ * BM, n0e = nt_ * BN, lane_e = lane;
    asm volatile("" : "+s"(m0e), "+s"(n0e), "+v"(lane_e));
    if constexpr (CPL) {
      static_assert(!CPL || STAGES * STAGE >= NW * 32 * 36 * 4, "the plane epilogue's LDS patches live in the main loop's stages");
      __syncthreads();                                  // every wave has finished reading the stages
      planes_epilogue<TM, TN, EPI>(p, acc, m0e, n0e, wm, wn, lane_e, reinterpret_cast<float*>(smem_pl) + wave * (32 * 36));
    } else {
      gemm_epilogue<TM, TN, EPI>(p, acc, m0e, n0e, wm, wn, lane_e, split_idx);
    }
  };

  if (p.wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.wave_prio == 3) __builtin_amdgcn_s_setprio(3);
  if (MT_PLANES_PRIO == 1) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));     // HW_ID[3:0] = wave slot on the SIMD
    if (__builtin_amdgcn_readfirstlane(hwid) & 1) __builtin_amdgcn_s_setprio(1);
  } else if (MT_PLANES_PRIO == 2) {
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
  }

  if constexpr (PS) {
    // ---- persistent: block (x, q) of XCD x = blockIdx % 8 takes tiles q, q + Q, ... of that XCD's contiguous share of the tile list
    // (the L2-blocked order of tile_coords() when group_n > 0), Q = gridDim / 8 blocks per XCD
    const int Q = gridDim.x >> 3, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int n_tiles = (p.N + BN - 1) / BN, m_tiles = (p.M + BM - 1) / BM;
    int base = 0, cnt;
    if (p.group_n > 0) {
      const int qq = m_tiles >> 3, rr = m_tiles & 7;
      for (int i = 0; i < xcd; ++i) base += (qq + (i < rr ? 1 : 0)) * n_tiles;
      cnt = (qq + (xcd < rr ? 1 : 0)) * n_tiles;
    } else {
      const int T = m_tiles * n_tiles, per = T >> 3, rem = T & 7;
      base = xcd * per + min(xcd, rem);
      cnt = per + (xcd < rem ? 1 : 0);
    }
    if (q >= cnt) return;
    const int nk = (p.K + BK - 1) / BK;
    // DMA pieces of this wave: which operand / plane / piece is tile-independent, the byte offsets are per tile
    bool is_a[IPW];
    int pl_[IPW], pc_[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      const int qj = wave * IPW + j;
      is_a[j] = qj < PA;
      const int qx = is_a[j] ? qj : qj - PA;
      pl_[j] = is_a[j] ? qx / PPA : qx / PPB;
      pc_[j] = is_a[j] ? qx % PPA : qx % PPB;
    }
    auto setup = [&](int mt_, int nt_, unsigned (&voff)[IPW]) {
      const int m0 = mt_ * BM, n0 = nt_ * BN;
#pragma unroll
      for (int j = 0; j < IPW; ++j) {
        const int64_t pstride = is_a[j] ? p.a_pstride : p.b_pstride;
        const int64_t cb16 = is_a[j] ? p.lda : p.ldb;
        const bool kmaj = is_a[j] ? AKM : BKM;
        int64_t off;
        if (!kmaj) {
          const int row = lane >> 1, sl = lane & 1;
          const int kh = sl ^ ((row >> 3) & 1);
          int g0 = (is_a[j] ? m0 : n0) + pc_[j] * 32;
          const int lim = is_a[j] ? p.M : p.N;
          if (g0 >= lim) g0 = (lim - 1) & ~31;
          off = pl_[j] * pstride + (int64_t)(g0 >> 5) * cb16 * 512 + row * 16 + kh * 8;
        } else {
          const int kl = pc_[j] * 4 + (lane >> 4), s16 = lane & 15;
          const int seg = (s16 >> 1) ^ (2 * (kl & 3)), half = s16 & 1;
          int cb = ((is_a[j] ? m0 : n0) >> 4) + seg;
          const int cbmax = (int)cb16 - 1;
          cb = cb < cbmax ? cb : cbmax;
          off = pl_[j] * pstride + (int64_t)cb * 512 + kl * 16 + half * 8;
        }
        voff[j] = (unsigned)(off * 2);
      }
    };
    auto issue = [&](const unsigned (&voff)[IPW], int kt, int slot_) {
      if (MT_PLANES_ABLATE & 1) return;
      const int kb = kt * BK;
      const __bf16* as = a_base + (AKM ? ((int64_t)(kb >> 5) * p.lda * 512 + (kb & 31) * 16) : (int64_t)(kb >> 4) * 512);
      const __bf16* bs = b_base + (BKM ? ((int64_t)(kb >> 5) * p.ldb * 512 + (kb & 31) * 16) : (int64_t)(kb >> 4) * 512);
      const unsigned dst = lds_base + (unsigned)(slot_ * STAGE);
#pragma unroll
      for (int j = 0; j < IPW; ++j) lds_dma16_s(is_a[j] ? as : bs, voff[j], dst + (unsigned)((wave * IPW + j) * 1024));
    };
    int i = q, mt_, nt_;
    sk_tile_coords<BM, BN>(p, base + i, mt_, nt_);
    unsigned vc[IPW], vn[IPW];
    setup(mt_, nt_, vc);
    issue(vc, 0, 0);
    int slot = 0;
    while (true) {
      const int inext = i + Q;
      const bool has_next = inext < cnt;
      int mtn = 0, ntn = 0;
      if (has_next) {
        sk_tile_coords<BM, BN>(p, base + inext, mtn, ntn);
        setup(mtn, ntn, vn);
      }
#pragma unroll
      for (int ii = 0; ii < TM; ++ii)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[ii][jj][r] = 0.f; if constexpr (PAIR) nacc[ii][jj][r] = 0.f; }
      auto step = [&](int kt, auto odd_c) {
        wait_vmcnt<0>();                                // this tile's stage kt has landed (and the previous tile's stores are acknowledged)
        MT_PL_BARRIER();
        const int nslot = slot ^ 1;
        if (kt + 1 < nk) issue(vc, kt + 1, nslot);
        else if (has_next) issue(vn, 0, nslot);         // the next tile's first stage rides under this tile's last step and epilogue
        Frags f;
        read_frags(slot, f);
        mma(f, odd_c, 0u);
        slot = nslot;
      };
      for (int kt = 0; kt < nk; kt += 2) {
        step(kt, std::false_type{});
        if (kt + 1 >= nk) break;
        step(kt + 1, std::true_type{});
      }
      if constexpr (PAIR) {
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ii][jj][r] -= nacc[ii][jj][r];
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) asm volatile("" : "+v"(acc[ii][jj]));
      }
      epilogue(mt_, nt_);
      if (!has_next) break;
      i = inext; mt_ = mtn; nt_ = ntn;
#pragma unroll
      for (int j = 0; j < IPW; ++j) vc[j] = vn[j];
    }
  } else if constexpr (!SK) {
    int mt_, nt_;
    int k_begin = 0, k_end = p.K;
    int split_idx = blockIdx.y;
    if (p.xcd_k) {                                     // split-K weight gradient, K-range-major over the XCDs (see gemm_split.hpp)
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
    const int nk = (k_end - k_begin + BK - 1) / BK;    // k_begin % 16 == 0; a ragged end reads the planes' zero padding
    if (nk <= 0) return;
    run_piece(mt_, nt_, k_begin, 0, nk);
    epilogue(mt_, nt_, split_idx);
  } else {
    // logical block index: the blocks of one XCD (blockIdx % 8, observed placement; only speed depends on it) take a contiguous
    // eighth of the work list, i.e. of the XCD-ordered tile list
    const int G = gridDim.x;
    const int L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int nkt = (p.K + BK - 1) / BK;               // k-steps per tile
    const int npt = (nkt + 1) >> 1;                    // work units per tile: PAIRS of k-steps (the accumulator pair's parity stays static)
    const int U = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * npt;     // (host: fits an int)
    const int uq = U / G, ur = U % G;
    int u = L * uq + min(L, ur);
    const int u_end = u + uq + (L < ur ? 1 : 0);
    constexpr int SLAB = BM * BN;                      // floats per block slab
    while (u < u_end) {
      const int t = u / npt;
      const int kt0 = 2 * (u - t * npt);
      const int ulen = min(npt - (u - t * npt), u_end - u);
      const int len = min(nkt - kt0, 2 * ulen);
      int mt_, nt_;
      sk_tile_coords<BM, BN>(p, t, mt_, nt_);
      __builtin_amdgcn_s_barrier();                    // the previous piece's last stage reads are done before new tiles land
      run_piece(mt_, nt_, 0, kt0, len);
      u += ulen;
      if (len != nkt) {
        // per-lane slab offset, kept out of the loop-invariant hoisting (it would sit in registers through every k-loop)
        int lane_s = lane, wave_s = wave;
        asm volatile("" : "+v"(lane_s), "+s"(wave_s));
        const int my_off = wave_s * (TM * TN * 1024) + lane_s * 4;
        if (kt0 != 0) {
          // tail of a tile whose head belongs to an earlier block: publish the partial sums (the first thing this block does)
          float* slab = p.sk_ws + (int64_t)L * SLAB + my_off;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(slab + ((i * TN + j) * 4 + q) * 256) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.sk_flags + L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          continue;
        }
        // head of a tile that later blocks finish: add their slabs in order, then the epilogue
        int covered = ulen;
        for (int lb = L + 1; covered < npt; ++lb) {
          const int span = uq + (lb < ur ? 1 : 0);
          if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(p.sk_flags + lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
              __builtin_amdgcn_s_sleep(8);
              if (++spins > (1 << 24)) break;          // bounded: a lost producer must not hang the device (the result is then wrong)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          const float* slab = p.sk_ws + (int64_t)lb * SLAB + my_off;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(slab + ((i * TN + j) * 4 + q) * 256);
                acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
              }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();                             // every wave has read the slab: the flag may be cleared for the next launch
          if (tid == 0) __hip_atomic_store(p.sk_flags + lb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          covered += min(npt - covered, span);
        }
      }
      epilogue(mt_, nt_);
    }
  }
#undef MT_PL_BARRIER
}

}  // namespace mt
