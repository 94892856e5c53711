// Dispatch of the split-operand GEMM main loop (gemm_split.hpp): fp32 operands split exactly into three bf16 pieces, six
// piece products per fp32 product on the bf16 matrix pipe, fp32 accumulators.  mt_gemm (gemm.hip) asks try_launch_split()
// first; problems it does not take (most operand prologues, tiny or K % 8 != 0 shapes) go on to the fp32-MFMA kernels.
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "gemm_split.hpp"
#include "det.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

using namespace mt;

namespace {

// 0 = fp32 MFMA everywhere, 1 = split-operand bf16x6 where eligible.  Process-wide; MT_GEMM_SPLIT sets the initial value.
int g_mode = -1;
int mode() {
  if (g_mode < 0) {
    const char* e = getenv("MT_GEMM_SPLIT");
    g_mode = e ? (atoi(e) != 0) : 1;
  }
  return g_mode;
}

enum { S_BIG = 0, S_MID = 1, S_SMALL = 2, S_COUNT };      // 128 x 128 (4 waves of 64 x 64), 128 x 64 (64 x 32), 64 x 64 (32 x 32)
struct Var { int bm, bn; };
constexpr Var kVar[S_COUNT] = {{128, 128}, {128, 64}, {64, 64}};

template <int WM, int WN, int TM, int TN, int AL, int BL, int EPI, int MINW, bool BAL, int PRO = PRO_NONE, bool BPL = false>
int launch_one(const GemmArgs& a, dim3 grid, hipStream_t s) {
  auto k = gemm_split_kernel<WM, WN, TM, TN, AL, BL, EPI, MINW, true, 2, BAL, PRO, BPL>;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  size_t lds = BPL ? (size_t)(6 * BM + 9 * BN) * 32 : (size_t)2 * 3 * (BM + BN) * 32;
  if (PRO == PRO_BN_SWISH_GATE) lds += (size_t)(2 + (BM - 1) / a.hw + 2) * a.K * 4;      // scale, shift, gate rows of the images a row tile touches
  if (PRO == PRO_BN_BWD && AL == LAYOUT_KCONTIG) lds += (size_t)3 * a.K * 4;             // ka, kb, kc
  if (!BAL && AL == LAYOUT_KMAJOR) {
    // experiment knob: extra LDS per weight-gradient block (8192 -> two instead of three blocks per CU, leaving room for a
    // main-stream GEMM block).  Measured in-step: 55.3 vs 54.4 ms -- the weight gradients lose more than the main stream gains.
    static const int pad = getenv("MT_WGRAD_LDS_PAD") ? atoi(getenv("MT_WGRAD_LDS_PAD")) : 0;
    lds += (size_t)pad;
  }
  if (lds > 160 * 1024) return 1;      // not this way: the caller falls back to the fp32 kernels
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_gemm(split): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), lds, s, a);
  return check_launch("mt_gemm(split)");
}

template <int AL, int BL, int EPI, int PRO = PRO_NONE>
int launch_variant(int v, const GemmArgs& a, dim3 grid, hipStream_t s) {
  // Balanced accumulators (gemm_split.hpp: BAL) wherever a result feeds further contractions (activations, data gradients): the
  // bf16 pipe's rounding bias is only a problem when it adds up coherently through depth.  Weight gradients (TN) are leaves --
  // their error goes no further than lr x bias into the next step's weights -- and take the single-accumulator loop
  // (max / rms error equal to the fp32 pipe's, 6-8 % faster, 146 instead of 206 VGPRs).  MT_SPLIT_WGRAD_BAL=1 balances them too.
  if constexpr (AL == LAYOUT_KMAJOR && BL == LAYOUT_KMAJOR) {
    static const bool wbal = getenv("MT_SPLIT_WGRAD_BAL") && atoi(getenv("MT_SPLIT_WGRAD_BAL")) != 0;
    if (!wbal) {
      if (v == S_BIG) return launch_one<2, 2, 2, 2, AL, BL, EPI, 2, false, PRO>(a, grid, s);
      if (v == S_MID) return launch_one<2, 2, 2, 1, AL, BL, EPI, 3, false, PRO>(a, grid, s);
      if (v == S_SMALL) return launch_one<2, 2, 1, 1, AL, BL, EPI, 4, false, PRO>(a, grid, s);
    }
  }
  if constexpr (AL == LAYOUT_KCONTIG && BL == LAYOUT_KCONTIG && EPI != EPI_STATS && PRO == PRO_NONE) {
    // weight pre-split into bf16 planes (mt_split_planes): B by DMA, 128 x 128 tiles only.  Bit-identical to the in-kernel split.
    // Round 2: 7-12 % slower (hipcc's vmcnt for the staged A tile also drained the DMA of the same step).  Round 3: the A loads are
    // inline asm under one counted wait per step (no compiler-inserted vmcnt left in the loop, checked in the ISA) -- and the variant
    // now runs EQUAL to the in-kernel split, not faster (profiles/r03_split_planes_manual_waits.txt:
    // QKV 154.7 / 153.6 us, FF2 203.8 / 201.2, FF1 data gradient 353.5 / 352.0, 4096^3 747 / 780), with two or with three A register
    // sets in flight: the loop is limited by its matrix + LDS issue, not by operand delivery.  Stays opt-in (MT_SPLIT_PLANES=1).
    const bool planes_on = getenv("MT_SPLIT_PLANES") && atoi(getenv("MT_SPLIT_PLANES")) != 0;
    if (planes_on && a.b_planes && (a.K % 16) == 0 && a.k_chunk == 0 && (a.ldb % 8) == 0 && a.b_map.gin == 0 && (v == S_BIG || EPI == EPI_GEGLU_BWD))
      return launch_one<2, 2, 2, 2, AL, BL, EPI, 2, true, PRO_NONE, true>(a, grid, s);
  }
  if (v == S_BIG) return launch_one<2, 2, 2, 2, AL, BL, EPI, 2, true, PRO>(a, grid, s);
  if constexpr (EPI != EPI_GEGLU) {
    if (v == S_MID) return launch_one<2, 2, 2, 1, AL, BL, EPI, 3, true, PRO>(a, grid, s);
    if (v == S_SMALL) return launch_one<2, 2, 1, 1, AL, BL, EPI, 4, true, PRO>(a, grid, s);
  }
  return fail(MT_ERR_UNSUPPORTED, "mt_gemm(split): no instance for variant %d", v);
}

}  // namespace

extern "C" int mt_gemm_set_split(int on) {
  const int prev = mode();
  g_mode = on != 0;
  return prev;
}

extern "C" int mt_gemm_get_split(void) { return mode(); }

namespace {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ planes, int64_t n) {
  const int64_t n8 = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 u = reinterpret_cast<const float4*>(src)[2 * i], v = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float lo[4] = {u.x, u.y, u.z, u.w}, hi[4] = {v.x, v.y, v.z, v.w};
    bf16x8_t x0, x1, x2;
    split_bf16<true>(lo, hi, x0, x1, x2);
    reinterpret_cast<bf16x8_t*>(planes)[i] = x0;
    reinterpret_cast<bf16x8_t*>(planes + n)[i] = x1;
    reinterpret_cast<bf16x8_t*>(planes + 2 * n)[i] = x2;
  }
}
}  // namespace

extern "C" int mt_split_planes(const float* src, void* planes, int64_t n, void* stream) {
  if (!src || !planes) return fail(MT_ERR_ARG, "mt_split_planes: null pointer");
  if (n <= 0 || (n & 7) || ((uintptr_t)src & 15) || ((uintptr_t)planes & 15))
    return fail(MT_ERR_ARG, "mt_split_planes: n must be a positive multiple of 8, pointers 16-byte aligned");
  const int64_t n8 = n >> 3;
  const unsigned blocks = (unsigned)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<__bf16*>(planes), n);
  return check_launch("mt_split_planes");
}

namespace mt {

// Returns 1 when the problem is not eligible (caller falls back), 0 on success, < 0 on error.
int try_launch_split(const mt_gemm_desc* d, GemmArgs a, hipStream_t s) {
  if (!mode()) return 1;
  if (const char* f = getenv("MT_SPLIT_ONLY_EPI")) {                // bisection aid: only this epilogue (and K, if given) takes the split loop
    if (d->epilogue != atoi(f)) return 1;
    if (const char* k = getenv("MT_SPLIT_ONLY_K")) if (d->K != atoi(k)) return 1;
    static int calls = 0;                                           // ... and only calls [MT_SPLIT_FIRST, MT_SPLIT_LAST] of those
    const int idx = calls++;
    if (const char* lo = getenv("MT_SPLIT_FIRST")) if (idx < atoi(lo)) return 1;
    if (const char* hi = getenv("MT_SPLIT_LAST")) if (idx > atoi(hi)) return 1;
    if (getenv("MT_SPLIT_TRACE")) fprintf(stderr, "[split] call %d: op %d M %d N %d K %d\n", idx, d->op, d->M, d->N, d->K);
  }
  if (d->b_prologue != MT_BPRO_NONE) return 1;
  if (d->K % 8 || d->M < 128 || d->N < 64) return 1;     // K % 16 == 8: the last k-tile is half zeros (gemm_split.hpp k_tail)
  if (d->prologue == MT_PRO_BN_SWISH_GATE) {
    // MBConv project convolution (forward): the operand transform rides in the staging registers (gemm_split.hpp PRO)
    static const int pro_on = getenv("MT_SPLIT_PRO") ? atoi(getenv("MT_SPLIT_PRO")) : 1;
    if (!pro_on || d->op != MT_OP_NT || d->M < 4096 || d->K < 256 || (d->K % 16)) return 1;
    if (d->epilogue != MT_EPI_STATS && d->epilogue != MT_EPI_STORE) return 1;
    // tile width by padding waste: 128 columns unless 64-wide tiles waste fewer padded columns
    const int pad128 = (d->N + 127) / 128 * 128, pad64 = (d->N + 63) / 64 * 64;
    const int v = pad64 < pad128 ? S_MID : S_BIG;
    const Var var = kVar[v];
    const int m_tiles = (d->M + var.bm - 1) / var.bm, n_tiles = (d->N + var.bn - 1) / var.bn;
    dim3 grid(m_tiles * n_tiles, 1, 1);
    a.group_n = 0; a.k_chunk = 0; a.trace = nullptr;
    if (m_tiles >= 32 && n_tiles >= 2 && !getenv("MT_NO_L2_BLOCKING")) {
      const int64_t panel = (int64_t)var.bn * d->K * 4;
      int gn = (int)((2 << 20) / (panel > 0 ? panel : 1));
      if (gn < 1) gn = 1;
      if (gn > n_tiles) gn = n_tiles;
      a.group_n = gn;
      grid.x = 8 * ((m_tiles + 7) / 8) * n_tiles;
    }
    constexpr int KC = LAYOUT_KCONTIG;
    if (d->epilogue == MT_EPI_STATS)
      return v == S_BIG ? launch_one<2, 2, 2, 2, KC, KC, EPI_STATS, 2, true, PRO_BN_SWISH_GATE>(a, grid, s)
                        : launch_one<2, 2, 2, 1, KC, KC, EPI_STATS, 3, true, PRO_BN_SWISH_GATE>(a, grid, s);
    return v == S_BIG ? launch_one<2, 2, 2, 2, KC, KC, EPI_STORE, 2, true, PRO_BN_SWISH_GATE>(a, grid, s)
                      : launch_one<2, 2, 2, 1, KC, KC, EPI_STORE, 3, true, PRO_BN_SWISH_GATE>(a, grid, s);
  }
  // BatchNorm-backward operand prologue (data gradients NN, weight gradients TN with a plain second operand): VALU work in the
  // staging registers like the split itself.  Long contractions only (the Xception pointwise convolutions, EfficientNet's
  // expand convolutions from stage 5 on and its head; the weight gradients' K is the row count).
  const bool bn_bwd = d->prologue == MT_PRO_BN_BWD;
  if (bn_bwd) {
    static const int on = getenv("MT_SPLIT_BN_BWD") ? atoi(getenv("MT_SPLIT_BN_BWD")) : 1;
    if (!on || d->a_map.gin != 0) return 1;
    const bool nn = d->op == MT_OP_NN && (d->epilogue == MT_EPI_STORE || d->epilogue == MT_EPI_BIAS_RES);
    const bool tn = d->op == MT_OP_TN && d->epilogue == MT_EPI_ATOMIC;
    if (!nn && !tn) return 1;
  } else if (d->prologue != MT_PRO_NONE) return 1;
  // short contractions stay on the fp32 pipe: the matrix time they could save is small next to their epilogue, and the bf16 pipe's
  // residual rounding bias (gemm_split.hpp) is then kept out of the extractors' long chains of small-K convolutions
  static const int min_k = getenv("MT_SPLIT_MIN_K") ? atoi(getenv("MT_SPLIT_MIN_K")) : 512;
  if (d->K < min_k) return 1;
  if (d->epilogue == MT_EPI_STATS && d->M < 4096) return 1;
  if (d->epilogue == MT_EPI_GEGLU && (d->n_half & 63)) return 1;
  // a handful of tiles: the fp32 kernels' small tiles fill the chip better -- unless it is a weight gradient over very many rows,
  // whose K-ranges supply the blocks
  if ((int64_t)d->M * d->N < (1 << 18) && !(d->op == MT_OP_TN && d->K >= 8192 && (int64_t)d->M * d->N >= (1 << 15))) return 1;
  int v = S_BIG;
  if (d->epilogue == MT_EPI_GEGLU_BWD) v = S_MID;
  else if (d->N < 128) v = S_MID;
  if (const char* f = getenv("MT_SPLIT_VARIANT")) v = atoi(f);      // tuning experiments only
  if (const char* f = getenv("MT_SPLIT_VARIANT_EPI")) {             // "epi:variant[,epi:variant...]"
    for (const char* q = f; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) {
      int e = -1, vv = -1;
      if (sscanf(q, "%d:%d", &e, &vv) == 2 && e == d->epilogue + 10 * d->op) v = vv;
    }
  }
  if (v != S_BIG && d->epilogue == MT_EPI_GEGLU) v = S_BIG;
  if (v < 0 || v >= S_COUNT) return 1;
  const Var var = kVar[v];
  const int m_tiles = (d->M + var.bm - 1) / var.bm, n_tiles = (d->N + var.bn - 1) / var.bn;
  dim3 grid(m_tiles * n_tiles, 1, 1);
  a.group_n = 0;
  a.k_chunk = 0;
  a.trace = nullptr;
  if (m_tiles >= 32 && n_tiles >= 2 && !getenv("MT_NO_L2_BLOCKING")) {
    const int64_t panel = (int64_t)var.bn * d->K * 4;
    int gn = (int)((2 << 20) / (panel > 0 ? panel : 1));
    if (gn < 1) gn = 1;
    if (gn > n_tiles) gn = n_tiles;
    a.group_n = gn;
    grid.x = 8 * ((m_tiles + 7) / 8) * n_tiles;
  }
  if (d->op == MT_OP_TN || d->epilogue == MT_EPI_ATOMIC) {
    int splits = d->split_k;
    if (d->op == MT_OP_TN && splits <= 0) {
      const int tiles = m_tiles * n_tiles;
      // blocks per launch.  Round 2: 2048 (16-49 K-ranges, whose fp32-atomic partial sums were as much HBM-side traffic as the
      // operands).  With the K-range-major XCD mapping the step time is flat from 384 to 2048 (53.7-54.2 ms); 640 keeps 8-40
      // ranges: partial-sum traffic 104 -> ~48 MB per launch (family total ~1.45x the algorithmic bytes).
      static const int target = getenv("MT_WGRAD_BLOCKS") ? atoi(getenv("MT_WGRAD_BLOCKS")) : 640;   // tuning knob
      splits = (target + tiles - 1) / tiles;
      const int max_splits = d->K / 256 > 0 ? d->K / 256 : 1;
      if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    static const int xcd_k_on = getenv("MT_WGRAD_XCD_K") ? atoi(getenv("MT_WGRAD_XCD_K")) : 1;
    if (d->op == MT_OP_TN && xcd_k_on && d->split_k <= 0 && d->K >= 8 * 256 && d->a_map.gin == 0 && d->b_map.gin == 0) {
      // K-range-major over the XCDs (gemm_split.hpp): a multiple of 8 ranges, exactly m_tiles * n_tiles blocks per range
      const int max_splits8 = (d->K / 256) / 8 * 8;                    // (>= 8: K >= 8 * 256 here)
      splits = (splits + 4) / 8 * 8;
      if (splits > max_splits8) splits = max_splits8;
      if (splits < 8) splits = 8;
      int chunk = (d->K + splits - 1) / splits;
      chunk = (chunk + 15) / 16 * 16;
      a.k_chunk = chunk;
      a.xcd_k = 1;
      a.group_n = 0;
      grid.x = m_tiles * n_tiles;
      grid.y = ((d->K + chunk - 1) / chunk + 7) / 8 * 8;               // no K-range beyond the last non-empty group of 8
    } else {
      int chunk = (d->K + splits - 1) / splits;
      chunk = (chunk + 15) / 16 * 16;
      a.k_chunk = chunk;
      grid.y = (d->K + chunk - 1) / chunk;
    }
    if (d->epilogue == MT_EPI_ATOMIC)
      if (int rc = det_gemm_setup(a.C, a.ldc, a.det_slab, d->M, d->N, (d->K + a.k_chunk - 1) / a.k_chunk, a.c_map.gin != 0, s)) return rc;
  }

  if (bn_bwd) {
    if (d->op == MT_OP_TN) return launch_variant<LAYOUT_KMAJOR, LAYOUT_KMAJOR, EPI_ATOMIC, PRO_BN_BWD>(v, a, grid, s);
    if (d->epilogue == MT_EPI_BIAS_RES) return launch_variant<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_BIAS_RES, PRO_BN_BWD>(v, a, grid, s);
    return launch_variant<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_STORE, PRO_BN_BWD>(v, a, grid, s);
  }
#define SPLIT_COMBO(OP, AL, BL, EPI)                                 \
  if (d->op == OP && d->epilogue == EPI) return launch_variant<AL, BL, EPI>(v, a, grid, s);
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STORE)
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_BIAS_RES)
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_GEGLU)
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_ATOMIC)
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STATS)
  SPLIT_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_GEGLU_BWD)   // data gradient over a transposed weight (tsf_backward.py)
  SPLIT_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_STORE)
  SPLIT_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_ACCUM)
  SPLIT_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_GEGLU_BWD)
  SPLIT_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_ATOMIC)
  SPLIT_COMBO(MT_OP_TN, LAYOUT_KMAJOR, LAYOUT_KMAJOR, EPI_ATOMIC)
#undef SPLIT_COMBO
  return 1;
}

}  // namespace mt
