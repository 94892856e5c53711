// Dispatch of the LDS-DMA GEMM main loop (gemm_dma.hpp) for prologue-free problems.  mt_gemm (gemm.hip) asks try_launch_dma()
// first and falls back to the register-staged kernels when a problem is not eligible.
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "gemm_dma.hpp"
#include "det.hpp"
#include <stdlib.h>
#include <string.h>

using namespace mt;

namespace {

// Variants (tile, BK, ring depth, waves/SIMD the register budget is held to).  Chosen from tools/lab/gemm_lab sweeps over the
// shapes of a B = 32 training step (profiles/r02_gemm_dma_lab.txt): deeper rings stop paying once 2-3 blocks share a CU, BK = 32
// halves the barrier count for the long-K problems, 64x64 tiles win whenever 128x128 leaves fewer than ~3 tiles per CU.
enum { V_BIG32 = 0, V_BIG16 = 1, V_MID16 = 2, V_SMALL32 = 3, V_SMALL16 = 4, V_BIG16W8 = 5, V_COUNT };
struct Var { int bm, bn, bk, stages; };
constexpr Var kVar[V_COUNT] = {{128, 128, 32, 2}, {128, 128, 16, 3}, {128, 64, 16, 3}, {64, 64, 32, 3}, {64, 64, 16, 3}, {128, 128, 16, 2}};

int pick_variant(const mt_gemm_desc* d) {
  if (const char* f = getenv("MT_DMA_VARIANT")) return atoi(f);      // tuning experiments only
  const bool k32 = (d->K % 32) == 0;
  if (d->epilogue == MT_EPI_GEGLU) return k32 ? V_BIG32 : V_BIG16;
  if (d->epilogue == MT_EPI_GEGLU_BWD) return V_MID16;
  if (d->op == MT_OP_TN) {
    // eight waves of 32 x 64 over the same 128 x 128 tile: the k-major operand reads of a weight gradient hide better behind
    // twice the waves per SIMD (lab: 1536x512x12576 205 -> 183 us, 4096x512x12576 436 -> 430 us)
    static const int tn8 = getenv("MT_DMA_TN8") ? atoi(getenv("MT_DMA_TN8")) : 0;   // in-step 63.5 -> 64.5 ms: off
    if (d->M >= 1024 && d->N >= 512) return tn8 ? V_BIG16W8 : V_BIG16;
    if ((int64_t)d->M * d->N >= (1 << 20)) return V_MID16;
    return k32 ? V_SMALL32 : V_SMALL16;
  }
  if (d->N >= 1024 && d->M >= 4096) return k32 ? V_BIG32 : V_BIG16;
  if (d->epilogue == MT_EPI_ATOMIC) {
    if (d->op == MT_OP_NN) return (d->K >= 4096 && k32) ? V_BIG32 : V_BIG16;
    return V_MID16;
  }
  if (d->K >= 1024 && d->M >= 4096) {       // tall, skinny, long contraction (FF2, the N = 512 data gradients)
    static const int v = getenv("MT_DMA_SKINNY_VARIANT") ? atoi(getenv("MT_DMA_SKINNY_VARIANT")) : -1;   // tuning knob
    if (v >= 0) return v;
  }
  return k32 ? V_SMALL32 : V_SMALL16;
}

template <int WM, int WN, int TM, int TN, int AL, int BL, int EPI, int BK, int ST, int MINW, int PRO = PRO_NONE>
int launch_one(const GemmArgs& a, dim3 grid, hipStream_t s) {
  auto k = gemm_dma_kernel<WM, WN, TM, TN, AL, BL, EPI, BK, ST, MINW, PRO>;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  size_t lds = (size_t)ST * (BM * (PRO == PRO_BN_BWD ? 2 : 1) + BN) * BK * 4;
  if (PRO == PRO_BN_SWISH_GATE) lds += (size_t)(2 + (BM - 1) / a.hw + 2) * a.K * 4;     // scale, shift, gate rows of the images a row tile touches
  if (PRO == PRO_BN_BWD) lds += (size_t)3 * a.K * 4;                                     // ka, kb, kc
  if (lds > 160 * 1024) return 1;      // not this way: caller falls back to the register-staged kernel
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_gemm(dma): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, grid, dim3(WM * WN * 64), lds, s, a);
  return check_launch("mt_gemm(dma)");
}

// prologue problems (the extractor's project convs and 1x1-conv data gradients): 64x64 or 128x64 tiles
template <int AL, int BL, int EPI, int PRO>
int launch_pro(int v, const GemmArgs& a, dim3 grid, hipStream_t s) {
  switch (v) {
    case V_MID16: return launch_one<2, 2, 2, 1, AL, BL, EPI, 16, 2, 3, PRO>(a, grid, s);
    case V_SMALL32: return launch_one<2, 2, 1, 1, AL, BL, EPI, 32, 2, 4, PRO>(a, grid, s);
    case V_SMALL16: return launch_one<2, 2, 1, 1, AL, BL, EPI, 16, 3, 4, PRO>(a, grid, s);
    default: break;
  }
  return 1;
}

template <int AL, int BL, int EPI>
int launch_variant(int v, const GemmArgs& a, dim3 grid, hipStream_t s) {
  static const bool st2 = getenv("MT_DMA_ST2") && atoi(getenv("MT_DMA_ST2")) != 0;   // experiment: 2-deep rings (smaller LDS footprint)
  if (st2) {
    switch (v) {
      case V_BIG16: return launch_one<2, 2, 2, 2, AL, BL, EPI, 16, 2, 3>(a, grid, s);
      default: break;
    }
    if constexpr (EPI != EPI_GEGLU) {
      switch (v) {
        case V_MID16: return launch_one<2, 2, 2, 1, AL, BL, EPI, 16, 2, 3>(a, grid, s);
        case V_SMALL32: return launch_one<2, 2, 1, 1, AL, BL, EPI, 32, 2, 4>(a, grid, s);
        case V_SMALL16: return launch_one<2, 2, 1, 1, AL, BL, EPI, 16, 2, 4>(a, grid, s);
        default: break;
      }
    }
  }
  switch (v) {
    case V_BIG32: return launch_one<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2>(a, grid, s);
    case V_BIG16: return launch_one<2, 2, 2, 2, AL, BL, EPI, 16, 3, 3>(a, grid, s);
    default: break;
  }
  if constexpr (EPI == EPI_ATOMIC && AL == LAYOUT_KMAJOR) {
    if (v == V_BIG16W8) return launch_one<4, 2, 1, 2, AL, BL, EPI, 16, 2, 8>(a, grid, s);
  }
  if constexpr (EPI != EPI_GEGLU) {
    switch (v) {
      case V_MID16: return launch_one<2, 2, 2, 1, AL, BL, EPI, 16, 3, 3>(a, grid, s);
      case V_SMALL32: return launch_one<2, 2, 1, 1, AL, BL, EPI, 32, 3, 4>(a, grid, s);
      case V_SMALL16: return launch_one<2, 2, 1, 1, AL, BL, EPI, 16, 3, 4>(a, grid, s);
      default: break;
    }
  }
  return fail(MT_ERR_UNSUPPORTED, "mt_gemm(dma): no instance for variant %d", v);
}

}  // namespace

namespace mt {

// Returns 1 when the problem is not eligible (caller falls back), 0 on success, < 0 on error.  `a` is the argument block
// mt_gemm already filled (pointers, shapes, maps, epilogue extras); tile-order and split-K fields are set here.
int try_launch_dma(const mt_gemm_desc* d, GemmArgs a, hipStream_t s) {
  static const bool disabled = getenv("MT_GEMM_DMA") && atoi(getenv("MT_GEMM_DMA")) == 0;
  if (disabled) return 1;
  if (d->b_prologue != MT_BPRO_NONE) return 1;
  if (d->M < 64 || d->N < 64) return 1;
  if (d->prologue != MT_PRO_NONE) {
    // operand transforms applied at fragment-read time (gemm_dma.hpp PRO): project conv forward and the 1x1-conv data gradients
    static const int pro_on = getenv("MT_DMA_PRO") ? atoi(getenv("MT_DMA_PRO")) : 1;
    if (!pro_on || (d->K % 16) || d->M < 4096) return 1;
    // the gated forward form is correct (tests) but measured slower than the register-staged kernel (swish evaluated per
    // fragment read, i.e. twice per element and inside each wave's MFMA stream: 115 -> 138 us on 50176x80x480): opt-in only
    const bool gate_on = getenv("MT_DMA_PRO_GATE") != nullptr;          // (read per call: tests toggle it)
    const bool gate_fwd = gate_on && d->op == MT_OP_NT && d->prologue == MT_PRO_BN_SWISH_GATE && (d->epilogue == MT_EPI_STATS || d->epilogue == MT_EPI_STORE);
    const bool bn_bwd = d->op == MT_OP_NN && d->prologue == MT_PRO_BN_BWD &&
                        (d->epilogue == MT_EPI_STORE || d->epilogue == MT_EPI_BIAS_RES || d->epilogue == MT_EPI_SE_RED || d->epilogue == MT_EPI_ACT_BWD);
    if (!gate_fwd && !bn_bwd) return 1;
    const int vforce = getenv("MT_DMA_PRO_VARIANT") ? atoi(getenv("MT_DMA_PRO_VARIANT")) : -1;
    int v = vforce >= 0 ? vforce : ((d->K % 32) == 0 ? V_SMALL32 : V_SMALL16);
    if (kVar[v].bk == 32 && (d->K % 32)) v = V_SMALL16;
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
    if (gate_fwd) {
      if (d->epilogue == MT_EPI_STATS) return launch_pro<LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STATS, PRO_BN_SWISH_GATE>(v, a, grid, s);
      return launch_pro<LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STORE, PRO_BN_SWISH_GATE>(v, a, grid, s);
    }
    if (d->epilogue == MT_EPI_BIAS_RES) return launch_pro<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_BIAS_RES, PRO_BN_BWD>(v, a, grid, s);
    if (d->epilogue == MT_EPI_SE_RED) return launch_pro<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_SE_RED, PRO_BN_BWD>(v, a, grid, s);
    if (d->epilogue == MT_EPI_ACT_BWD) return launch_pro<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_ACT_BWD, PRO_BN_BWD>(v, a, grid, s);
    return launch_pro<LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_STORE, PRO_BN_BWD>(v, a, grid, s);
  }
  if (d->epilogue == MT_EPI_GEGLU && (d->n_half & 63)) return 1;
  int v = pick_variant(d);
  if (v < 0 || v >= V_COUNT) return 1;
  if (d->K % kVar[v].bk) {
    if (d->K % 16) return 1;
    v = v == V_BIG32 ? V_BIG16 : (v == V_SMALL32 ? V_SMALL16 : v);
    if (d->K % kVar[v].bk) return 1;
  }
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
      static const int target = getenv("MT_WGRAD_BLOCKS") ? atoi(getenv("MT_WGRAD_BLOCKS")) : 2048;   // tuning knob
      splits = (target + tiles - 1) / tiles;
      const int max_splits = d->K / 256 > 0 ? d->K / 256 : 1;
      if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    int chunk = (d->K + splits - 1) / splits;
    chunk = (chunk + var.bk - 1) / var.bk * var.bk;
    a.k_chunk = chunk;
    grid.y = (d->K + chunk - 1) / chunk;
    if (d->epilogue == MT_EPI_ATOMIC)
      if (int rc = det_gemm_setup(a.C, a.ldc, a.det_slab, d->M, d->N, (int)grid.y, a.c_map.gin != 0, s)) return rc;
  }

#define DMA_COMBO(OP, AL, BL, EPI)                                 \
  if (d->op == OP && d->epilogue == EPI) return launch_variant<AL, BL, EPI>(v, a, grid, s);
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STORE)
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_BIAS_RES)
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_GEGLU)
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_STATS)
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_ATOMIC)
  DMA_COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, EPI_GEGLU_BWD)
  DMA_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_STORE)
  DMA_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_ACCUM)
  DMA_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_GEGLU_BWD)
  DMA_COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, EPI_ATOMIC)
  DMA_COMBO(MT_OP_TN, LAYOUT_KMAJOR, LAYOUT_KMAJOR, EPI_ATOMIC)
#undef DMA_COMBO
  return 1;
}

}  // namespace mt
