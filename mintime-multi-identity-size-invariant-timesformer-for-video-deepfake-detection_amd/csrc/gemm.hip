// C-ABI dispatch for the fp32 MFMA GEMM family (see gemm_core.hpp, include/mintime_hip.h).
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "gemm_core.hpp"
#include "det.hpp"
#include <stdlib.h>

using namespace mt;

namespace {

enum { CFG_BIG = 0, CFG_MID = 1, CFG_NARROW = 2, CFG_SMALL = 3, CFG_WG64 = 4, CFG_WG64K = 5 };

struct Cfg { int bm, bn, threads; };
// CFG_WG64: the weight gradient of a dense 3x3 convolution with <= 64 output channels and 9 x 32 gathered columns (Xception's conv2,
// reference models/xception.py:164: out[64][288] over 11 M pixel rows): ONE 64 x 288 tile per K range (six wavefronts, 32 x 96 each)
// -- the 128 x 64 tile computed 64 rows of zeros and re-read the dy / z operand pair once per column tile (5x).
constexpr Cfg kCfg[6] = {{128, 128, 256}, {128, 64, 256}, {256, 32, 256}, {64, 64, 256}, {64, 288, 384}, {64, 288, 768}};

// Tile choice.  Replaying every GEMM of a B = 32 training step alone under each configuration (round-1 tuning script; results: profiles/r01_gemm_tile_config_sweep.txt, 70
// shapes) favours 64x64 tiles almost everywhere, but inside the real step -- where the weight-gradient GEMMs of a second stream
// share the CUs -- only the cases below kept their gain (and Xception's large pointwise GEMMs lost 7 % with 64x64 everywhere):
//   * tall problems with so few 128x128 tiles that they cannot fill the resident slots once (12576 x 512: 396 tiles on 256 CUs
//     x 3-4 blocks): 64x64 quadruples the block count and evens out the per-CU load (out-proj 97 -> 81 us);
//   * the GEGLU-backward epilogue (reads the pre-activations, writes two gradients per element): small tiles let one block's
//     epilogue hide under its neighbours' main loops (365 -> 298 us).
int pick_cfg(int op, int M, int N, int prologue, int epilogue) {
  (void)prologue;
  if (epilogue == MT_EPI_GEGLU) return CFG_BIG;
  if (const char* f = getenv("MT_FORCE_CFG")) return atoi(f);   // tuning experiments only
  if (N <= 32) return CFG_NARROW;
  if (epilogue == MT_EPI_GEGLU_BWD) return CFG_SMALL;
  if (op != MT_OP_TN && M >= 4096 && N >= 128 && (int64_t)((M + 127) / 128) * ((N + 127) / 128) <= 768) return CFG_SMALL;
  const int pad_big = (N + 127) / 128 * 128;
  const int pad_mid = (N + 63) / 64 * 64;
  return pad_mid < pad_big ? CFG_MID : CFG_BIG;
}

template <int AL, int BL, int PRO, int EPI, int BPRO = BPRO_NONE>
int launch(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
  switch (cfg) {
    case CFG_BIG:
      hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2, AL, BL, PRO, EPI, BPRO>), grid, dim3(256), 0, s, a);
      break;
    case CFG_MID:
      if constexpr (EPI == EPI_GEGLU) return fail(MT_ERR_UNSUPPORTED, "GEGLU needs the 128x128 tile");
      else hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 1, AL, BL, PRO, EPI, BPRO>), grid, dim3(256), 0, s, a);
      break;
    case CFG_NARROW:
      if constexpr (EPI == EPI_GEGLU) return fail(MT_ERR_UNSUPPORTED, "GEGLU needs the 128x128 tile");
      else hipLaunchKernelGGL((gemm_kernel<4, 1, 2, 1, AL, BL, PRO, EPI, BPRO>), grid, dim3(256), 0, s, a);
      break;
    case CFG_SMALL:
      if constexpr (EPI == EPI_GEGLU) return fail(MT_ERR_UNSUPPORTED, "GEGLU needs a 128-column tile");
      else hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1, AL, BL, PRO, EPI, BPRO>), grid, dim3(256), 0, s, a);
      break;
    case CFG_WG64:
      if constexpr ((BPRO == BPRO_IM2COL || BPRO == BPRO_IM2COL_ANY) && EPI == EPI_ATOMIC)
        hipLaunchKernelGGL((gemm_kernel<2, 3, 1, 3, AL, BL, PRO, EPI, BPRO>), grid, dim3(384), 0, s, a);
      else return fail(MT_ERR_UNSUPPORTED, "the 64 x 288 tile is the im2col weight gradient's");
      break;
    case CFG_WG64K:     // the same tile by two K groups of six wavefronts (gemm_core.hpp WAVES_K): 3 wavefronts on every SIMD
      if constexpr ((BPRO == BPRO_IM2COL || BPRO == BPRO_IM2COL_ANY) && EPI == EPI_ATOMIC)
        hipLaunchKernelGGL((gemm_kernel<2, 3, 1, 3, AL, BL, PRO, EPI, BPRO, 2>), grid, dim3(768), 0, s, a);
      else return fail(MT_ERR_UNSUPPORTED, "the 64 x 288 tile is the im2col weight gradient's");
      break;

  }
  return check_launch("mt_gemm");
}

}  // namespace

namespace mt { int try_launch_dma(const mt_gemm_desc* d, GemmArgs a, hipStream_t s); }   // gemm_dma.hip
namespace mt { int try_launch_split(const mt_gemm_desc* d, GemmArgs a, hipStream_t s); } // gemm_split.hip

static long long* g_trace = nullptr;
// tuning aid, not part of the ABI header: per-block phase timestamps of the following mt_gemm launches (NULL = off)
extern "C" void mt_debug_gemm_trace(long long* buf) { g_trace = buf; }

static int gemm_impl(const mt_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return fail(MT_ERR_ARG, "mt_gemm: null pointer");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return fail(MT_ERR_ARG, "mt_gemm: bad shape %d %d %d", d->M, d->N, d->K);
  if ((d->lda & 3) || (d->ldb & 3)) return fail(MT_ERR_ARG, "mt_gemm: lda/ldb must be multiples of 4 floats");
  const bool u8_src = d->conv_src_u8 != 0;
  if ((!(u8_src && d->prologue == MT_PRO_IM2COL) && ((uintptr_t)d->A & 15)) || (!(u8_src && d->b_prologue == MT_BPRO_IM2COL) && ((uintptr_t)d->B & 15)))
    return fail(MT_ERR_ARG, "mt_gemm: A/B must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;

  GemmArgs a;
  a.A = d->A; a.B = d->B; a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.a_map = {d->a_map.gin, d->a_map.gout, d->a_map.off};
  a.b_map = {d->b_map.gin, d->b_map.gout, d->b_map.off};
  a.c_map = {d->c_map.gin, d->c_map.gout, d->c_map.off};
  a.bias = d->bias; a.R = d->R; a.ldr = d->ldr;
  a.scale = d->scale; a.shift = d->shift; a.gate = d->gate; a.hw = d->hw > 0 ? d->hw : 1;
  a.C2 = d->C2; a.ldc2 = d->ldc2; a.stats = d->stats; a.stats_slots = d->stats_slots != 0 ? d->stats_slots : 1;
  a.n_half = d->n_half; a.k_chunk = 0;
  a.conv = conv_desc_of(d->conv_H, d->conv_W, d->conv_C, d->conv_Ho, d->conv_Wo, d->conv_k, d->conv_stride, d->conv_pad, d->conv_act, d->conv_src_u8);
  a.col_sum = d->col_sum;
  a.b_planes = d->b_planes; a.b_pstride = d->b_plane_stride;
  a.a_planes = nullptr; a.a_pstride = 0; a.c_planes = nullptr; a.c_pstride = 0; a.ldcp = 0; a.sk_ws = nullptr; a.sk_flags = nullptr; a.sk_on = 0; a.wave_prio = 0;
  a.e_scale = d->e_scale; a.e_shift = d->e_shift; a.e_gate = d->e_gate; a.e_dpool = d->e_dpool; a.e_mi = d->e_mi;
  a.e_hw = d->e_hw > 0 ? d->e_hw : 1;
  a.xcd_k = 0; a.det_slab = 0;
  a.det.vals = nullptr; a.det.base = nullptr; a.det.R = 0; a.det.P = 0;
  if (d->epilogue == MT_EPI_GEGLU_BWD)
    if (int rc = det_gemm_colsum_setup(a.det, d->M, d->n_half, d->col_sum, s)) return rc;
  a.A2 = d->A2; a.b_scale = d->b_scale; a.b_shift = d->b_shift; a.b_gate = d->b_gate; a.b_hw = d->b_hw > 0 ? d->b_hw : 1;

  // K-contiguous operands need K % 4 == 0 (float4 along K); k-major operands need M resp. N % 4 == 0
  if (d->op == MT_OP_NT && (d->K & 3)) return fail(MT_ERR_ARG, "mt_gemm NT: K %% 4 != 0");
  if (d->op == MT_OP_NN && ((d->K & 3) || (d->N & 3))) return fail(MT_ERR_ARG, "mt_gemm NN: K,N %% 4 != 0");
  if (d->op == MT_OP_TN && ((d->M & 3) || (d->N & 3))) return fail(MT_ERR_ARG, "mt_gemm TN: M,N %% 4 != 0");
  if (d->epilogue == MT_EPI_GEGLU && (d->n_half * 2 != d->N || (d->n_half & 63)))
    return fail(MT_ERR_ARG, "mt_gemm GEGLU: N must be 2*n_half, n_half %% 64 == 0");
  if (d->epilogue == MT_EPI_BIAS_RES && !d->R) return fail(MT_ERR_ARG, "mt_gemm: BIAS_RES needs R");
  if (d->epilogue == MT_EPI_STATS && !d->stats) return fail(MT_ERR_ARG, "mt_gemm: STATS needs stats");
  if (d->epilogue == MT_EPI_GEGLU_BWD && !d->C2) return fail(MT_ERR_ARG, "mt_gemm: GEGLU_BWD needs C2 (pre-activations)");
  if (d->prologue != MT_PRO_NONE && d->prologue != MT_PRO_IM2COL && (!d->scale || !d->shift))
    return fail(MT_ERR_ARG, "mt_gemm: prologue needs scale/shift");
  if (d->prologue == MT_PRO_IM2COL || d->b_prologue == MT_BPRO_IM2COL) {
    if (d->conv_k <= 0 || d->conv_stride <= 0 || d->conv_C <= 0 || d->conv_Ho <= 0 || d->conv_Wo <= 0)
      return fail(MT_ERR_ARG, "mt_gemm: im2col prologue needs the conv_* geometry");
    if (d->conv_src_u8 && (d->conv_C & 3) == 0) return fail(MT_ERR_UNSUPPORTED, "mt_gemm: uint8 im2col source needs C %% 4 != 0 (3-channel crops)");
    if ((d->scale == nullptr) != (d->shift == nullptr) || (d->b_scale == nullptr) != (d->b_shift == nullptr))
      return fail(MT_ERR_ARG, "mt_gemm: im2col affine needs both scale and shift");
  }
  if (d->prologue == MT_PRO_BN_BWD && (!d->A2 || !d->gate)) return fail(MT_ERR_ARG, "mt_gemm: BN_BWD prologue needs A2 and kc");
  if (d->prologue == MT_PRO_BN_BWD && ((uintptr_t)d->A2 & 15)) return fail(MT_ERR_ARG, "mt_gemm: A2 must be 16-byte aligned");
  if (d->b_prologue == MT_BPRO_BN_SWISH_GATE && (d->op != MT_OP_TN || !d->b_scale || !d->b_shift || !d->b_gate))
    return fail(MT_ERR_ARG, "mt_gemm: B prologue needs op TN and b_scale/b_shift/b_gate");
  if (d->b_prologue == MT_BPRO_IM2COL && d->op != MT_OP_TN) return fail(MT_ERR_ARG, "mt_gemm: im2col B prologue needs op TN");
  if (d->prologue == MT_PRO_BN_SWISH_GATE && !d->gate) return fail(MT_ERR_ARG, "mt_gemm: gate prologue needs gate");
  if (d->epilogue == MT_EPI_SE_RED || d->epilogue == MT_EPI_ACT_BWD) {
    if (d->op != MT_OP_NN || d->prologue != MT_PRO_BN_BWD) return fail(MT_ERR_UNSUPPORTED, "mt_gemm: SE_RED / ACT_BWD need op NN with the BN_BWD prologue");
    if (!d->C2 || !d->e_scale || !d->e_shift || d->e_hw <= 0 || d->c_map.gin != 0)
      return fail(MT_ERR_ARG, "mt_gemm: SE_RED / ACT_BWD need C2 (z), e_scale, e_shift, e_hw > 0 and identity output rows");
    if (d->epilogue == MT_EPI_ACT_BWD && (!d->e_gate || !d->e_dpool || !d->e_mi || !d->stats))
      return fail(MT_ERR_ARG, "mt_gemm: ACT_BWD needs e_gate, e_dpool, e_mi and stats");
  }

  // K-contiguous operands need K % 4 == 0 etc. were checked above; plain problems take the LDS-DMA pipeline when it has an instance
  if (!g_trace) {
    int rc = try_launch_split(d, a, s);
    if (rc <= 0) return rc;
    rc = try_launch_dma(d, a, s);
    if (rc <= 0) return rc;
  }
  int cfg = pick_cfg(d->op, d->M, d->N, d->prologue, d->epilogue);
  static const int wg64_var = getenv("MT_CONV_WG64") ? atoi(getenv("MT_CONV_WG64")) : 2;      // lab: 0 = the 128 x 64 tile, 1 = six wavefronts
  if (d->op == MT_OP_TN && d->b_prologue == MT_BPRO_IM2COL && d->epilogue == MT_EPI_ATOMIC && d->M <= 64 && d->N > 192 && d->N <= 288 && wg64_var != 0)
    cfg = (det_enabled() || wg64_var == 1) ? CFG_WG64 : CFG_WG64K;   // (deterministic mode writes per-split slabs: one writer per tile)
  const int m_tiles = (d->M + kCfg[cfg].bm - 1) / kCfg[cfg].bm;
  const int n_tiles = (d->N + kCfg[cfg].bn - 1) / kCfg[cfg].bn;
  dim3 grid(m_tiles * n_tiles, 1, 1);
  a.group_n = 0;
  a.trace = g_trace;
  if (m_tiles >= 32 && n_tiles >= 2 && !getenv("MT_NO_L2_BLOCKING")) {
    // size a column group so its B panels take ~2 MB of the XCD's 4 MB L2
    const int64_t panel = (int64_t)kCfg[cfg].bn * d->K * 4;
    int gn = (int)((2 << 20) / (panel > 0 ? panel : 1));
    if (gn < 1) gn = 1;
    if (gn > n_tiles) gn = n_tiles;
    a.group_n = gn;
    const int max_rows = (m_tiles + 7) / 8;
    grid.x = 8 * max_rows * n_tiles;
  }

  // im2col prologues: the granule form (gemm_core.hpp) for float images with C % 4 == 0 and k <= 5, else the per-element gather
  static const bool im_any = getenv("MT_IM2COL_ANY") != nullptr;                  // lab: the per-element gather everywhere
  const bool im_granule = (d->conv_C & 3) == 0 && !d->conv_src_u8 && d->conv_k * d->conv_k <= 32 && !im_any;
#define COMBO(OP, AL, BL, PRO, EPI)                                                        \
  if (d->op == OP && d->prologue == PRO && d->epilogue == EPI)                             \
    return launch<AL, BL, PRO, EPI>(cfg, a, grid, s);

  if (d->op == MT_OP_TN) {
    int splits = d->split_k;
    if (splits <= 0) {
      // auto: enough blocks to fill 256 CUs several times over (these outputs are skinny: a handful of tiles), but keep
      // >= 256 contraction rows per block so the fp32 atomics of the epilogue stay a small fraction of the work
      const int tiles = m_tiles * n_tiles;
      static const int target = getenv("MT_WGRAD_BLOCKS_OLD") ? atoi(getenv("MT_WGRAD_BLOCKS_OLD")) : 2048;   // tuning knobs
      static const int min_rows = getenv("MT_WGRAD_MINROWS_OLD") ? atoi(getenv("MT_WGRAD_MINROWS_OLD")) : 256;
      splits = (target + tiles - 1) / tiles;
      const int max_splits = d->K / min_rows > 0 ? d->K / min_rows : 1;
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
    int chunk = (d->K + splits - 1) / splits;
    chunk = (chunk + MT_BK - 1) / MT_BK * MT_BK;
    splits = (d->K + chunk - 1) / chunk;
    a.k_chunk = chunk;
    grid.y = splits;
    if (d->epilogue == MT_EPI_ATOMIC)
      if (int rc = det_gemm_setup(a.C, a.ldc, a.det_slab, d->M, d->N, splits, a.c_map.gin != 0, s)) return rc;
    if (d->b_prologue == MT_BPRO_BN_SWISH_GATE) {
      if (d->prologue == MT_PRO_BN_BWD && d->epilogue == MT_EPI_ATOMIC)
        return launch<LAYOUT_KMAJOR, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_ATOMIC, BPRO_BN_SWISH_GATE>(cfg, a, grid, s);
      return fail(MT_ERR_UNSUPPORTED, "mt_gemm TN: B prologue only with BN_BWD/ATOMIC");
    }
    if (d->b_prologue == MT_BPRO_IM2COL) {
      if (d->prologue == MT_PRO_BN_BWD && d->epilogue == MT_EPI_ATOMIC)
        return im_granule ? launch<LAYOUT_KMAJOR, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_ATOMIC, BPRO_IM2COL>(cfg, a, grid, s)
                          : launch<LAYOUT_KMAJOR, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_ATOMIC, BPRO_IM2COL_ANY>(cfg, a, grid, s);
      return fail(MT_ERR_UNSUPPORTED, "mt_gemm TN: im2col B prologue only with BN_BWD/ATOMIC");
    }
    COMBO(MT_OP_TN, LAYOUT_KMAJOR, LAYOUT_KMAJOR, PRO_NONE, EPI_ATOMIC)
    COMBO(MT_OP_TN, LAYOUT_KMAJOR, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_ATOMIC)
    return fail(MT_ERR_UNSUPPORTED, "mt_gemm TN: unsupported prologue/epilogue %d/%d", d->prologue, d->epilogue);
  }
  if (d->epilogue == MT_EPI_ATOMIC && (d->op == MT_OP_NT || d->op == MT_OP_NN)) {
    // split-K with fp32 atomics into a caller-initialised C: evens out the tail of skinny problems (e.g. 396 tiles on 256 CUs)
    int splits = d->split_k > 0 ? d->split_k : 1;
    int chunk = (d->K + splits - 1) / splits;
    chunk = (chunk + MT_BK - 1) / MT_BK * MT_BK;
    splits = (d->K + chunk - 1) / chunk;
    a.k_chunk = chunk;
    grid.y = splits;
    if (int rc = det_gemm_setup(a.C, a.ldc, a.det_slab, d->M, d->N, splits, a.c_map.gin != 0, s)) return rc;
    COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_ATOMIC)
    COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_NONE, EPI_ATOMIC)
    return fail(MT_ERR_UNSUPPORTED, "mt_gemm: split-K atomic epilogue only without prologue");
  }
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_STORE)
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_BIAS_RES)
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_GEGLU)
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_STATS)
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_NONE, EPI_GEGLU_BWD)   // FF data gradient over a transposed weight: the fallback of the split loop's instance
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_BN_SWISH_GATE, EPI_STATS)
  COMBO(MT_OP_NT, LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_BN_SWISH_GATE, EPI_STORE)
  if (d->op == MT_OP_NT && d->prologue == MT_PRO_IM2COL && d->epilogue == MT_EPI_STORE)
    return im_granule ? launch<LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_IM2COL, EPI_STORE>(cfg, a, grid, s)
                      : launch<LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_IM2COL_ANY, EPI_STORE>(cfg, a, grid, s);
  if (d->op == MT_OP_NT && d->prologue == MT_PRO_IM2COL && d->epilogue == MT_EPI_STATS)
    return im_granule ? launch<LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_IM2COL, EPI_STATS>(cfg, a, grid, s)
                      : launch<LAYOUT_KCONTIG, LAYOUT_KCONTIG, PRO_IM2COL_ANY, EPI_STATS>(cfg, a, grid, s);
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_NONE, EPI_STORE)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_NONE, EPI_ACCUM)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_NONE, EPI_GEGLU_BWD)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_STORE)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_BIAS_RES)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_SE_RED)
  COMBO(MT_OP_NN, LAYOUT_KCONTIG, LAYOUT_KMAJOR, PRO_BN_BWD, EPI_ACT_BWD)
#undef COMBO
  return fail(MT_ERR_UNSUPPORTED, "mt_gemm: unsupported op/prologue/epilogue %d/%d/%d", d->op, d->prologue, d->epilogue);
}

extern "C" int mt_gemm(const mt_gemm_desc* d, void* stream) {
  const int rc = gemm_impl(d, stream);
  if (rc) { (void)mt::det_gemm_finish(nullptr, false); return rc; }
  return mt::det_gemm_finish((hipStream_t)stream, true);       // deterministic mode: split-K slabs -> C in split order (det.hpp)
}

extern "C" int mt_version(void) { return MT_VERSION; }
extern "C" const char* mt_last_error(void) { return mt::err_buf(); }
