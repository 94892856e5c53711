// Plane-operand GEMM (gemm_planes.hpp) behind the C ABI: mt_gemm_planes, and the fp32 -> blocked-plane converters
// (mt_split_planes_blk, mt_split_planes_blk_multi) for tensors that no producer kernel emits as planes (the Linear weights once per
// step, the head's gradient of the residual stream).
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "gemm_planes.hpp"
#include "det.hpp"
#include <stdlib.h>
#include <string.h>

using namespace mt;

namespace {

// fp32 row-major [R][C] (leading dimension ld) -> blocked planes, zero padded.  One wavefront per 32 x 16 block: lane = (row, half),
// its 16-byte stores make whole 1 KB blocks; a 256-thread workgroup takes four adjacent column blocks (256 B of every row).
__device__ __forceinline__ void to_blk_block(const float* __restrict__ src, int64_t ld, int R, int C, const PlaneRef& o, int rb, int cb, int lane) {
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    const float* s = src + (int64_t)r * ld + c;
    if (c + 8 <= C && ((ld & 3) == 0) && (((uintptr_t)src & 15) == 0)) {
      const float4 u = *reinterpret_cast<const float4*>(s), v = *reinterpret_cast<const float4*>(s + 4);
      x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (c + e < C) x[e] = s[e];
    }
  }
  planes_store8(o, r, c, x);
}

__global__ __launch_bounds__(256) void split_planes_blk_kernel(const float* __restrict__ src, int64_t ld, int R, int C, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  to_blk_block(src, ld, R, C, o, rb, cb, lane);
}

// BatchNorm-backward affine dz = ka * du + kb * z + kc (per column; kabc = [3][C]) written straight as planes: the operand of a
// convolution's data- and weight-gradient GEMMs, evaluated once instead of in both GEMMs' prologues
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float* __restrict__ du, const float* __restrict__ z,
                                                                  const float* __restrict__ kabc, int R, int C, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      if (c + e + 4 <= C) {
        const float4 a = *reinterpret_cast<const float4*>(du + (int64_t)r * C + c + e), b = *reinterpret_cast<const float4*>(z + (int64_t)r * C + c + e);
        const float4 ka = *reinterpret_cast<const float4*>(kabc + c + e), kb = *reinterpret_cast<const float4*>(kabc + C + c + e),
                     kc = *reinterpret_cast<const float4*>(kabc + 2 * C + c + e);
        x[e] = fmaf(ka.x, a.x, fmaf(kb.x, b.x, kc.x)); x[e + 1] = fmaf(ka.y, a.y, fmaf(kb.y, b.y, kc.y));
        x[e + 2] = fmaf(ka.z, a.z, fmaf(kb.z, b.z, kc.z)); x[e + 3] = fmaf(ka.w, a.w, fmaf(kb.w, b.w, kc.w));
      }
    }
  }
  planes_store8(o, r, c, x);
}

struct SplitItem { const float* src; __bf16* planes; int64_t rows, cols, first; };   // first = index of the item's first wave-block

__global__ __launch_bounds__(256) void split_planes_blk_multi_kernel(const SplitItem* __restrict__ items, int count, int64_t total) {
  const int64_t wb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wb >= total) return;
  int lo = 0, hi = count - 1;                        // the item whose block range holds wb
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].first <= wb) lo = mid; else hi = mid - 1; }
  const SplitItem it = items[lo];
  const int R = (int)it.rows, Cc = (int)it.cols;
  const int cb16 = (Cc + 15) >> 4, rp = (R + 31) & ~31;
  const int64_t loc = wb - it.first;
  const PlaneRef o{it.planes, (int64_t)rp * cb16 * 16, cb16, rp};
  to_blk_block(it.src, Cc, R, Cc, o, (int)(loc / cb16), (int)(loc % cb16), threadIdx.x & 63);
}

constexpr int kSkMaxGrid = 1024;                      // persistent blocks a stream-K launch may use (workspace is sized for this)
constexpr int64_t kSkFlagBytes = 4096;                // kSkMaxGrid ints, then the slabs

int cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& c = cus[dev & 63];
  if (c == 0) {
    hipDeviceProp_t pr;
    c = hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return c;
}

// persistent form of the fp32-output forward / data-gradient GEMMs: resident blocks per CU (0 = one block per tile, the default:
// QKV 122 -> 114 us standalone, the step unchanged -- profiles/README.md round 5).  MT_PLANES_PERSIST sets the initial value.
int g_persist = -1;
int persist_blocks() {
  if (g_persist < 0) {
    const char* e = getenv("MT_PLANES_PERSIST");
    g_persist = e ? atoi(e) : 0;
    if (g_persist < 0 || g_persist > 4) g_persist = 0;
  }
  return g_persist;
}

template <bool AKM, bool BKM, int EPI, int BAL, bool CPL, int SK = 0>
int launch_planes(const GemmArgs& a, dim3 grid, hipStream_t s) {
  constexpr int ST = 2;
  auto k = gemm_planes_kernel<2, 2, 2, 2, AKM, BKM, EPI, ST, BAL == BAL_PAIR ? 2 : 3, BAL, CPL, SK>;
  constexpr size_t lds = (size_t)ST * 3 * (128 + 128) * 32;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_gemm_planes: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  return check_launch("mt_gemm_planes");
}

}  // namespace

extern "C" int mt_gemm_planes_set_persist(int blocks_per_cu) {
  const int prev = persist_blocks();
  g_persist = blocks_per_cu < 0 || blocks_per_cu > 4 ? 0 : blocks_per_cu;
  return prev;
}

extern "C" int64_t mt_gemm_planes_workspace_bytes(void) { return kSkFlagBytes + (int64_t)kSkMaxGrid * 128 * 128 * 4; }

extern "C" int64_t mt_planes_elems(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (int64_t)((rows + 31) & ~31) * ((cols + 15) & ~15);
}

extern "C" int mt_split_planes_blk(const float* src, int64_t ld, int rows, int cols, void* planes, void* stream) {
  if (!src || !planes) return fail(MT_ERR_ARG, "mt_split_planes_blk: null pointer");
  if (rows <= 0 || cols <= 0 || ld < cols) return fail(MT_ERR_ARG, "mt_split_planes_blk: bad shape %d x %d (ld %lld)", rows, cols, (long long)ld);
  if ((uintptr_t)planes & 15) return fail(MT_ERR_ARG, "mt_split_planes_blk: planes must be 16-byte aligned");
  const int cb16 = (cols + 15) >> 4, rp = (rows + 31) & ~31;
  const PlaneRef o{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
  hipLaunchKernelGGL(split_planes_blk_kernel, dim3((cb16 + 3) / 4, rp / 32), dim3(256), 0, (hipStream_t)stream, src, ld, rows, cols, o);
  return check_launch("mt_split_planes_blk");
}

extern "C" int mt_bn_bwd_apply_planes(const float* du, const float* z, const float* kabc, void* planes, int rows, int C, void* stream) {
  if (!du || !z || !kabc || !planes || rows <= 0 || C <= 0) return fail(MT_ERR_ARG, "mt_bn_bwd_apply_planes: bad arguments");
  if ((C & 3) || (((uintptr_t)du | (uintptr_t)z | (uintptr_t)kabc | (uintptr_t)planes) & 15))
    return fail(MT_ERR_ARG, "mt_bn_bwd_apply_planes: C %% 4 == 0 and 16-byte alignment");
  const int cb16 = (C + 15) >> 4, rp = (rows + 31) & ~31;
  const PlaneRef o{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3((cb16 + 3) / 4, rp / 32), dim3(256), 0, (hipStream_t)stream, du, z, kabc, rows, C, o);
  return check_launch("mt_bn_bwd_apply_planes");
}

extern "C" int mt_split_planes_blk_multi(const void* items, int count, int64_t total_blocks, void* stream) {
  if (!items || count <= 0 || total_blocks <= 0) return fail(MT_ERR_ARG, "mt_split_planes_blk_multi: bad arguments");
  hipLaunchKernelGGL(split_planes_blk_multi_kernel, dim3((unsigned)((total_blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const SplitItem*>(items), count, total_blocks);
  return check_launch("mt_split_planes_blk_multi");
}

static int gemm_planes_impl(const mt_gemm_planes_desc* d, void* stream) {
  if (!d || !d->a_planes || !d->b_planes) return fail(MT_ERR_ARG, "mt_gemm_planes: null operand planes");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return fail(MT_ERR_ARG, "mt_gemm_planes: bad shape %d %d %d", d->M, d->N, d->K);
  if (((uintptr_t)d->a_planes & 15) || ((uintptr_t)d->b_planes & 15) || ((uintptr_t)d->c_planes & 15))
    return fail(MT_ERR_ARG, "mt_gemm_planes: planes must be 16-byte aligned");
  const int op = d->op, epi = d->epilogue;
  const bool cpl = d->c_planes != nullptr;
  if (!d->C && !cpl) return fail(MT_ERR_ARG, "mt_gemm_planes: no output");
  if (cpl && epi != MT_EPI_GEGLU && epi != MT_EPI_GEGLU_BWD) return fail(MT_ERR_UNSUPPORTED, "mt_gemm_planes: plane output only from the GEGLU pair");
  if (!cpl && !d->C) return fail(MT_ERR_ARG, "mt_gemm_planes: C is null");
  if (epi == MT_EPI_BIAS_RES && !d->R) return fail(MT_ERR_ARG, "mt_gemm_planes: BIAS_RES needs R");
  if (epi == MT_EPI_GEGLU && (d->n_half * 2 != d->N || (d->n_half & 63))) return fail(MT_ERR_ARG, "mt_gemm_planes GEGLU: N must be 2*n_half, n_half %% 64 == 0");
  if (epi == MT_EPI_GEGLU_BWD && (!d->C2 || d->n_half != d->N || (d->n_half & 31)))
    return fail(MT_ERR_ARG, "mt_gemm_planes GEGLU_BWD: needs C2, n_half == N and n_half %% 32 == 0");
  hipStream_t s = (hipStream_t)stream;

  // operand geometry as stored: A is [M][K] (NT, NN) or [K][M] (TN); B is [N][K] (NT) or [K][N] (NN, TN)
  const int a_rows = op == MT_OP_TN ? d->K : d->M, a_cols = op == MT_OP_TN ? d->M : d->K;
  const int b_rows = op == MT_OP_NT ? d->N : d->K, b_cols = op == MT_OP_NT ? d->K : d->N;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.C = d->C; a.M = d->M; a.N = d->N; a.K = d->K; a.ldc = d->ldc;
  a.a_planes = d->a_planes; a.a_pstride = mt_planes_elems(a_rows, a_cols); a.lda = (a_cols + 15) >> 4;
  a.b_planes = d->b_planes; a.b_pstride = mt_planes_elems(b_rows, b_cols); a.ldb = (b_cols + 15) >> 4;
  // the loop addresses an operand's three planes by 32-bit byte offsets from one base (gemm_planes.hpp voff): 4 GB per operand
  if (6 * a.a_pstride > 0xFFFFFFFFll || 6 * a.b_pstride > 0xFFFFFFFFll)
    return fail(MT_ERR_UNSUPPORTED, "mt_gemm_planes: an operand's planes span %lld bytes (> 4 GB: 32-bit DMA offsets); use mt_gemm for it (lib.planes_fit)",
                (long long)(6 * (a.a_pstride > a.b_pstride ? a.a_pstride : a.b_pstride)));
  a.bias = d->bias; a.R = d->R; a.ldr = d->ldr; a.C2 = d->C2; a.ldc2 = d->ldc2; a.n_half = d->n_half; a.col_sum = d->col_sum;
  a.hw = 1; a.stats_slots = d->stats_slots != 0 ? d->stats_slots : 1; a.stats = d->stats; a.b_hw = 1; a.e_hw = 1;
  if (epi == MT_EPI_STATS && !d->stats) return fail(MT_ERR_ARG, "mt_gemm_planes: STATS needs stats");
  if (epi == MT_EPI_GEGLU_BWD)
    if (int rc = det_gemm_colsum_setup(a.det, d->M, d->n_half, d->col_sum, s)) return rc;
  if (cpl) {
    const int c_cols = epi == MT_EPI_GEGLU ? d->n_half : 2 * d->n_half;
    a.c_planes = d->c_planes; a.c_pstride = mt_planes_elems(d->M, c_cols); a.ldcp = (c_cols + 15) >> 4;
  }

  const int m_tiles = (d->M + 127) / 128, n_tiles = (d->N + 127) / 128;
  dim3 grid(m_tiles * n_tiles, 1, 1);
  if (op == MT_OP_TN) {
    if (epi != MT_EPI_ATOMIC) return fail(MT_ERR_UNSUPPORTED, "mt_gemm_planes TN: ATOMIC epilogue only (C pre-zeroed)");
    // K-range-major split-K (gemm_split.hpp): a multiple of 8 ranges, m_tiles * n_tiles blocks per range
    int splits = d->split_k;
    if (splits <= 0) {
      static const int target = getenv("MT_WGRAD_BLOCKS") ? atoi(getenv("MT_WGRAD_BLOCKS")) : 640;
      splits = (target + (int)grid.x - 1) / (int)grid.x;
      const int max_splits = d->K / 256 > 0 ? d->K / 256 : 1;
      if (splits > max_splits) splits = max_splits;
    }
    splits = (splits + 4) / 8 * 8;
    if (splits < 8) splits = 8;
    int chunk = (d->K + splits - 1) / splits;
    chunk = (chunk + 15) / 16 * 16;
    a.k_chunk = chunk; a.xcd_k = 1;
    grid.y = (unsigned)(((d->K + chunk - 1) / chunk + 7) / 8 * 8);
    if (int rc = det_gemm_setup(a.C, a.ldc, a.det_slab, d->M, d->N, (d->K + chunk - 1) / chunk, false, s)) return rc;
    return launch_planes<true, true, EPI_ATOMIC, BAL_NONE, false, 0>(a, grid, s);
  }
  {
    // forward / data-gradient GEMMs sit on the critical queue, the weight gradients they share the matrix cores with do not
    static const int prio = getenv("MT_PLANES_MAIN_PRIO") ? atoi(getenv("MT_PLANES_MAIN_PRIO")) : 0;
    a.wave_prio = prio;
  }
  if (m_tiles >= 32 && n_tiles >= 2 && !getenv("MT_NO_L2_BLOCKING")) {
    const int64_t panel = (int64_t)128 * d->K * 6;   // one column group's B panels: three bf16 planes
    static const int64_t group_bytes = getenv("MT_PLANES_GROUP_KB") ? (int64_t)atoi(getenv("MT_PLANES_GROUP_KB")) << 10 : (2 << 20);
    int gn = (int)(group_bytes / (panel > 0 ? panel : 1));
    if (gn < 1) gn = 1;
    if (gn > n_tiles) gn = n_tiles;
    a.group_n = gn;
    grid.x = 8 * ((m_tiles + 7) / 8) * n_tiles;
  }
  // stream-K over a persistent grid (gemm_planes.hpp) when the caller lends a workspace: two resident blocks per CU share the
  // linearised (tile, k-step) list evenly.  Tiny problems (less than ~4 k-steps per block) keep one block per tile.
  const int sk_env = 1;   // (the caller decides by lending a workspace; measured slower than one block per tile on the TimeSformer's shapes: the loop is power-bound and a tile tail's idle CUs give their power to the busy ones)
  bool sk = false;
  if (sk_env && d->sk_workspace && d->sk_workspace_bytes >= mt_gemm_planes_workspace_bytes() && !((uintptr_t)d->sk_workspace & 255)) {
    static const int per_cu = getenv("MT_PLANES_SK_BLOCKS") ? atoi(getenv("MT_PLANES_SK_BLOCKS")) : 2;
    int g = cu_count() * per_cu;
    g = g / 8 * 8;
    if (g > kSkMaxGrid) g = kSkMaxGrid;
    const int64_t units = (int64_t)m_tiles * n_tiles * (((d->K + 15) / 16 + 1) / 2);     // pairs of k-steps
    if (g >= 8 && units >= (int64_t)g * 2 && units < (1ll << 30)) {
      sk = true;
      grid = dim3((unsigned)g, 1, 1);
      a.sk_on = 1;
      a.sk_flags = reinterpret_cast<int*>(d->sk_workspace);
      a.sk_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->sk_workspace) + kSkFlagBytes);
    }
  }
  // persistent blocks with the next tile's first stage prefetched under the epilogue (gemm_planes.hpp SKM = 2): fp32-output forward /
  // data-gradient GEMMs with more tiles than resident block slots.  MT_PLANES_PERSIST: 0 = off, N = blocks per CU (default below).
  if (!sk && !cpl && (epi == MT_EPI_STORE || epi == MT_EPI_BIAS_RES)) {
    const int per_cu = persist_blocks();
    const int g = cu_count() * per_cu / 8 * 8;
    if (per_cu > 0 && g >= 8 && m_tiles * n_tiles > g) {
      const dim3 pg((unsigned)g, 1, 1);
      if (op == MT_OP_NT && epi == MT_EPI_STORE) return launch_planes<false, false, EPI_STORE, BAL_PAIR, false, 2>(a, pg, s);
      if (op == MT_OP_NT && epi == MT_EPI_BIAS_RES) return launch_planes<false, false, EPI_BIAS_RES, BAL_PAIR, false, 2>(a, pg, s);
      if (op == MT_OP_NN && epi == MT_EPI_STORE) return launch_planes<false, true, EPI_STORE, BAL_PAIR, false, 2>(a, pg, s);
    }
  }
#define PL_COMBO(OP, BKM_, EPI_, CPL_) \
  if (op == OP && epi == EPI_ && cpl == CPL_)                                                       \
    return sk ? launch_planes<false, BKM_, EPI_, BAL_PAIR, CPL_, 1>(a, grid, s)                     \
              : launch_planes<false, BKM_, EPI_, BAL_PAIR, CPL_, 0>(a, grid, s);
  PL_COMBO(MT_OP_NT, false, EPI_STORE, false)
  PL_COMBO(MT_OP_NT, false, EPI_BIAS_RES, false)
  PL_COMBO(MT_OP_NT, false, EPI_STATS, false)
  PL_COMBO(MT_OP_NT, false, EPI_GEGLU, false)
  PL_COMBO(MT_OP_NT, false, EPI_GEGLU, true)
  PL_COMBO(MT_OP_NN, true, EPI_STORE, false)
  PL_COMBO(MT_OP_NN, true, EPI_BIAS_RES, false)
  PL_COMBO(MT_OP_NN, true, EPI_GEGLU_BWD, false)
  PL_COMBO(MT_OP_NN, true, EPI_GEGLU_BWD, true)
#undef PL_COMBO
  return fail(MT_ERR_UNSUPPORTED, "mt_gemm_planes: unsupported op / epilogue %d / %d", op, epi);
}

extern "C" int mt_gemm_planes(const mt_gemm_planes_desc* d, void* stream) {
  const int rc = gemm_planes_impl(d, stream);
  if (rc) { (void)mt::det_gemm_finish(nullptr, false); return rc; }
  return mt::det_gemm_finish((hipStream_t)stream, true);       // deterministic mode: split-K slabs / column-sum log -> their targets (det.hpp)
}
