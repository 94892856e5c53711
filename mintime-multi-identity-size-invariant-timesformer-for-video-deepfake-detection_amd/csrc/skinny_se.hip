// Squeeze-excite stage of the MBConv reverse walk for the early blocks (EfficientNet-B0 blocks 0-2: 0.8-3.2 M rows, project conv
// 32/96/144 -> 16/24 channels) WITHOUT the project conv's data gradient in memory (autograd of efficientnet_pytorch/model.py:104-117
// driven by train.py:371).
//
//   dz_p[r, o] = ka[o]*du_p[r, o] + kb[o]*z_p[r, o] + kc[o]            BatchNorm (bn2) backward folded into the load
//   da[r, c]   = sum_o dz_p[r, o] * Wp[o, c]                           project conv data gradient: K = 16 / 24, a few MFMAs per tile
//   u = z_d[r, c]*scale[c] + shift[c]
//   MODE_RED:  dgate[r / hw, c] += da[r, c] * swish(u)                 (d of the squeeze-excite gate)
//   MODE_ACT:  du_d[r, c] = (da[r, c]*gate[r / hw, c] + dpool[r / hw, c]/hw) * swish'(u)    + the bn1 backward sums of du_d
//
// The materialised form runs three launches over the expanded tensor -- data gradient (writes da), reduction (reads da, z_d),
// activation backward (reads da, z_d, writes du_d): 6 passes.  Recomputing da from the NARROW gradient costs (Cout_p / 2) MFMA steps
// per 32 x 32 tile, so both consumers can rebuild it: 3 passes (read z_d; read z_d, write du_d) and no da tensor.  The same idea as
// GEMM epilogues (MT_EPI_SE_RED / MT_EPI_ACT_BWD) lost to the streaming kernels (profiles/r03_se_fused_epilogue_vs_streaming.txt)
// because a one-tile-per-block GEMM touches z_d per lane-column behind its MFMAs; here it IS a streaming kernel: persistent blocks,
// 64-row chunks, z_d and du_d as float4 rows with the next chunk's loads in flight, da handed from the accumulators to the
// row-major float4 phase through LDS.
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SeArgs {
  const float* du; const float* z; const float* kabc;        // narrow: [rows, Co] x2, [3, Co]
  const float* w;                                              // [Co, C] project weight (row o = the C taps of output channel o)
  const float* zd; const float* scale; const float* shift;     // [rows, C], [C], [C]
  float* dgate;                                                // RED: [rows / hw, C], atomically accumulated (zero-filled by the caller)
  const float* gate; const float* dpool; const float* mi;      // ACT: [rows / hw, C] x2, mean | invstd [2][C]
  float* dud; double* stats; int slots;                        // ACT: [rows, C]; [slots][2][C]
  int64_t rows; int Co, C, hw;
};

__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   /* v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the swish kernels are VALU-bound */

enum { MODE_RED = 0, MODE_ACT = 1 };

// NT = 32-wide tiles of C (1, 3, 5), CQB = C / 4 (8, 24, 36).  Thread (cql, pl): column quad cql, row lane pl of the PB = 256 / CQB lanes.
template <int NT, int CQB, int MODE>
__global__ __launch_bounds__(256) void se_stage_kernel(SeArgs p) {
  constexpr int R = 64;
  constexpr int LDP = 33;                  // dz_p tile [R][33]: read by row (lanes = rows) for the MFMA A operand
  constexpr int LDW = NT * 32 + 4;         // W tile [32 k][C + 4]: B operand, lanes over c
  constexpr int LDA = NT * 32 + 4;         // da tile [R][C + 4]: written by column lanes, read back as float4 rows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dzs = smem;                       // [R][LDP]
  float* wt = dzs + R * LDP;               // [32][LDW]
  float* das = wt + 32 * LDW;              // [R][LDA]
  float* red = das;                        // end of kernel (ACT): stats reduction [PB][CQB][8]; per flush (RED): [PB][CQB][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PB = 256 / CQB;
  const int cql = tid % CQB, pl = tid / CQB;
  const bool on = pl < PB;
  const int c4 = cql * 4;
  constexpr int VS = (R + PB - 1) / PB;    // row slots per thread (2, 7, 10)
  const int oq = p.Co >> 2;                // narrow float4 per row (4 or 6)

  for (int i = tid; i < R * LDP + 32 * LDW; i += 256) smem[i] = 0.f;       // k rows / columns past Co stay zero
  __syncthreads();
  for (int i = tid; i < p.Co * p.C; i += 256) {
    const int o = i / p.C, c = i - o * p.C;
    wt[o * LDW + c] = p.w[i];
  }
  const float4 sc = on ? *reinterpret_cast<const float4*>(p.scale + c4) : make_float4(0, 0, 0, 0);
  const float4 sh = on ? *reinterpret_cast<const float4*>(p.shift + c4) : make_float4(0, 0, 0, 0);
  float4 mean = make_float4(0, 0, 0, 0), istd = mean;
  if (MODE == MODE_ACT && on) { mean = *reinterpret_cast<const float4*>(p.mi + c4); istd = *reinterpret_cast<const float4*>(p.mi + p.C + c4); }
  const float inv_hw = 1.0f / (float)p.hw;

  // narrow operand: R * oq float4 per tensor (256 or 384): two slots per thread
  int nr[2], nc[2];
  bool n_on[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    nr[i] = idx / oq;
    nc[i] = (idx - nr[i] * oq) * 4;
    n_on[i] = nr[i] < R;
    if (!n_on[i]) { nr[i] = 0; nc[i] = 0; }
  }
  float4 ka[2], kb[2], kc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ka[i] = *reinterpret_cast<const float4*>(p.kabc + nc[i]);
    kb[i] = *reinterpret_cast<const float4*>(p.kabc + p.Co + nc[i]);
    kc[i] = *reinterpret_cast<const float4*>(p.kabc + 2 * p.Co + nc[i]);
  }

  // contiguous chunk ranges per block: a block stays inside one image for long runs (RED flushes its sums on image changes only)
  const int64_t nchunks = (p.rows + R - 1) / R;
  const int64_t per = (nchunks + gridDim.x - 1) / gridDim.x;
  const int64_t c_lo = (int64_t)blockIdx.x * per, c_hi = c_lo + per < nchunks ? c_lo + per : nchunks;

  float4 rdu[2], rz[2], rzd[VS];
  auto fetch_narrow = [&](int64_t chunk) {
    const int64_t r0 = chunk * R;
    const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t off = (r0 + min(nr[i], last)) * p.Co + nc[i];
      rdu[i] = *reinterpret_cast<const float4*>(p.du + off);
      rz[i] = *reinterpret_cast<const float4*>(p.z + off);
    }
  };
  auto fetch_wide = [&](int64_t chunk) {
    const int64_t r0 = chunk * R;
    const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
#pragma unroll
    for (int i = 0; i < VS; ++i) {
      const int row = min(pl + i * PB, last);
      rzd[i] = *reinterpret_cast<const float4*>(p.zd + (r0 + row) * p.C + (on ? c4 : 0));
    }
  };

  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;     // ACT: bn1 backward sums; RED: the running d-gate sum of the current image (s1)
  int64_t cur_img = -1;
  auto flush_gate = [&]() {                          // RED: block-reduce the row lanes' sums, one atomic per column
    __syncthreads();
    if (on) *reinterpret_cast<float4*>(red + (pl * CQB + cql) * 4) = s1;
    __syncthreads();
    if (tid < p.C) {
      const int q = tid >> 2, e = tid & 3;
      float v = 0.f;
      for (int l = 0; l < PB; ++l) v += red[(l * CQB + q) * 4 + e];
      atomicAdd(p.dgate + cur_img * p.C + tid, v);
    }
    __syncthreads();
    s1 = make_float4(0, 0, 0, 0);
  };

  if (c_lo < c_hi) { fetch_narrow(c_lo); fetch_wide(c_lo); }
  __syncthreads();                                   // W in place
  const int kh = lane >> 5, cl = lane & 31;
  for (int64_t chunk = c_lo; chunk < c_hi; ++chunk) {
    const int64_t r0 = chunk * R;
    const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);
    const int64_t img = div_rows(r0, p.hw);                   // hw % R == 0: a chunk lies inside one image
    if (MODE == MODE_RED && img != cur_img) {
      if (cur_img >= 0) flush_gate();
      cur_img = img;
    }
    // ---- narrow tile -> LDS (BatchNorm backward applied)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (n_on[i]) {
        const bool ok = nr[i] < left;
        float* dst = dzs + nr[i] * LDP + nc[i];
        dst[0] = ok ? fmaf(ka[i].x, rdu[i].x, fmaf(kb[i].x, rz[i].x, kc[i].x)) : 0.f;
        dst[1] = ok ? fmaf(ka[i].y, rdu[i].y, fmaf(kb[i].y, rz[i].y, kc[i].y)) : 0.f;
        dst[2] = ok ? fmaf(ka[i].z, rdu[i].z, fmaf(kb[i].z, rz[i].z, kc[i].z)) : 0.f;
        dst[3] = ok ? fmaf(ka[i].w, rdu[i].w, fmaf(kb[i].w, rz[i].w, kc[i].w)) : 0.f;
      }
    __syncthreads();
    if (chunk + 1 < c_hi) fetch_narrow(chunk + 1);   // small; in flight through the rest of the chunk
    // ---- da tiles: (row tile rt, column tile ct) pairs dealt to the 4 wavefronts; K = 32 (zero-padded past Co)
    for (int t = wave; t < 2 * NT; t += 4) {
      const int rt = t & 1, ct = t >> 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* a_w = dzs + (rt * 32 + cl) * LDP + kh;
      const float* b_w = wt + kh * LDW + ct * 32 + cl;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[2 * ks], b_w[2 * ks * LDW], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) das[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * LDA + ct * 32 + cl] = acc[r];
    }
    __syncthreads();
    // ---- row-major float4 phase on this thread's column quad
    float4 g = make_float4(0, 0, 0, 0), dp = g;
    if (MODE == MODE_ACT && on) {
      g = *reinterpret_cast<const float4*>(p.gate + img * p.C + c4);
      const float4 t = *reinterpret_cast<const float4*>(p.dpool + img * p.C + c4);
      dp = make_float4(t.x * inv_hw, t.y * inv_hw, t.z * inv_hw, t.w * inv_hw);
    }
    if (on) {
#pragma unroll
      for (int i = 0; i < VS; ++i) {
        const int row = pl + i * PB;
        if (row < left) {
          const float4 a = *reinterpret_cast<const float4*>(das + row * LDA + c4);
          const float4 zz = rzd[i];
          const float ux = fmaf(zz.x, sc.x, sh.x), uy = fmaf(zz.y, sc.y, sh.y), uz = fmaf(zz.z, sc.z, sh.z), uw = fmaf(zz.w, sc.w, sh.w);
          const float sx = sigmoid_(ux), sy = sigmoid_(uy), sz = sigmoid_(uz), sw = sigmoid_(uw);
          if (MODE == MODE_RED) {
            s1.x = fmaf(a.x, ux * sx, s1.x); s1.y = fmaf(a.y, uy * sy, s1.y); s1.z = fmaf(a.z, uz * sz, s1.z); s1.w = fmaf(a.w, uw * sw, s1.w);
          } else {
            float4 d;
            d.x = fmaf(a.x, g.x, dp.x) * (sx * (1.0f + ux * (1.0f - sx)));
            d.y = fmaf(a.y, g.y, dp.y) * (sy * (1.0f + uy * (1.0f - sy)));
            d.z = fmaf(a.z, g.z, dp.z) * (sz * (1.0f + uz * (1.0f - sz)));
            d.w = fmaf(a.w, g.w, dp.w) * (sw * (1.0f + uw * (1.0f - sw)));
            *reinterpret_cast<float4*>(p.dud + (r0 + row) * p.C + c4) = d;
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = fmaf(d.x, (zz.x - mean.x) * istd.x, s2.x); s2.y = fmaf(d.y, (zz.y - mean.y) * istd.y, s2.y);
            s2.z = fmaf(d.z, (zz.z - mean.z) * istd.z, s2.z); s2.w = fmaf(d.w, (zz.w - mean.w) * istd.w, s2.w);
          }
        }
      }
    }
    if (chunk + 1 < c_hi) fetch_wide(chunk + 1);     // the registers are free again: in flight through the next chunk's MFMA phase
  }
  if (MODE == MODE_RED) {
    if (cur_img >= 0) flush_gate();
    return;
  }
  // ---- ACT: bn1 backward sums -> fp64 slots (block reduction over the row lanes first)
  __syncthreads();
  if (on) {
    float* rr = red + (pl * CQB + cql) * 8;
    rr[0] = s1.x; rr[1] = s1.y; rr[2] = s1.z; rr[3] = s1.w; rr[4] = s2.x; rr[5] = s2.y; rr[6] = s2.z; rr[7] = s2.w;
  }
  __syncthreads();
  for (int i = tid; i < CQB * 8; i += 256) {
    const int q = i >> 3, e = i & 7;
    float v = 0.f;
    for (int l = 0; l < PB; ++l) v += red[(l * CQB + q) * 8 + e];
    const int ch = q * 4 + (e & 3), which = e >> 2;
    stat_add(p.stats + ((int64_t)(blockIdx.x % stat_slots(p.slots)) * 2 + which) * p.C + ch, stat_limb(p.slots, p.C), v);
  }
}

template <int NT>
int launch_se(const SeArgs& a, int mode, hipStream_t st) {
  constexpr int R = 64;
  const size_t smem = ((size_t)R * 33 + 32 * (NT * 32 + 4) + R * (NT * 32 + 4)) * 4;
  const int64_t nchunks = (a.rows + R - 1) / R;
  const int blocks = (int)(nchunks < 512 ? nchunks : 512);
  auto go = [&](auto k) {
    (void)ensure_dynamic_lds((const void*)k, smem);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), smem, st, a);
  };
  constexpr int CQB = NT == 1 ? 8 : (NT == 3 ? 24 : 36);
  if (mode == MODE_RED) go(se_stage_kernel<NT, CQB, MODE_RED>);
  else go(se_stage_kernel<NT, CQB, MODE_ACT>);
  return check_launch("mt_se_stage_fused");
}

}  // namespace

extern "C" int mt_se_stage_fused_supported(int Co, int C, int hw) {
  if (Co <= 0 || Co > 32 || (Co & 3) || hw <= 0 || (hw % 64)) return 0;
  return C == 32 || C == 96 || C == 144;
}

extern "C" int mt_se_stage_fused(const float* du_p, const float* z_p, const float* kabc_p, const float* w_p, const float* z_d,
                                 const float* scale_d, const float* shift_d, int mode, float* dgate, const float* gate,
                                 const float* dpooled, const float* mean_invstd_d, float* du_d, double* stats, int slots,
                                 int64_t rows, int Co, int C, int hw, void* stream) {
  if (!du_p || !z_p || !kabc_p || !w_p || !z_d || !scale_d || !shift_d) return fail(MT_ERR_ARG, "mt_se_stage_fused: null pointer");
  if (!mt_se_stage_fused_supported(Co, C, hw)) return fail(MT_ERR_UNSUPPORTED, "mt_se_stage_fused: no instance for %d -> %d channels, hw %d", C, Co, hw);
  if (mode == 0 && !dgate) return fail(MT_ERR_ARG, "mt_se_stage_fused: reduction mode needs dgate");
  if (mode == 1 && (!gate || !dpooled || !mean_invstd_d || !du_d || !stats || slots == 0))
    return fail(MT_ERR_ARG, "mt_se_stage_fused: activation mode needs gate, dpooled, mean_invstd, du_d and stats");
  if (mode != 0 && mode != 1) return fail(MT_ERR_ARG, "mt_se_stage_fused: bad mode");
  if (rows % hw) return fail(MT_ERR_ARG, "mt_se_stage_fused: rows must be a multiple of hw");
  SeArgs a{du_p, z_p, kabc_p, w_p, z_d, scale_d, shift_d, dgate, gate, dpooled, mean_invstd_d, du_d, stats, slots, rows, Co, C, hw};
  hipStream_t st = (hipStream_t)stream;
  if (C == 32) return launch_se<1>(a, mode, st);
  if (C == 96) return launch_se<3>(a, mode, st);
  return launch_se<5>(a, mode, st);
}
