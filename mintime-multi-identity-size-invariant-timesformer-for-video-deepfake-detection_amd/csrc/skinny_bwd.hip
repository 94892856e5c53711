// Data AND weight gradient of a 1x1 "expand" convolution with few input channels and very many rows in ONE streaming pass
// (EfficientNet-B0 blocks 1-3 at 112^2 / 56^2: 0.8-3.2 M rows, 96 / 144 expanded channels against 16 / 24 block channels;
// reference efficientnet_pytorch/model.py:96-99 driven backwards by train.py:371).
//
//   dz[r, co]  = ka[co]*du[r, co] + kb[co]*z[r, co] + kc[co]        BatchNorm backward folded into the load (never stored)
//   dx[r, ci]  = sum_co dz[r, co] * W[co, ci]  (+ res[r, ci])       data gradient  -> the block input's gradient
//   dW[co, ci] += sum_r dz[r, co] * x[r, ci]                        weight gradient
//
// Both gradients need the same dz tile; as two launches (mt_conv1x1_rows mode 2 + mt_conv1x1_wgrad) each streamed du and z -- the
// two widest tensors of the backward pass (1.2 GB each for block 1 of a 256-crop batch) -- once: 4 passes over the expanded
// tensor.  Here a block streams 64-row chunks (loads of chunk i+1 in flight while chunk i is multiplied) and its four wavefronts
// split the work by ROLE: waves 0, 1 own a 32-row tile each and produce dx (M = rows, K = Cout, W resident in LDS), waves 2, 3
// each accumulate the weight gradient of 32 of the chunk's rows (K = rows, the result resident in MFMA accumulators for the whole
// launch).  Both roles cost (Cout / 2) 32-row MFMA steps per chunk, so the waves stay balanced.  2 passes instead of 4.  Measured on
// the 256-crop batch: 2.09 ms (three data-gradient + three weight-gradient launches) -> 1.32 ms (three fused launches);
// algorithmic bytes = rows * (2*Cout + 2*Cin [+ Cin for the residual]) * 4.
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FusedArgs {
  const float* du; const float* z; const float* kabc;      // [rows, Cout] x2, [3, Cout]
  const float* x;                                            // [rows, Cin]
  const float* w;                                            // [Cout, Cin] (the forward weight)
  const float* res;                                          // optional [rows, Cin]
  float* dx;                                                 // [rows, Cin]
  float* dw;                                                 // [Cout, Cin], accumulated with atomics
  int64_t rows; int Cout, Cin;
};

// MT = 32-wide tiles of Cout (3: 96 channels, 5: 144 channels); Cin <= 32 (one tile)
// NTHR = 256, or 512 for MT = 5: waves 4-7 only load and stage (the 10 float4 prefetch slots per thread of a 256-thread block
// put the kernel at 296 VGPRs = one wave per SIMD; with 512 threads it is 5 slots and two waves per SIMD)
// ROWSPLIT: the two weight-gradient waves split the chunk's ROWS (all MT tiles each, partial results meet in LDS at the end) instead
// of the Cout tiles.  Measured (256-crop batch, us per launch, row split / tile split): block 1 (96 -> 16, 3.2 M rows) 675 / 1151;
// blocks 2, 3 (144 -> 24, 0.8 M rows, 512 threads) 309-346 / 343-365.  Both instances use the row split; the tile split stays
// compilable (it is what fits 144 channels into 256 VGPRs with a 256-thread block).
template <int MT, int NTHR, bool ROWSPLIT>
__global__ __launch_bounds__(NTHR) void conv1x1_bwd_fused_kernel(FusedArgs p) {
  constexpr int R = 64;
  constexpr int LDZ = MT * 32 + 1;         // odd pitch: the data gradient reads dz by ROW (32 lanes = 32 rows -> 32 banks), the weight
                                           // gradient by column (consecutive lanes = consecutive floats): both conflict-free
  constexpr int LDX = 32;                  // x tile [R][32] (read by column only)
  constexpr int LDW = 33;                  // W tile [MT*32][33]: read as b[k = co][n = ci] with lanes over ci, k uniform per half-wave
  constexpr int VZ = R * MT * 32 / 4 / NTHR;           // float4 slots per thread covering [R, MT*32]
  static_assert(R * MT * 32 / 4 % NTHR == 0, "chunk must divide over the block");
  extern __shared__ float smem[];
  float* dzs = smem;                       // [R][LDZ]
  float* xs = dzs + R * LDZ;               // [R][LDX]
  float* wt = xs + R * LDX;                // [MT*32][LDW]
  float* kab = wt + MT * 32 * LDW;         // [3][MT*32]  ka | kb | kc (in registers they cost 12 VGPRs per float4 slot: occupancy 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = p.Cout >> 2, aq = p.Cin >> 2;

  for (int i = tid; i < R * LDZ + R * LDX + MT * 32 * LDW; i += NTHR) smem[i] = 0.f;     // padding columns stay zero
  __syncthreads();
  for (int i = tid; i < p.Cout * p.Cin; i += NTHR) {
    const int co = i / p.Cin, ci = i - co * p.Cin;
    wt[co * LDW + ci] = p.w[i];
  }
  for (int i = tid; i < 3 * MT * 32; i += NTHR) {
    const int which = i / (MT * 32), c = i - which * MT * 32;
    kab[i] = c < p.Cout ? p.kabc[which * p.Cout + c] : 0.f;
  }

  // loop-invariant placement of this thread's slots: dz [R, Cout] as float4 along Cout, x [R, Cin] as float4 along Cin
  int zr[VZ], zc[VZ];
#pragma unroll
  for (int i = 0; i < VZ; ++i) {
    const int idx = tid + NTHR * i;
    zr[i] = idx / cq;
    zc[i] = (idx - zr[i] * cq) * 4;
    if (zr[i] >= R) { zr[i] = -1; zc[i] = 0; }
  }
  int xr_[2], xc_[2];                       // two float4 slots per thread cover [R, Cin <= 32]
  bool x_on[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + NTHR * i;
    xr_[i] = idx / aq;
    xc_[i] = (idx - xr_[i] * aq) * 4;
    x_on[i] = xr_[i] < R;
    if (!x_on[i]) { xr_[i] = 0; xc_[i] = 0; }
  }

  const int64_t nchunks = (p.rows + R - 1) / R;
  float4 rdu[VZ], rz[VZ], rx[2];
  float rcur[16];                           // data-gradient waves: the residual values of their 16 output rows
  const int kh_ = lane >> 5, cl_ = min(lane & 31, p.Cin - 1);

  auto fetch = [&](int64_t chunk) {          // unconditional loads on clamped rows (a predicated load de-pipelines: skinny_wgrad.hip)
    const int64_t r0 = chunk * R;
    const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
    const float* du_c = p.du + r0 * p.Cout;
    const float* z_c = p.z + r0 * p.Cout;
#pragma unroll
    for (int i = 0; i < VZ; ++i) {
      const int off = min(max(zr[i], 0), last) * p.Cout + zc[i];
      rdu[i] = *reinterpret_cast<const float4*>(du_c + off);
      rz[i] = *reinterpret_cast<const float4*>(z_c + off);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) rx[i] = *reinterpret_cast<const float4*>(p.x + (r0 + min(xr_[i], last)) * p.Cin + xc_[i]);
  };
  auto stage = [&](int64_t chunk) {
    const int64_t r0 = chunk * R;
    const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);

#pragma unroll
    for (int i = 0; i < VZ; ++i) {
      if (zr[i] >= 0) {
        const bool ok = zr[i] < left;
        float* dst = dzs + zr[i] * LDZ + zc[i];
        const float4 ka = *reinterpret_cast<const float4*>(kab + zc[i]);
        const float4 kb = *reinterpret_cast<const float4*>(kab + MT * 32 + zc[i]);
        const float4 kc = *reinterpret_cast<const float4*>(kab + 2 * MT * 32 + zc[i]);
        dst[0] = ok ? fmaf(ka.x, rdu[i].x, fmaf(kb.x, rz[i].x, kc.x)) : 0.f;
        dst[1] = ok ? fmaf(ka.y, rdu[i].y, fmaf(kb.y, rz[i].y, kc.y)) : 0.f;
        dst[2] = ok ? fmaf(ka.z, rdu[i].z, fmaf(kb.z, rz[i].z, kc.z)) : 0.f;
        dst[3] = ok ? fmaf(ka.w, rdu[i].w, fmaf(kb.w, rz[i].w, kc.w)) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (x_on[i]) {
        const bool ok = xr_[i] < left;
        *reinterpret_cast<float4*>(xs + xr_[i] * LDX + xc_[i]) =
            make_float4(ok ? rx[i].x : 0.f, ok ? rx[i].y : 0.f, ok ? rx[i].z : 0.f, ok ? rx[i].w : 0.f);
      }
  };

  const int kh = lane >> 5, cl = lane & 31;
  // weight-gradient waves: wave 2 owns the Cout tiles [0, MTA), wave 3 the tiles [MTA, MT), each over all 64 rows of a chunk:
  // dW tiles [32 co][32 ci] resident in accumulators for the whole launch, MTA of them per wave (splitting the ROWS instead
  // kept MT tiles live in every wave: 344 VGPRs at MT = 5, one wave per SIMD)
  constexpr int MTA = ROWSPLIT ? MT : (MT + 1) / 2;
  static_assert(!ROWSPLIT || 2 * MT * 32 * 32 <= R * LDZ, "the row-split reduction buffer reuses the dz tile");
  f32x16 wacc[MTA];
#pragma unroll
  for (int i = 0; i < MTA; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) wacc[i][r] = 0.f;
  const int t0 = (!ROWSPLIT && wave == 3) ? MTA : 0;       // first tile of this wave (weight-gradient role)

  int64_t chunk = blockIdx.x;
  if (chunk < nchunks) fetch(chunk);
  __syncthreads();                          // zero fill and W in place
  for (; chunk < nchunks; chunk += gridDim.x) {
    stage(chunk);
    __syncthreads();
    if (p.res && wave < 2) {
      // requested BEFORE the next chunk's prefetch: memory operations retire in order, so the epilogue's wait for these 16 values
      // leaves the prefetch in flight (issued after it, or in the epilogue itself, that wait would drain the prefetch too)
      const int64_t r0 = chunk * R;
      const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rcur[r] = p.res[(r0 + min(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_, last)) * p.Cin + cl_];
    }
    const int64_t nxt = chunk + gridDim.x;
    if (nxt < nchunks) fetch(nxt);          // in flight while this chunk is multiplied
    if (wave < 2) {
      // ---- data gradient of row tile `wave`: out[32 rows][32 ci] = dz[32 rows][Cout] . W[Cout][32 ci]
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* a_w = dzs + (wave * 32 + cl) * LDZ + kh;
      const float* b_w = wt + kh * LDW + cl;
#pragma unroll 8
      for (int ks = 0; ks < MT * 16; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[2 * ks], b_w[2 * ks * LDW], acc, 0, 0, 0);
      const int64_t rbase = chunk * R + wave * 32;
      if (cl < p.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = rbase + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (row < p.rows) {
            float v = acc[r];
            if (p.res) v += rcur[r];
            p.dx[row * p.Cin + cl] = v;
          }
        }
      }
    } else if (wave < 4) {
      if constexpr (ROWSPLIT) {
        // ---- weight gradient over rows [(wave - 2) * 32, +32) of the chunk, all Cout tiles: dW[co][ci] += dz[r][co] * x[r][ci], k = rows
        const int rb = (wave - 2) * 32;
        const float* dz_w = dzs + rb * LDZ + cl;
        const float* x_w = xs + rb * LDX + cl;
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
          const int r = 2 * ks + kh;
          const float bf = x_w[r * LDX];
#pragma unroll
          for (int i = 0; i < MT; ++i) wacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz_w[r * LDZ + i * 32], bf, wacc[i], 0, 0, 0);
        }
      } else {
        // ---- weight gradient of this wave's Cout tiles over the chunk's 64 rows: dW[co][ci] += dz[r][co] * x[r][ci], k = rows
        const float* dz_w = dzs + t0 * 32 + cl;
        const float* x_w = xs + cl;
#pragma unroll 4
        for (int ks = 0; ks < R / 2; ++ks) {
          const int r = 2 * ks + kh;
          const float bf = x_w[r * LDX];
#pragma unroll
          for (int i = 0; i < MTA; ++i)
            if (MT % 2 == 0 || i < MTA - 1 || wave == 2)          // the second wave has one tile less when MT is odd
              wacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz_w[r * LDZ + i * 32], bf, wacc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  if constexpr (ROWSPLIT) {
    // the two weight-gradient waves meet in LDS (the dz tile is free now), then one global atomic per weight and block
    float* red = dzs;                       // [2][MT*32][32]
    if (wave >= 2 && wave < 4) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          red[((wave - 2) * MT * 32 + m) * 32 + cl] = wacc[i][r];
        }
    }
    __syncthreads();
    for (int i = tid; i < p.Cout * p.Cin; i += NTHR) {
      const int co = i / p.Cin, ci = i - co * p.Cin;
      atomicAdd(p.dw + i, red[co * 32 + ci] + red[(MT * 32 + co) * 32 + ci]);
    }
  } else if (wave >= 2 && wave < 4) {
    // every weight-gradient tile has one owner per block: straight to the global accumulation
#pragma unroll
    for (int i = 0; i < MTA; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (t0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (t0 + i < MT && m < p.Cout && cl < p.Cin) atomicAdd(p.dw + m * p.Cin + cl, wacc[i][r]);
      }
  }
}

template <int MT, int NTHR, bool ROWSPLIT>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  constexpr int R = 64, LDZ = MT * 32 + 1;
  const size_t smem = ((size_t)R * LDZ + R * 32 + MT * 32 * 33 + 3 * MT * 32) * 4;
  const int64_t nchunks = (a.rows + R - 1) / R;
  const int blocks = (int)(nchunks < 512 ? nchunks : 512);
  auto k = conv1x1_bwd_fused_kernel<MT, NTHR, ROWSPLIT>;
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_conv1x1_bwd_fused: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3(blocks), dim3(NTHR), smem, st, a);
  return check_launch("mt_conv1x1_bwd_fused");
}

}  // namespace

extern "C" int mt_conv1x1_bwd_fused_supported(int Cout, int Cin) {
  if ((Cout & 3) || (Cin & 3) || Cin <= 0 || Cin > 32) return 0;
  const int mt_ = (Cout + 31) / 32;
  return Cout == mt_ * 32 ? (mt_ == 3) : (mt_ == 5 && Cout == 144);     // 96 -> <= 32 and 144 -> <= 32 channels
}

extern "C" int mt_conv1x1_bwd_fused(const float* du, const float* z, const float* kabc, const float* x, const float* w, const float* res,
                                    float* dx, float* dw, int64_t rows, int Cout, int Cin, void* stream) {
  if (!du || !z || !kabc || !x || !w || !dx || !dw) return fail(MT_ERR_ARG, "mt_conv1x1_bwd_fused: null pointer");
  if (!mt_conv1x1_bwd_fused_supported(Cout, Cin))
    return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_bwd_fused: no instance for %d -> %d channels", Cout, Cin);
  if (((uintptr_t)du | (uintptr_t)z | (uintptr_t)x | (uintptr_t)kabc) & 15) return fail(MT_ERR_ARG, "mt_conv1x1_bwd_fused: 16-byte alignment");
  FusedArgs a{du, z, kabc, x, w, res, dx, dw, rows, Cout, Cin};
  hipStream_t st = (hipStream_t)stream;
  return (Cout + 31) / 32 == 3 ? launch_fused<3, 256, true>(a, st) : launch_fused<5, 512, true>(a, st);
}
