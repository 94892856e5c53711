// Weight gradient of the MBConv PROJECT convolution in the late stages (EfficientNet-B0 blocks 5-15 at 14^2 / 7^2: 12.5-50 K rows per
// 256-crop batch, 80-320 output channels against 240-1152 expanded channels; reference efficientnet_pytorch/model.py:104-106 driven
// backwards by train.py:371).
//
//   dW[co, c] += sum_r dz[r, co] * a[r, c]
//   dz[r, co]  = ka[co]*du[r, co] + kb[co]*z[r, co] + kc[co]                       BatchNorm (bn2) backward folded into the load
//   a[r, c]    = swish(sc[c]*x[r, c] + sh[c]) * gate[r / hw, c]                    BN1 + swish + squeeze-excite gate on load
//
// The generic TN GEMM re-applied both operand transforms at every fragment read and re-read the operands three times: 160-220 us
// per launch at 21 % of the fp32 MFMA rate (11 launches, 1.8 ms per step).  skinny_wgrad.hip keeps the WHOLE result in one block's
// accumulators, which stops at 64 x 256 weights.  Here the result is cut into 128-column slabs of C: a block owns one slab x all Co
// rows (its four MFMA wavefronts one 32-column tile each, Co / 32 accumulator tiles per wavefront) and a contiguous range of the
// rows; partial sums meet in global fp32 atomics.  Same role split as skinny_bwd.hip: waves 4-7 load 32-row chunks, apply both
// transforms ONCE per element and stage them into double-buffered LDS tiles; waves 0-3 only multiply (operands of the next k-step
// requested before the current one is multiplied); one barrier per chunk.  Both operands are read from LDS "by column" (k = rows),
// so no transposes.  Blocks of the same row range sit next to each other on ONE XCD (the narrow operand comes from HBM once per
// range and from that L2 for the other slabs).
// Algorithmic bytes = rows * (C + 2 * Co) * 4.
#include "common.hpp"
#include <stdint.h>
#include <type_traits>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WideArgs {
  const float* du; const float* z; const float* kabc;      // narrow: [rows, Co] x2, [3, Co]
  const float* x; const float* sc; const float* sh;         // [rows, C], [C], [C]
  const float* gate;                                         // [rows / hw, C]
  float* dw;                                                 // [Co, C], accumulated with atomics
  int64_t rows; int Co, C, hw;
  int nslab, nsplit; int64_t chunks_per_split;
};

__device__ __forceinline__ float swish_w(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }   /* v_rcp_f32 (1 ulp), see effnet_fwd.hip sigmoidf_ */

constexpr int R = 32;                      // rows per chunk
constexpr int SLAB = 128;                  // columns of C per block
constexpr int LDA = SLAB + 32;             // pitch % 64 == 32: the two k-rows of a fragment read hit disjoint banks
constexpr int ldz_for(int tiles) { return tiles * 32 + ((tiles & 1) ? 0 : 32); }

template <int MT> struct WideSmem {
  static constexpr int LDZ = ldz_for(MT);
  static constexpr int DZ = R * LDZ, AS = R * LDA;
  static constexpr int FLOATS = 2 * DZ + 2 * AS + 3 * MT * 32 + 2 * SLAB;
};

// MT = 32-row tiles of Co (3: 80, 4: 112, 6: 192, 10: 320)
template <int MT>
__global__ __launch_bounds__(512) void wgrad_wide_kernel(WideArgs p) {
  using S = WideSmem<MT>;
  constexpr int LDZ = S::LDZ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dzs = smem;                       // [2][R][LDZ]
  float* as = dzs + 2 * S::DZ;             // [2][R][LDA]
  float* kab = as + 2 * S::AS;             // [3][MT*32]  ka | kb | kc
  float* ssh = kab + 3 * MT * 32;          // [2][SLAB]   sc | sh of this slab
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = lane >> 5, cl = lane & 31;

  // (row range, slab): workgroups are dealt round-robin to the 8 XCDs, so XCD x gets the ranges x, x + 8, ... with all their slabs
  const int lin = blockIdx.x, xcd = lin & 7, idx = lin >> 3;
  const int split = xcd + 8 * (idx / p.nslab), slab = idx % p.nslab;
  const int c0 = slab * SLAB;
  const int cw = min(SLAB, p.C - c0);      // valid columns of this slab (a multiple of 4)
  const int64_t nchunks = (p.rows + R - 1) / R;
  const int64_t c_lo = (int64_t)split * p.chunks_per_split;
  const int64_t c_hi = c_lo + p.chunks_per_split < nchunks ? c_lo + p.chunks_per_split : nchunks;
  const int n_it = c_hi > c_lo ? (int)(c_hi - c_lo) : 0;
  if (n_it == 0) return;                   // uniform over the block: before any barrier

  for (int i = tid; i < S::FLOATS; i += 512) smem[i] = 0.f;          // padding columns stay zero for the whole launch
  __syncthreads();
  for (int i = tid; i < MT * 32; i += 512)
    if (i < p.Co) { kab[i] = p.kabc[i]; kab[MT * 32 + i] = p.kabc[p.Co + i]; kab[2 * MT * 32 + i] = p.kabc[2 * p.Co + i]; }
  for (int i = tid; i < cw; i += 512) { ssh[i] = p.sc[c0 + i]; ssh[SLAB + i] = p.sh[c0 + i]; }
  __syncthreads();

  if (wave >= 4) {
    // ================================================= PRODUCERS =================================================
    const int ptid = tid - 256;
    const int cq = p.Co >> 2;
    int zr[MT], zc[MT];                     // MT float4 slots per thread cover [R, MT*32 >= Co]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int id = ptid + 256 * i;
      zr[i] = id / cq;
      zc[i] = (id - zr[i] * cq) * 4;
      if (zr[i] >= R) { zr[i] = -1; zc[i] = 0; }
    }
    const int ac = (ptid & 31) * 4, ar0 = ptid >> 5;        // a slots: rows ar0 + 8 i, i < 4; one column quad per thread
    const bool a_on = ac < cw;
    const int acl = a_on ? ac : 0;
    const float4 s4 = *reinterpret_cast<const float4*>(ssh + acl), h4 = *reinterpret_cast<const float4*>(ssh + SLAB + acl);
    constexpr int NSET = MT <= 6 ? 2 : 1;   // register sets in flight (two: chunks i+1 and i+2 are loading while chunk i is multiplied;
                                            // 320 output channels leave room for one)
    float4 rdu[NSET][MT], rz[NSET][MT], rx[NSET][4], rg[NSET][4];

    // unconditional loads on clamped rows and columns (a predicated load de-pipelines: skinny_wgrad.hip); the gate rows ride along
    // (a load inside the staging step would put its latency on the producers' critical path once per chunk)
    auto fetch = [&](auto set_c, int64_t chunk) {
      constexpr int SET = decltype(set_c)::value;
      const int64_t r0 = chunk * R;
      const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int64_t off = (r0 + min(max(zr[i], 0), last)) * p.Co + zc[i];
        rdu[SET][i] = *reinterpret_cast<const float4*>(p.du + off);
        rz[SET][i] = *reinterpret_cast<const float4*>(p.z + off);
      }
      const float* xc = p.x + c0 + acl;
      const float* gc = p.gate + c0 + acl;
      const int64_t img0 = div_rows(r0, p.hw);                        // uniform: one division per chunk
      const int rem0 = (int)(r0 - img0 * p.hw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(ar0 + 8 * i, last);
        rx[SET][i] = *reinterpret_cast<const float4*>(xc + (r0 + row) * p.C);
        const int64_t img = img0 + (rem0 + row >= p.hw ? 1 : 0);        // hw >= R: a chunk touches at most two images
        rg[SET][i] = *reinterpret_cast<const float4*>(gc + img * p.C);
      }
    };
    auto stage = [&](auto set_c, int64_t chunk, int buf) {
      constexpr int SET = decltype(set_c)::value;
      const int64_t r0 = chunk * R;
      const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);
      float* dzb = dzs + buf * S::DZ;
      float* ab = as + buf * S::AS;
#pragma unroll
      for (int i = 0; i < MT; ++i)
        if (zr[i] >= 0) {
          const bool ok = zr[i] < left;
          const float4 ka = *reinterpret_cast<const float4*>(kab + zc[i]);
          const float4 kb = *reinterpret_cast<const float4*>(kab + MT * 32 + zc[i]);
          const float4 kc = *reinterpret_cast<const float4*>(kab + 2 * MT * 32 + zc[i]);
          float4 v;
          v.x = ok ? fmaf(ka.x, rdu[SET][i].x, fmaf(kb.x, rz[SET][i].x, kc.x)) : 0.f;
          v.y = ok ? fmaf(ka.y, rdu[SET][i].y, fmaf(kb.y, rz[SET][i].y, kc.y)) : 0.f;
          v.z = ok ? fmaf(ka.z, rdu[SET][i].z, fmaf(kb.z, rz[SET][i].z, kc.z)) : 0.f;
          v.w = ok ? fmaf(ka.w, rdu[SET][i].w, fmaf(kb.w, rz[SET][i].w, kc.w)) : 0.f;
          *reinterpret_cast<float4*>(dzb + zr[i] * LDZ + zc[i]) = v;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = ar0 + 8 * i;
        const bool ok = a_on && row < left;
        const float4 v = rx[SET][i], g = rg[SET][i];
        float4 o;
        o.x = ok ? swish_w(fmaf(s4.x, v.x, h4.x)) * g.x : 0.f;
        o.y = ok ? swish_w(fmaf(s4.y, v.y, h4.y)) * g.y : 0.f;
        o.z = ok ? swish_w(fmaf(s4.z, v.z, h4.z)) * g.z : 0.f;
        o.w = ok ? swish_w(fmaf(s4.w, v.w, h4.w)) * g.w : 0.f;
        *reinterpret_cast<float4*>(ab + row * LDA + ac) = o;
      }
    };
    // barrier k publishes chunk k - 1; the consumers reach it after finishing chunk k - 2, whose buffer chunk k refills.
    // Memory operations retire in order: staging one set waits for ITS loads only and leaves the other set's in flight.
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, NSET - 1>;
    if constexpr (NSET == 2) {
      fetch(I0{}, c_lo);
      if (n_it > 1) fetch(I1{}, c_lo + 1);
      for (int j = 0; j < n_it; j += 2) {
        stage(I0{}, c_lo + j, 0);
        if (j + 2 < n_it) fetch(I0{}, c_lo + j + 2);
        lds_barrier();
        if (j + 1 < n_it) {
          stage(I1{}, c_lo + j + 1, 1);
          if (j + 3 < n_it) fetch(I1{}, c_lo + j + 3);
          lds_barrier();
        }
      }
    } else {
      fetch(I0{}, c_lo);
      for (int j = 0; j < n_it; ++j) {
        stage(I0{}, c_lo + j, j & 1);
        if (j + 1 < n_it) fetch(I0{}, c_lo + j + 1);
        lds_barrier();
      }
    }
  } else {
    // ================================================= CONSUMERS =================================================
    f32x16 acc[MT];                         // dW tiles [32 co][32 c] of this wavefront's column tile, live for the whole launch
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bool w_on = wave * 32 < cw;       // column tiles past C only keep the barriers company
    for (int j = 0; j < n_it; ++j) {
      lds_barrier();                      // buffer j & 1 staged
      if (!w_on) continue;
      const float* dz_w = dzs + (j & 1) * S::DZ + kh * LDZ + cl;
      const float* a_w = as + (j & 1) * S::AS + kh * LDA + wave * 32 + cl;
      float d0[MT], d1[MT], b0, b1;         // operands of the next k-step requested before the current one is multiplied
      b0 = a_w[0];
#pragma unroll
      for (int i = 0; i < MT; ++i) d0[i] = dz_w[i * 32];
#pragma unroll 1
      for (int ks = 0; ks < R / 2; ks += 2) {
        b1 = a_w[2 * (ks + 1) * LDA];
#pragma unroll
        for (int i = 0; i < MT; ++i) d1[i] = dz_w[2 * (ks + 1) * LDZ + i * 32];
        __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of the multiplies (the scheduler sinks them otherwise)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0[i], b0, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = ks + 2 < R / 2 ? ks + 2 : 0;
        b0 = a_w[2 * k2 * LDA];
#pragma unroll
        for (int i = 0; i < MT; ++i) d0[i] = dz_w[2 * k2 * LDZ + i * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[i], b1, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // every wavefront owns its tiles: straight to the global accumulation
    const int c = wave * 32 + cl;
    if (c < cw) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (co < p.Co) atomicAdd(p.dw + (int64_t)co * p.C + c0 + c, acc[i][r]);
        }
    }
  }
}

template <int MT>
int launch_wide(WideArgs a, hipStream_t st) {
  const size_t smem = (size_t)WideSmem<MT>::FLOATS * 4;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t nchunks = (a.rows + R - 1) / R;
  a.nslab = (a.C + SLAB - 1) / SLAB;
  int per_xcd = cus / (8 * a.nslab);       // row ranges per XCD: one persistent block per CU, ranges in multiples of 8
  if (per_xcd < 1) per_xcd = 1;
  a.nsplit = 8 * per_xcd;
  if ((int64_t)a.nsplit > nchunks) a.nsplit = (int)((nchunks + 7) / 8 * 8);
  a.chunks_per_split = (nchunks + a.nsplit - 1) / a.nsplit;
  auto k = wgrad_wide_kernel<MT>;
  hipError_t e = ensure_dynamic_lds((const void*)k, smem);
  if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_conv1x1_wgrad_wide: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
  hipLaunchKernelGGL(k, dim3(a.nsplit * a.nslab), dim3(512), smem, st, a);
  return check_launch("mt_conv1x1_wgrad_wide");
}

}  // namespace

extern "C" int mt_conv1x1_wgrad_wide_supported(int Cout, int Cin) {
  if ((Cout & 3) || (Cin & 3) || Cout < 65 || Cout > 320 || Cin < 2 * SLAB - 32 || Cin > 16 * SLAB) return 0;
  const int mt_ = (Cout + 31) / 32;
  return mt_ == 3 || mt_ == 4 || mt_ == 6 || mt_ == 10;
}

extern "C" int mt_conv1x1_wgrad_wide(const float* du, const float* z, const float* kabc, const float* x, const float* sc, const float* sh,
                                     const float* gate, int hw, float* dw, int64_t rows, int Cout, int Cin, void* stream) {
  if (!du || !z || !kabc || !x || !sc || !sh || !gate || !dw) return fail(MT_ERR_ARG, "mt_conv1x1_wgrad_wide: null pointer");
  if (hw < R || rows <= 0) return fail(MT_ERR_ARG, "mt_conv1x1_wgrad_wide: rows must be positive and hw >= %d", R);
  if (!mt_conv1x1_wgrad_wide_supported(Cout, Cin))
    return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_wgrad_wide: no instance for %d x %d weights", Cout, Cin);
  if (((uintptr_t)du | (uintptr_t)z | (uintptr_t)x | (uintptr_t)kabc | (uintptr_t)gate | (uintptr_t)sc | (uintptr_t)sh) & 15)
    return fail(MT_ERR_ARG, "mt_conv1x1_wgrad_wide: 16-byte alignment");
  WideArgs a{du, z, kabc, x, sc, sh, gate, dw, rows, Cout, Cin, hw, 0, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  switch ((Cout + 31) / 32) {
    case 3: return launch_wide<3>(a, st);
    case 4: return launch_wide<4>(a, st);
    case 6: return launch_wide<6>(a, st);
    default: return launch_wide<10>(a, st);
  }
}
