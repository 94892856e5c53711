// 1x1 convolutions with few channels and very many rows, forward and data gradient (EfficientNet-B0 stages 1-4: 0.2-3.2 M rows,
// 16..240 channels on either side; reference efficientnet_pytorch/model.py:93-118).  These are pure streaming problems -- a few
// hundred MB in, a few hundred MB out, MFMA time 5-10x below the HBM time -- and the tiled GEMM leaves them at 60-75 % of what a
// streaming kernel reaches, because every 128-row tile pays its own un-overlapped load -> compute -> store sequence.
//
//   out[r, co] = sum_ci a[r, ci] * W[co, ci]        a = x                                      (expand conv, forward)
//                                                     | swish(sc*x + sh) * gate[r / hw]          (project conv, forward)
//                                                     | ka*du + kb*z + kc                        (any conv, data gradient: W passed transposed)
//   optional: + R[r, co] (residual branch of the data gradient), BatchNorm statistics of `out` (fp64 slots, one atomic per
//   column per block)
//
// Same skeleton as skinny_wgrad.hip: a persistent block streams 64- or 128-row chunks through registers into LDS with the next
// chunk in flight, W stays in LDS for the whole launch, each wavefront owns row tile(s) of the chunk and multiplies them against
// every 32-column tile of W (v_mfma_f32_32x32x2_f32, M = rows), results go straight from the accumulators to global memory.
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { A_PLAIN = 0, A_GATE = 1, A_BNBWD = 2 };

struct ConvArgs {
  const float* x; const float* x2;                       // [rows, Cin]; x2 = second source (BNBWD: z)
  const float* w;                                        // w_t == 0: [Cout, ldw] (row co holds the Cin taps of output column co);
  int ldw, w_t;                                          // w_t == 1: [Cin, ldw] (the forward weight, used transposed by the data gradient)
  const float* sc; const float* sh; const float* gate;   // GATE: [Cin], [Cin], [rows / hw, Cin];  BNBWD: ka, kb, kc [Cin]
  const float* res;                                      // optional [rows, Cout] added to the output
  float* out;                                            // [rows, Cout]
  double* stats; int slots;                              // optional BatchNorm statistics of `out`: [slots][2][Cout]
  int64_t rows; int Cin, Cout, hw;
};

__device__ __forceinline__ float swish_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }   /* v_rcp_f32 (1 ulp), see effnet_fwd.hip sigmoidf_ */

// KT = 32-wide k tiles (Cin <= 32 KT), NT = 32-wide output column tiles, R = rows per chunk (one 32-row tile per wavefront at
// R = 128, two wavefronts per row tile at R = 64)
template <int KT, int NT, int R, int AMODE>
__global__ __launch_bounds__(256) void conv1x1_rows_kernel(ConvArgs p) {
  constexpr int LDK = KT * 32 + 1;                        // odd pitch: the 32 rows of a fragment read hit 32 different banks
  constexpr int VA = (R * KT * 32 / 4 + 255) / 256;       // float4 slots per thread covering [R, Cin]
  constexpr int WPT = R == 128 ? 1 : 2;                   // wavefronts sharing one row tile
  constexpr int NTW = (NT + WPT - 1) / WPT;               // column tiles per wavefront
  constexpr int NTP = NTW * WPT;                          // W is padded with zero tiles up to this, so the MFMA loop needs no tile test
  extern __shared__ float smem[];
  float* at = smem;                                       // [R][LDK]
  float* wt = at + R * LDK;                               // [NTP*32][LDK]
  float* cst = wt + NTP * 32 * LDK;                        // [3][KT*32] per-channel prologue constants
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int aq = p.Cin >> 2;

  for (int i = tid; i < R * LDK + NTP * 32 * LDK; i += 256) smem[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < p.Cout * p.Cin; i += 256) {
    const int co = i / p.Cin, ci = i - co * p.Cin;
    wt[co * LDK + ci] = p.w_t ? p.w[(int64_t)ci * p.ldw + co] : p.w[(int64_t)co * p.ldw + ci];
  }
  if constexpr (AMODE != A_PLAIN)
    for (int i = tid; i < KT * 32; i += 256) {
      const bool ok = i < p.Cin;
      cst[i] = ok ? p.sc[i] : 0.f;
      cst[KT * 32 + i] = ok ? p.sh[i] : 0.f;
      cst[2 * KT * 32 + i] = (ok && AMODE == A_BNBWD) ? p.gate[i] : 0.f;
    }

  int ar[VA], ac[VA];
#pragma unroll
  for (int i = 0; i < VA; ++i) {
    const int idx = tid + 256 * i;
    ar[i] = idx / aq;
    ac[i] = (idx - ar[i] * aq) * 4;
    if (ar[i] >= R) { ar[i] = -1; ac[i] = 0; }
  }
  const int64_t nchunks = (p.rows + R - 1) / R;
  float4 rx[VA], rx2[AMODE == A_BNBWD ? VA : 1];

  auto fetch = [&](int64_t chunk) {            // unconditional loads on clamped rows (see skinny_wgrad.hip)
    const int64_t r0 = chunk * R;
    const int last = (int)((p.rows - r0) < R ? (p.rows - r0) : R) - 1;
    const float* x_c = p.x + r0 * p.Cin;
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      const int off = min(max(ar[i], 0), last) * p.Cin + ac[i];
      rx[i] = *reinterpret_cast<const float4*>(x_c + off);
      if constexpr (AMODE == A_BNBWD) rx2[i] = *reinterpret_cast<const float4*>(p.x2 + r0 * p.Cin + off);
    }
  };
  auto stage = [&](int64_t chunk) {
    const int64_t r0 = chunk * R;
    const int left = (int)((p.rows - r0) < R ? (p.rows - r0) : R);
    int64_t img0 = 0;
    int rem0 = 0;
    if constexpr (AMODE == A_GATE) { img0 = div_rows(r0, p.hw); rem0 = (int)(r0 - img0 * p.hw); }
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      if (ar[i] < 0) continue;
      const bool ok = ar[i] < left;
      float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
      if constexpr (AMODE == A_GATE) {
        if (ok) {
          int t = rem0 + ar[i];
          int64_t img = img0;
          while (t >= p.hw) { t -= p.hw; ++img; }
          const float4 g = *reinterpret_cast<const float4*>(p.gate + img * p.Cin + ac[i]);
          const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = swish_f(fmaf(cst[ac[i] + e], v[e], cst[KT * 32 + ac[i] + e])) * gg[e];
        }
      } else if constexpr (AMODE == A_BNBWD) {
        const float z[4] = {rx2[i].x, rx2[i].y, rx2[i].z, rx2[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(cst[ac[i] + e], v[e], fmaf(cst[KT * 32 + ac[i] + e], z[e], cst[2 * KT * 32 + ac[i] + e]));
      }
      float* dst = at + ar[i] * LDK + ac[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = ok ? v[e] : 0.f;
    }
  };

  const int kh = lane >> 5, cl = lane & 31;
  const int rt = WPT == 1 ? wave : (wave >> 1);           // this wavefront's row tile
  const int ct0 = WPT == 1 ? 0 : (wave & 1) * NTW;        // its first column tile
  float s1[NTW], s2[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  int64_t chunk = blockIdx.x;
  if (chunk < nchunks) fetch(chunk);
  __syncthreads();                                       // W and the constants are in place
  for (; chunk < nchunks; chunk += gridDim.x) {
    stage(chunk);
    __syncthreads();
    const int64_t nxt = chunk + gridDim.x;
    if (nxt < nchunks) fetch(nxt);                       // in flight while this chunk is multiplied and stored
    f32x16 acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* a_w = at + (rt * 32 + cl) * LDK + kh;
#pragma unroll 4
    for (int ks = 0; ks < KT * 16; ++ks) {
      const float af = a_w[2 * ks];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const float bf = wt[((ct0 + j) * 32 + cl) * LDK + 2 * ks + kh];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[j], 0, 0, 0);
      }
    }
    // accumulators -> global: lane holds column (ct*32 + cl) of rows rt*32 + (r&3) + 8(r>>2) + 4 kh
    const int64_t r0 = chunk * R + rt * 32;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int co = (ct0 + j) * 32 + cl;
      if (co < p.Cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (row < p.rows) {
            float v = acc[j][r];
            if (p.res) v += p.res[row * p.Cout + co];
            if (p.out) p.out[row * p.Cout + co] = v;        // out == NULL: BatchNorm statistics of the product only
            s1[j] += v; s2[j] = fmaf(v, v, s2[j]);
          }
        }
      }
    }
    __syncthreads();
  }

  if (p.stats) {
    // per column: the two halves of a wavefront, then the wavefronts holding the same column tile, then one fp64 atomic
    float* red = smem;                                    // [4 waves][NTW][2][32]
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float a = s1[j] + __shfl_xor(s1[j], 32), b = s2[j] + __shfl_xor(s2[j], 32);
      if (kh == 0) { red[((wave * NTW + j) * 2 + 0) * 32 + cl] = a; red[((wave * NTW + j) * 2 + 1) * 32 + cl] = b; }
    }
    __syncthreads();
    for (int i = tid; i < NT * 32 * 2; i += 256) {
      const int which = i / (NT * 32), co = i - which * NT * 32;
      if (co >= p.Cout) continue;
      const int ct = co >> 5, c = co & 31;
      float v = 0.f;
      for (int w = 0; w < 4; ++w) {
        const int w_ct0 = WPT == 1 ? 0 : (w & 1) * NTW;
        const int j = ct - w_ct0;
        if (j >= 0 && j < NTW) v += red[((w * NTW + j) * 2 + which) * 32 + c];
      }
      stat_add(p.stats + ((int64_t)(blockIdx.x % stat_slots(p.slots)) * 2 + which) * p.Cout + co, stat_limb(p.slots, p.Cout), v);
    }
  }
}

template <int KT, int NT, int R>
int launch(const ConvArgs& a, int amode, hipStream_t st) {
  constexpr int LDK = KT * 32 + 1;
  constexpr int WPT = R == 128 ? 1 : 2, NTP = (NT + WPT - 1) / WPT * WPT;
  const size_t smem = ((size_t)R * LDK + (size_t)NTP * 32 * LDK + 3 * KT * 32) * 4;
  const int64_t nchunks = (a.rows + R - 1) / R;
  const int blocks = (int)(nchunks < 512 ? nchunks : 512);
  auto go = [&](auto k) {
    (void)ensure_dynamic_lds((const void*)k, smem);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), smem, st, a);
  };
  if (amode == A_GATE) go(conv1x1_rows_kernel<KT, NT, R, A_GATE>);
  else if (amode == A_BNBWD) go(conv1x1_rows_kernel<KT, NT, R, A_BNBWD>);
  else go(conv1x1_rows_kernel<KT, NT, R, A_PLAIN>);
  return check_launch("mt_conv1x1_rows");
}

}  // namespace

// instances: (Cin tiles, Cout tiles, rows per chunk).  Only the shapes with one k tile (or 96 -> <= 32 channels) beat the tiled GEMM
// (measured on the 256-crop batch, us, streaming / GEMM): data gradients 16->32 165/216, 24->96 101/155, 96->16 495/665,
// 24->144 227/249; forward 24->144 163/186.  Wider k (144, 240 channels in) loses to the GEMM (the register->LDS transpose of a
// 64 x 144 tile costs more than it hides) and is not instantiated; the gated forward is a tie and stays on the GEMM.
#define MT_ROWS_INSTANCES(X) X(1, 1, 128) X(1, 3, 128) X(1, 5, 128) X(3, 1, 128)

static bool rows_instance(int Cin, int Cout) {
  if ((Cin & 3) || Cin <= 0 || Cout <= 0) return false;
  const int kt = (Cin + 31) / 32, nt = (Cout + 31) / 32;
#define MT_CASE(K_, N_, R_) if (kt == K_ && nt == N_) return true;
  MT_ROWS_INSTANCES(MT_CASE)
#undef MT_CASE
  return false;
}

// does mt_conv1x1_rows have an instance for this channel pair at all?  (mt_conv1x1_rows_supported: is it the better choice)
extern "C" int mt_conv1x1_rows_instance(int Cin, int Cout) { return rows_instance(Cin, Cout) ? 1 : 0; }

// is the streaming kernel the better choice for this conv?  (amode as in mt_conv1x1_rows)
extern "C" int mt_conv1x1_rows_supported(int Cin, int Cout, int amode) {
  if (!rows_instance(Cin, Cout)) return 0;
  const int kt = (Cin + 31) / 32, nt = (Cout + 31) / 32;
  if (amode == A_BNBWD) return 1;
  if (amode == A_PLAIN) return kt == 1 && nt == 5;
  return 0;
}

extern "C" int mt_conv1x1_rows(const float* x, const float* x2, const float* w, int ldw, int w_transposed, const float* c0,
                               const float* c1, const float* c2, int hw, int amode, const float* res, float* out, double* stats,
                               int slots, int64_t rows, int Cin, int Cout, void* stream) {
  if (!x || !w || (!out && !stats)) return fail(MT_ERR_ARG, "mt_conv1x1_rows: null pointer");
  if (!rows_instance(Cin, Cout)) return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_rows: no instance for %d -> %d channels", Cin, Cout);
  if (amode == A_GATE && (!c0 || !c1 || !c2 || hw <= 0)) return fail(MT_ERR_ARG, "mt_conv1x1_rows: gate mode needs scale, shift, gate, hw");
  if (amode == A_BNBWD && (!c0 || !c1 || !c2 || !x2)) return fail(MT_ERR_ARG, "mt_conv1x1_rows: BN-backward mode needs ka, kb, kc and z");
  if (amode < 0 || amode > 2) return fail(MT_ERR_ARG, "mt_conv1x1_rows: bad mode");
  if (((uintptr_t)x | (uintptr_t)x2) & 15) return fail(MT_ERR_ARG, "mt_conv1x1_rows: 16-byte alignment");
  ConvArgs a{x, x2, w, ldw, w_transposed ? 1 : 0, c0, c1, c2, res, out, stats, slots != 0 ? slots : 1, rows, Cin, Cout, hw > 0 ? hw : 1};
  hipStream_t st = (hipStream_t)stream;
  const int kt = (Cin + 31) / 32, nt = (Cout + 31) / 32;
#define MT_CASE(K_, N_, R_) if (kt == K_ && nt == N_) return launch<K_, N_, R_>(a, amode, st);
  MT_ROWS_INSTANCES(MT_CASE)
#undef MT_CASE
  return fail(MT_ERR_UNSUPPORTED, "mt_conv1x1_rows: no instance");
}
