// Tuning lab for the plane-operand GEMM main loop (csrc/gemm_planes.hpp): checks it against the production mt_gemm (in-kernel split)
// and times variants on the TimeSformer's shapes at B = 32.  Build: tools/lab/build_planes.sh ; run on the GPU box: tools/lab/planes_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../include/mintime_hip.h"
#define gemm_planes_kernel gemm_planes_kernel_lab
#define gemm_split_kernel gemm_split_kernel_lab
#include "../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc/gemm_planes.hpp"

using namespace mt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define CKM(x) do { int r_ = (x); if (r_ != 0) { printf("mt error %d (%s) at %s:%d\n", r_, mt_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static float* dalloc(size_t n, unsigned seed, float scale) {
  if (getenv("LAB_ZERO")) scale = 0.f;
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) % 2001 - 1000) * 0.001f * scale * (1.0f + (s & 255) * 1e-4f); }
  float* d; CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

template <typename F> static float time_us(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 25; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

// ---- 1. semantics probe of ds_read_b64_tr_b16 (what gemm_planes.hpp assumes)
__global__ void tr_probe(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lds + threadIdx.x * 4));
  *reinterpret_cast<s16x4_t*>(out + threadIdx.x * 4) = v;
}

static void probe_tr() {
  std::vector<short> h(256);
  for (int i = 0; i < 256; ++i) h[i] = (short)i;
  short *din, *dout; CK(hipMalloc(&din, 512)); CK(hipMalloc(&dout, 512));
  CK(hipMemcpy(din, h.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, din, dout);
  std::vector<short> o(256);
  CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
  // expectation: group g (16 lanes) holds the 4x16 row-major matrix M[j][c] = element g*64 + j*16 + c  (lane i supplied the 4 elements
  // from i*4); lane i receives M[0..3][i]
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int expect = (l >> 4) * 64 + j * 16 + (l & 15);
      if (o[l * 4 + j] != expect) ++bad;
    }
  printf("[tr probe] mismatches vs the assumed semantics: %d\n", bad);
  if (bad) {
    for (int l = 0; l < 20; ++l) printf("  lane %2d: %d %d %d %d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  CK(hipFree(din)); CK(hipFree(dout));
}

// fp32 row-major [R][C] (leading dimension ld) -> blocked planes [3][Rp/32][Cp/16][32][16], zero padded
__global__ __launch_bounds__(256) void to_blk_planes(const float* __restrict__ src, int64_t ld, int R, int C, __bf16* __restrict__ planes, int64_t pstride, int cb16) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
  if (r < R) {
    for (int e = 0; e < 4; ++e) { if (c + e < C) lo[e] = src[(int64_t)r * ld + c + e]; if (c + 4 + e < C) hi[e] = src[(int64_t)r * ld + c + 4 + e]; }
  }
  bf16x8_t x0, x1, x2;
  split_bf16<true>(lo, hi, x0, x1, x2);
  const int64_t o = ((int64_t)rb * cb16 + cb) * 512 + lane * 8;
  *reinterpret_cast<bf16x8_t*>(planes + o) = x0;
  *reinterpret_cast<bf16x8_t*>(planes + pstride + o) = x1;
  *reinterpret_cast<bf16x8_t*>(planes + 2 * pstride + o) = x2;
}

struct BlkPlanes { void* p; int64_t pstride; int cb16; };
static BlkPlanes make_blk(const float* src, int64_t ld, int R, int C) {
  const int Rp = (R + 31) / 32 * 32, Cp = (C + 15) / 16 * 16;
  BlkPlanes b; b.cb16 = Cp / 16; b.pstride = (int64_t)Rp * Cp;
  CK(hipMalloc(&b.p, (size_t)b.pstride * 6));
  hipLaunchKernelGGL(to_blk_planes, dim3((b.cb16 + 3) / 4, Rp / 32), dim3(256), 0, 0, src, ld, R, C, (__bf16*)b.p, b.pstride, b.cb16);
  CK(hipDeviceSynchronize());
  return b;
}

struct Shape { const char* name; int op; int M, N, K; int epi; };

struct Problem {
  Shape s; float *A, *B, *C, *Cref, *bias, *R, *C2; BlkPlanes Ap, Bp; size_t a_el, b_el, c_el; int64_t lda, ldb, ldc;
};

static Problem make_problem(const Shape& s) {
  Problem pr; memset(&pr, 0, sizeof(pr)); pr.s = s;
  // NT: A [M][K], B [N][K];  TN: A [K][M], B [K][N]
  pr.a_el = (size_t)s.M * s.K; pr.b_el = (size_t)s.N * s.K;
  pr.lda = s.op == MT_OP_TN ? s.M : s.K; pr.ldb = s.op == MT_OP_NT ? s.K : s.N;
  int out_n = s.N;
  if (s.epi == MT_EPI_GEGLU) out_n = s.N / 2;
  if (s.epi == MT_EPI_GEGLU_BWD) out_n = 2 * s.N;
  pr.ldc = out_n; pr.c_el = (size_t)s.M * out_n;
  pr.A = dalloc(pr.a_el, 1, 1.0f); pr.B = dalloc(pr.b_el, 2, 0.05f);
  pr.bias = dalloc((size_t)(s.epi == MT_EPI_GEGLU_BWD ? 2 * s.N : s.N), 3, 0.1f);
  CK(hipMalloc(&pr.C, pr.c_el * 4)); CK(hipMalloc(&pr.Cref, pr.c_el * 4));
  CK(hipMemset(pr.C, 0, pr.c_el * 4)); CK(hipMemset(pr.Cref, 0, pr.c_el * 4));
  pr.R = dalloc(pr.c_el, 4, 1.0f);
  if (s.epi == MT_EPI_GEGLU) { CK(hipMalloc(&pr.C2, (size_t)s.M * s.N * 4)); }
  if (s.epi == MT_EPI_GEGLU_BWD) pr.C2 = dalloc((size_t)s.M * 2 * s.N, 5, 1.0f);
  // operand tensors as stored: NT: A [M][K], B [N][K];  NN: A [M][K], B [K][N];  TN: A [K][M], B [K][N]
  pr.Ap = s.op == MT_OP_TN ? make_blk(pr.A, pr.lda, s.K, s.M) : make_blk(pr.A, pr.lda, s.M, s.K);
  pr.Bp = s.op == MT_OP_NT ? make_blk(pr.B, pr.ldb, s.N, s.K) : make_blk(pr.B, pr.ldb, s.K, s.N);
  return pr;
}

static void free_problem(Problem& pr) {
  hipFree(pr.A); hipFree(pr.B); hipFree(pr.C); hipFree(pr.Cref); hipFree(pr.bias); hipFree(pr.R); if (pr.C2) hipFree(pr.C2); hipFree(pr.Ap.p); hipFree(pr.Bp.p);
}

static mt_gemm_desc ref_desc(const Problem& pr, float* C) {
  const Shape& s = pr.s;
  mt_gemm_desc d; memset(&d, 0, sizeof(d));
  d.op = s.op; d.M = s.M; d.N = s.N; d.K = s.K; d.A = pr.A; d.B = pr.B; d.C = C; d.lda = pr.lda; d.ldb = pr.ldb; d.ldc = pr.ldc;
  d.epilogue = s.epi; d.bias = (s.epi == MT_EPI_GEGLU_BWD || s.op == MT_OP_TN) ? nullptr : pr.bias;
  if (s.epi == MT_EPI_BIAS_RES) { d.R = pr.R; d.ldr = pr.ldc; }
  if (s.epi == MT_EPI_GEGLU) { d.n_half = s.N / 2; d.C2 = pr.C2; d.ldc2 = s.N; }
  if (s.epi == MT_EPI_GEGLU_BWD) { d.n_half = s.N; d.C2 = pr.C2; d.ldc2 = 2 * s.N; }
  return d;
}

static GemmArgs make_args(const Problem& pr, int bm, int bn, int& gx, int& gy, int wgrad_blocks) {
  const Shape& s = pr.s;
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.A = pr.A; a.B = pr.B; a.C = pr.C; a.M = s.M; a.N = s.N; a.K = s.K; a.ldc = pr.ldc;
  a.a_planes = pr.Ap.p; a.a_pstride = pr.Ap.pstride; a.lda = pr.Ap.cb16; a.b_planes = pr.Bp.p; a.b_pstride = pr.Bp.pstride; a.ldb = pr.Bp.cb16;
  a.bias = (s.epi == MT_EPI_GEGLU_BWD || s.op == MT_OP_TN) ? nullptr : pr.bias; a.R = pr.R; a.ldr = pr.ldc; a.hw = 1; a.stats_slots = 1; a.b_hw = 1; a.e_hw = 1;
  if (s.epi == EPI_GEGLU) { a.n_half = s.N / 2; a.C2 = pr.C2; a.ldc2 = s.N; }
  if (s.epi == EPI_GEGLU_BWD) { a.n_half = s.N; a.C2 = pr.C2; a.ldc2 = 2 * s.N; }
  const int m_tiles = (s.M + bm - 1) / bm, n_tiles = (s.N + bn - 1) / bn;
  gx = m_tiles * n_tiles; gy = 1;
  if (s.op == MT_OP_TN) {
    int splits = (wgrad_blocks + gx - 1) / gx;
    const int mx = s.K / 256 > 0 ? s.K / 256 : 1;
    if (splits > mx) splits = mx;
    splits = (splits + 4) / 8 * 8; if (splits < 8) splits = 8;
    int chunk = (s.K + splits - 1) / splits; chunk = (chunk + 15) / 16 * 16;
    a.k_chunk = chunk; a.xcd_k = 1; gy = splits;
  } else if (m_tiles >= 32 && n_tiles >= 2) {
    const int64_t panel = (int64_t)bn * s.K * 6;     // three bf16 planes
    int gn = (int)((2 << 20) / (panel > 0 ? panel : 1));
    if (gn < 1) gn = 1;
    if (gn > n_tiles) gn = n_tiles;
    a.group_n = gn;
    gx = 8 * ((m_tiles + 7) / 8) * n_tiles;
  }
  return a;
}

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM, int EPI, int ST, int MINW, int BAL, bool CPL = false>
static float run_variant(const Problem& pr, int reps, const char* tag, bool check) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  int gx, gy;
  GemmArgs a = make_args(pr, BM, BN, gx, gy, 640);
  static void* cpl_buf = nullptr;
  if (CPL) {
    const int c_cols = EPI == EPI_GEGLU ? a.n_half : 2 * a.n_half;
    const int64_t el = (int64_t)((a.M + 31) / 32 * 32) * c_cols;
    if (!cpl_buf) CK(hipMalloc(&cpl_buf, (size_t)el * 6));
    a.c_planes = cpl_buf; a.c_pstride = el; a.ldcp = c_cols / 16; a.C = nullptr;
    static float* cs = nullptr;
    if (!cs) { CK(hipMalloc(&cs, (size_t)c_cols * 4)); CK(hipMemset(cs, 0, (size_t)c_cols * 4)); }
    a.col_sum = cs;
  }
  auto k = gemm_planes_kernel<WM, WN, TM, TN, AKM, BKM, EPI, ST, MINW, BAL, CPL>;
  const size_t lds = (size_t)ST * 3 * (BM + BN) * 32;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto f = [&]() {
    if (EPI == EPI_ATOMIC) CK(hipMemsetAsync(pr.C, 0, pr.c_el * 4, 0));
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(WM * WN * 64), lds, 0, a);
  };
  f(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  double maxd = 0, maxr = 0, sumd = 0; size_t nbad = 0;
  if (check) {
    std::vector<float> got(pr.c_el), ref(pr.c_el);
    CK(hipMemcpy(got.data(), pr.C, pr.c_el * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data(), pr.Cref, pr.c_el * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < pr.c_el; ++i) {
      const double d = (double)got[i] - ref[i];
      if (fabs(d) > maxd) maxd = fabs(d);
      sumd += d;
      if (fabs(ref[i]) > maxr) maxr = fabs(ref[i]);
      if (memcmp(&got[i], &ref[i], 4)) ++nbad;
    }
  }
  const float us = time_us(f, reps);
  const double tf = 2.0 * pr.s.M * pr.s.N * pr.s.K / (us * 1e-6) / 1e12;
  printf("  %-40s %8.1f us  %6.1f TF-eq  (%4.1f%%)  grid %dx%d lds %zu", tag, us, tf, tf / 416.7 * 100, gx, gy, lds);
  if (check) printf("  | max|d| %.2e (rel %.1e) mean d %.2e, %zu of %zu differ", maxd, maxr > 0 ? maxd / maxr : 0, sumd / pr.c_el, nbad, pr.c_el);
  printf("\n");
  fflush(stdout);
  return us;
}

template <int EPI, int OP>
static void run_shape(const Shape& s, int reps) {
  Problem pr = make_problem(s);
  mt_gemm_desc d = ref_desc(pr, pr.Cref);
  auto fref = [&]() {
    if (EPI == EPI_ATOMIC) CK(hipMemsetAsync(pr.Cref, 0, pr.c_el * 4, 0));
    CKM(mt_gemm(&d, nullptr));
  };
  const float us = time_us(fref, reps);
  const double tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
  printf("%s  op %d  M %d N %d K %d epi %d\n  %-40s %8.1f us  %6.1f TF-eq  (%4.1f%%)\n", s.name, s.op, s.M, s.N, s.K, s.epi,
         "production mt_gemm (in-kernel split)", us, tf, tf / 416.7 * 100);
  constexpr bool AKM = OP == MT_OP_TN, BKM = OP != MT_OP_NT;
#ifdef LAB_QUICK
  if constexpr (EPI == EPI_GEGLU_BWD) run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 2, BAL_PAIR, true>(pr, reps, "128x128 2 stages PAIR + planes out", false);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 2, BAL_PHASE>(pr, reps, "128x128 2 stages PHASE", true);
  if constexpr (OP != MT_OP_TN) run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 2, BAL_PAIR>(pr, reps, "128x128 2 stages PAIR", true);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 3, BAL_NONE>(pr, reps, "128x128 2 stages NONE minw3", true);
#else
  if constexpr (OP != MT_OP_TN) run_variant<2, 2, 2, 2, AKM, BKM, EPI, 3, 2, BAL_PAIR>(pr, reps, "128x128 3 stages PAIR", true);
  if constexpr (OP != MT_OP_TN) run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 2, BAL_PAIR>(pr, reps, "128x128 2 stages PAIR", true);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 3, 2, BAL_PHASE>(pr, reps, "128x128 3 stages PHASE minw2", true);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 2, BAL_PHASE>(pr, reps, "128x128 2 stages PHASE minw2", true);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 3, BAL_PHASE>(pr, reps, "128x128 2 stages PHASE minw3", true);
  run_variant<2, 2, 2, 2, AKM, BKM, EPI, 2, 3, BAL_NONE>(pr, reps, "128x128 2 stages NONE minw3", true);
  if constexpr (!AKM && EPI != EPI_GEGLU && EPI != EPI_GEGLU_BWD) {
    run_variant<2, 2, 4, 2, AKM, BKM, EPI, 2, 2, BAL_PHASE>(pr, reps, "256x128 (4 waves of 128x64) 2 st PHASE", true);
    run_variant<2, 2, 4, 2, AKM, BKM, EPI, 3, 2, BAL_PHASE>(pr, reps, "256x128 (4 waves of 128x64) 3 st PHASE", true);
  }
#endif
  free_problem(pr);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  const int M = 32 * 393;
  int idx = 0;
#define RUN(EPI, OP, ...) do { if (only < 0 || only == idx) run_shape<EPI, OP>(Shape __VA_ARGS__, reps); ++idx; } while (0)
  RUN(EPI_STORE, MT_OP_NT, {"4096^3", MT_OP_NT, 4096, 4096, 4096, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"QKV", MT_OP_NT, M, 1536, 512, MT_EPI_STORE});
  RUN(EPI_BIAS_RES, MT_OP_NT, {"out-proj", MT_OP_NT, M, 512, 512, MT_EPI_BIAS_RES});
  RUN(EPI_GEGLU, MT_OP_NT, {"FF1+GEGLU", MT_OP_NT, M, 4096, 512, MT_EPI_GEGLU});
  RUN(EPI_BIAS_RES, MT_OP_NT, {"FF2", MT_OP_NT, M, 512, 2048, MT_EPI_BIAS_RES});
  RUN(EPI_STORE, MT_OP_NN, {"FF1 dgrad (NN)", MT_OP_NN, M, 512, 4096, MT_EPI_STORE});
  RUN(EPI_GEGLU_BWD, MT_OP_NN, {"FF2 dgrad+GEGLU' (NN)", MT_OP_NN, M, 2048, 512, MT_EPI_GEGLU_BWD});
  RUN(EPI_STORE, MT_OP_NN, {"QKV dgrad (NN)", MT_OP_NN, M, 512, 1536, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NN, {"out dgrad (NN)", MT_OP_NN, M, 512, 512, MT_EPI_STORE});
  RUN(EPI_ATOMIC, MT_OP_TN, {"FF1 wgrad", MT_OP_TN, 4096, 512, M, MT_EPI_ATOMIC});
  RUN(EPI_ATOMIC, MT_OP_TN, {"FF2 wgrad", MT_OP_TN, 512, 2048, M, MT_EPI_ATOMIC});
  RUN(EPI_ATOMIC, MT_OP_TN, {"QKV wgrad", MT_OP_TN, 1536, 512, M, MT_EPI_ATOMIC});
  RUN(EPI_ATOMIC, MT_OP_TN, {"out wgrad", MT_OP_TN, 512, 512, M, MT_EPI_ATOMIC});
  RUN(EPI_STORE, MT_OP_NT, {"4096x4096x512", MT_OP_NT, 4096, 4096, 512, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"8192x8192x512", MT_OP_NT, 8192, 8192, 512, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"8192x8192x2048", MT_OP_NT, 8192, 8192, 2048, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"16384x512x512", MT_OP_NT, 16384, 512, 512, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"16384x1024x512", MT_OP_NT, 16384, 1024, 512, MT_EPI_STORE});
  RUN(EPI_STORE, MT_OP_NT, {"16384x512x2048", MT_OP_NT, 16384, 512, 2048, MT_EPI_STORE});
  return 0;
}
