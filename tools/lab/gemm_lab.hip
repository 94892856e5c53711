// Tuning lab for the LDS-DMA GEMM main loop (csrc/gemm_dma.hpp): times variants against the production mt_gemm on the shapes of a
// B = 32 training step and checks the results against it.  Build: tools/lab/build.sh ; run on the GPU box: tools/lab/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../include/mintime_hip.h"
// the library exports the same template instances: give the lab's copies their own symbol names, or the runtime resolves a
// launch by NAME to the library's (non-ablated) kernel
#define gemm_dma_kernel gemm_dma_kernel_lab
#define gemm_split_kernel gemm_split_kernel_lab
#include "../../mintime-multi-identity-size-invariant-timesformer-for-video-deepfake-detection_amd/csrc/gemm_split.hpp"

using namespace mt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int op; int M, N, K; int epi; int split; };

static float* dalloc(size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 9) % 2001 - 1000) * 0.001f * scale; }
  float* d; CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

template <typename F> static float time_ms(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 25; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Problem {
  Shape s; float *A, *B, *C, *Cref, *bias, *R, *C2; size_t c_elems; int64_t lda, ldb, ldc;
};

static GemmArgs make_args(const Problem& pr, int bm, int bn, int bk, int& gx, int& gy) {
  const Shape& s = pr.s;
  GemmArgs a; memset(&a, 0, sizeof(a));
  a.A = pr.A; a.B = pr.B; a.C = pr.C; a.M = s.M; a.N = s.N; a.K = s.K; a.lda = pr.lda; a.ldb = pr.ldb; a.ldc = pr.ldc;
  a.bias = pr.bias; a.R = pr.C; a.ldr = pr.ldc; a.hw = 1; a.stats_slots = 1; a.b_hw = 1;
  if (s.epi == EPI_GEGLU) { a.n_half = s.N / 2; a.C2 = pr.C2; a.ldc2 = s.N; }
  if (s.epi == EPI_GEGLU_BWD) { a.n_half = s.N; a.C2 = pr.C2; a.ldc2 = 2 * s.N; }
  const int m_tiles = (s.M + bm - 1) / bm, n_tiles = (s.N + bn - 1) / bn;
  gx = m_tiles * n_tiles; gy = 1;
  if (m_tiles >= 32 && n_tiles >= 2) {
    const int64_t panel = (int64_t)bn * s.K * 4;
    int gn = (int)((2 << 20) / (panel > 0 ? panel : 1));
    if (gn < 1) gn = 1;
    if (gn > n_tiles) gn = n_tiles;
    a.group_n = gn;
    gx = 8 * ((m_tiles + 7) / 8) * n_tiles;
  }
  int splits = s.split;
  if (s.op == MT_OP_TN && splits <= 0) {
    const int tiles = m_tiles * n_tiles;
    splits = (2048 + tiles - 1) / tiles;
    const int mx = s.K / 256 > 0 ? s.K / 256 : 1;
    if (splits > mx) splits = mx;
  }
  if (splits > 1 || s.op == MT_OP_TN) {
    int chunk = (s.K + splits - 1) / splits;
    chunk = (chunk + bk - 1) / bk * bk;
    a.k_chunk = chunk; gy = (s.K + chunk - 1) / chunk;
  }
  return a;
}

// max |x - fp64 reference| over a sample of outputs, relative to the largest reference magnitude (plain matmul epilogues only)
static void err64(const Problem& pr, const std::vector<float>& got, const std::vector<float>& base, double& e_got, double& e_base) {
  e_got = e_base = -1;
  const Shape& s = pr.s;
  if (s.epi != EPI_STORE && s.epi != EPI_ATOMIC) return;
  static std::vector<float> hA, hB, hBias; static const float* cached = nullptr;
  size_t a_el = (size_t)s.M * s.K, b_el = (size_t)s.N * s.K;
  if (cached != pr.A) { hA.resize(a_el); hB.resize(b_el); CK(hipMemcpy(hA.data(), pr.A, a_el * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), pr.B, b_el * 4, hipMemcpyDeviceToHost)); hBias.resize(s.N); CK(hipMemcpy(hBias.data(), pr.bias, (size_t)s.N * 4, hipMemcpyDeviceToHost)); cached = pr.A; }
  double mg = 0, mb = 0, mr = 0;
  uint64_t st = 12345;
  for (int it = 0; it < 4000; ++it) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int m = (int)((st >> 33) % s.M);
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int n = (int)((st >> 33) % s.N);
    double acc = 0;
    for (int k = 0; k < s.K; ++k) {
      const double av = s.op == MT_OP_TN ? hA[(size_t)k * pr.lda + m] : hA[(size_t)m * pr.lda + k];
      const double bv = s.op == MT_OP_NT ? hB[(size_t)n * pr.ldb + k] : hB[(size_t)k * pr.ldb + n];
      acc += av * bv;
    }
    acc += hBias[n];                 // both plain epilogues add the bias
    const size_t ci = (size_t)m * pr.ldc + n;
    mg = fmax(mg, fabs(got[ci] - acc)); mb = fmax(mb, fabs(base[ci] - acc)); mr = fmax(mr, fabs(acc));
  }
  e_got = mg / mr; e_base = mb / mr;
}

template <int WM, int WN, int TM, int TN, int AL, int BL, int EPI, int BK, int ST, int MINW, int MMA = MMA_F32>
static float run_variant(const Problem& pr, int reps) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  int gx, gy;
  GemmArgs a = make_args(pr, BM, BN, BK, gx, gy);
  if (pr.s.K % BK || (a.k_chunk && a.k_chunk % BK)) return -1.f;
  auto k = gemm_dma_kernel<WM, WN, TM, TN, AL, BL, EPI, BK, ST, MINW, PRO_NONE, MMA>;
  const size_t lds = (size_t)ST * (BM + BN) * BK * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto f = [&]() {
    if (EPI == EPI_ATOMIC) CK(hipMemsetAsync(pr.C, 0, pr.c_elems * 4, 0));
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(WM * WN * 64), lds, 0, a);
  };
  f(); CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  // compare with the reference result
  std::vector<float> h(pr.c_elems), r(pr.c_elems);
  CK(hipMemcpy(h.data(), pr.C, pr.c_elems * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), pr.Cref, pr.c_elems * 4, hipMemcpyDeviceToHost));
  double md = 0, mr = 0;
  for (size_t i = 0; i < pr.c_elems; ++i) { md = fmax(md, fabs((double)h[i] - r[i])); mr = fmax(mr, fabs((double)r[i])); }
  const float ms = time_ms(f, reps);
  const float memset_ms = EPI == EPI_ATOMIC ? time_ms([&]() { CK(hipMemsetAsync(pr.C, 0, pr.c_elems * 4, 0)); }, reps) : 0.f;
  double e_got, e_base;
  err64(pr, h, r, e_got, e_base);
  printf("    %dx%d bk%d st%d w%d %s : %8.1f us  %6.1f TF   relerr %.1e%s   err-vs-fp64 %.2e (fp32 pipe %.2e)\n", BM, BN, BK, ST, MINW,
         MMA == MMA_F32 ? "f32  " : (MMA == MMA_BF16X6 ? "bf16x6" : "bf16x3"), (ms - memset_ms) * 1e3,
         2.0 * pr.s.M * pr.s.N * pr.s.K / ((ms - memset_ms) * 1e-3) / 1e12, md / (mr > 0 ? mr : 1), md / (mr > 0 ? mr : 1) > 1e-4 ? "  <<<<<< MISMATCH" : "",
         e_got, e_base);
  fflush(stdout);
  return ms;
}

template <int WM, int WN, int TM, int TN, int AL, int BL, int EPI, int MINW, bool X6 = true, int PIPE = 2, bool BAL = false>
static float run_split(const Problem& pr, int reps) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  int gx, gy;
  GemmArgs a = make_args(pr, BM, BN, 16, gx, gy);
  if (pr.s.K % 16 || (a.k_chunk && a.k_chunk % 16)) return -1.f;
  auto k = gemm_split_kernel_lab<WM, WN, TM, TN, AL, BL, EPI, MINW, X6, PIPE, BAL>;
  const size_t lds = (size_t)2 * 3 * (BM + BN) * 32;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto f = [&]() {
    if (EPI == EPI_ATOMIC) CK(hipMemsetAsync(pr.C, 0, pr.c_elems * 4, 0));
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(WM * WN * 64), lds, 0, a);
  };
  f(); CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> h(pr.c_elems), r(pr.c_elems);
  CK(hipMemcpy(h.data(), pr.C, pr.c_elems * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), pr.Cref, pr.c_elems * 4, hipMemcpyDeviceToHost));
  double md = 0, mr = 0;
  for (size_t i = 0; i < pr.c_elems; ++i) { md = fmax(md, fabs((double)h[i] - r[i])); mr = fmax(mr, fabs((double)r[i])); }
  const float ms = time_ms(f, reps);
  const float memset_ms = EPI == EPI_ATOMIC ? time_ms([&]() { CK(hipMemsetAsync(pr.C, 0, pr.c_elems * 4, 0)); }, reps) : 0.f;
  double e_got, e_base;
  err64(pr, h, r, e_got, e_base);
  printf("    SPLIT p%d%s %dx%d %dw(%dx%d) w%d %s : %8.1f us  %6.1f TF   relerr %.1e%s   err-vs-fp64 %.2e (fp32 pipe %.2e)\n", PIPE, BAL ? " bal" : "", BM, BN, WM * WN, WM, WN, MINW,
         X6 ? "bf16x6" : "bf16x3", (ms - memset_ms) * 1e3,
         2.0 * pr.s.M * pr.s.N * pr.s.K / ((ms - memset_ms) * 1e-3) / 1e12, md / (mr > 0 ? mr : 1), md / (mr > 0 ? mr : 1) > 1e-4 ? "  <<<<<< MISMATCH" : "",
         e_got, e_base);
  fflush(stdout);
  return ms;
}
#define PIPE_ARG

static float run_baseline(Problem& pr, int reps) {
  const Shape& s = pr.s;
  mt_gemm_desc d; memset(&d, 0, sizeof(d));
  d.op = s.op; d.epilogue = s.epi; d.A = pr.A; d.B = pr.B; d.C = pr.Cref; d.M = s.M; d.N = s.N; d.K = s.K;
  d.lda = pr.lda; d.ldb = pr.ldb; d.ldc = pr.ldc; d.bias = pr.bias; d.R = pr.Cref; d.ldr = pr.ldc; d.split_k = s.split;
  if (s.epi == EPI_GEGLU) { d.n_half = s.N / 2; d.C2 = pr.C2; d.ldc2 = s.N; }
  if (s.epi == EPI_GEGLU_BWD) { d.n_half = s.N; d.C2 = pr.C2; d.ldc2 = 2 * s.N; }
  auto f = [&]() {
    if (s.epi == EPI_ATOMIC) CK(hipMemsetAsync(pr.Cref, 0, pr.c_elems * 4, 0));
    int rc = mt_gemm(&d, 0);
    if (rc) { printf("mt_gemm failed: %s\n", mt_last_error()); exit(1); }
  };
  const float ms = time_ms(f, reps);
  const float memset_ms = s.epi == EPI_ATOMIC ? time_ms([&]() { CK(hipMemsetAsync(pr.Cref, 0, pr.c_elems * 4, 0)); }, reps) : 0.f;
  f(); CK(hipDeviceSynchronize());
  printf("  %-34s baseline (mt_gemm)   : %8.1f us  %6.1f TF\n", s.name, (ms - memset_ms) * 1e3, 2.0 * s.M * s.N * s.K / ((ms - memset_ms) * 1e-3) / 1e12);
  fflush(stdout);
  return ms;
}

static Problem make_problem(const Shape& s) {
  Problem pr; pr.s = s;
  size_t a_el, b_el;
  if (s.op == MT_OP_NT) { pr.lda = s.K; pr.ldb = s.K; a_el = (size_t)s.M * s.K; b_el = (size_t)s.N * s.K; }
  else if (s.op == MT_OP_NN) { pr.lda = s.K; pr.ldb = s.N; a_el = (size_t)s.M * s.K; b_el = (size_t)s.K * s.N; }
  else { pr.lda = s.M; pr.ldb = s.N; a_el = (size_t)s.K * s.M; b_el = (size_t)s.K * s.N; }
  pr.A = dalloc(a_el, 1, 1.0f); pr.B = dalloc(b_el, 2, 0.05f);
  pr.ldc = s.N; pr.c_elems = (size_t)s.M * s.N;
  if (s.epi == EPI_GEGLU) { pr.ldc = s.N / 2; pr.c_elems = (size_t)s.M * s.N / 2; }
  if (s.epi == EPI_GEGLU_BWD) { pr.ldc = 2 * s.N; pr.c_elems = (size_t)s.M * 2 * s.N; }
  CK(hipMalloc(&pr.C, pr.c_elems * 4)); CK(hipMalloc(&pr.Cref, pr.c_elems * 4));
  CK(hipMemset(pr.C, 0, pr.c_elems * 4)); CK(hipMemset(pr.Cref, 0, pr.c_elems * 4));
  pr.bias = dalloc(s.N, 3, 0.1f);
  pr.C2 = nullptr;
  if (s.epi == EPI_GEGLU) CK(hipMalloc(&pr.C2, (size_t)s.M * s.N * 4));
  if (s.epi == EPI_GEGLU_BWD) pr.C2 = dalloc((size_t)s.M * 2 * s.N, 4, 1.0f);
  pr.R = nullptr;
  return pr;
}

static void free_problem(Problem& pr) { hipFree(pr.A); hipFree(pr.B); hipFree(pr.C); hipFree(pr.Cref); hipFree(pr.bias); if (pr.C2) hipFree(pr.C2); }

#define KC LAYOUT_KCONTIG
#define KM LAYOUT_KMAJOR

template <int AL, int BL, int EPI> static void sweep(Problem& pr, int reps) {
#ifdef MT_LAB_SPLIT_ONE
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, true, 2, true>(pr, reps);
  return;
#endif
#ifdef MT_LAB_SPLIT_BAL
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, true, 2, false>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, true, 2, true>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 1, true, 2, true>(pr, reps);
  run_split<4, 2, 1, 2, AL, BL, EPI, 2, true, 2, true>(pr, reps);
  run_split<4, 2, 1, 2, AL, BL, EPI, 3, true, 2, true>(pr, reps);
  run_split<4, 2, 1, 2, AL, BL, EPI, 3, true, 2, false>(pr, reps);
  run_split<4, 2, 1, 2, AL, BL, EPI, 4, true, 2, true>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) {
    run_split<2, 4, 2, 1, AL, BL, EPI, 4, true, 2, true>(pr, reps);
    run_split<2, 4, 2, 1, AL, BL, EPI, 3, true, 2, true>(pr, reps);
    run_split<4, 2, 2, 1, AL, BL, EPI, 3, true, 2, true>(pr, reps);     // 256 x 64
  }
  return;
#endif
#ifdef MT_LAB_SPLIT_AB
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, true, 1>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, true, 2>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 1, true, 2>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) run_split<4, 2, 2, 2, AL, BL, EPI, 2, true, 1>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) run_split<4, 2, 2, 2, AL, BL, EPI, 2, true, 2>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) run_split<4, 2, 2, 2, AL, BL, EPI, 1, true, 2>(pr, reps);
  return;
#endif
#ifdef MT_LAB_SPLIT
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2, MMA_BF16X6>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 2>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 3>(pr, reps);
  run_split<2, 2, 2, 2, AL, BL, EPI, 2, false>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) {
    run_split<2, 2, 2, 1, AL, BL, EPI, 3>(pr, reps);
    run_split<2, 2, 2, 4, AL, BL, EPI, 1>(pr, reps);
    run_split<4, 2, 2, 2, AL, BL, EPI, 2>(pr, reps);
    run_split<4, 2, 1, 2, AL, BL, EPI, 4>(pr, reps);
    run_split<2, 4, 2, 1, AL, BL, EPI, 4>(pr, reps);
  }
  return;
#endif
#ifdef MT_LAB_BF16
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2, MMA_BF16X6>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 3, 2, MMA_BF16X6>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 2, 2, MMA_BF16X6>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2, MMA_BF16X3>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) {
    run_variant<2, 2, 2, 1, AL, BL, EPI, 16, 3, 3, MMA_BF16X6>(pr, reps);
    run_variant<2, 2, 1, 1, AL, BL, EPI, 32, 3, 4, MMA_BF16X6>(pr, reps);
    run_variant<4, 2, 1, 2, AL, BL, EPI, 16, 2, 4, MMA_BF16X6>(pr, reps);
    run_variant<4, 2, 2, 2, AL, BL, EPI, 16, 2, 2, MMA_BF16X6>(pr, reps);      // 256 x 128 tile, 8 waves of 64 x 64
    run_variant<2, 2, 2, 4, AL, BL, EPI, 16, 2, 1, MMA_BF16X6>(pr, reps);      // 128 x 256 tile, 4 waves of 64 x 128
  }
  return;
#endif
#ifdef MT_LAB_8WAVE
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 2, 3>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 2, 4>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) {
    run_variant<4, 2, 1, 2, AL, BL, EPI, 16, 2, 6>(pr, reps);
    run_variant<4, 2, 1, 2, AL, BL, EPI, 16, 2, 8>(pr, reps);
    run_variant<4, 2, 1, 2, AL, BL, EPI, 16, 3, 6>(pr, reps);
    run_variant<4, 2, 1, 2, AL, BL, EPI, 32, 2, 6>(pr, reps);
    run_variant<2, 4, 2, 1, AL, BL, EPI, 16, 2, 6>(pr, reps);
    run_variant<4, 2, 2, 2, AL, BL, EPI, 16, 2, 4>(pr, reps);      // 256 x 128 tile, 8 waves of 64 x 64
    run_variant<4, 2, 2, 2, AL, BL, EPI, 32, 2, 4>(pr, reps);
  } else {
    run_variant<4, 2, 2, 2, AL, BL, EPI, 16, 2, 4>(pr, reps);
  }
  return;
#endif
#if MT_DMA_ABLATE
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 3, 3>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 2, 3>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) run_variant<2, 2, 1, 1, AL, BL, EPI, 32, 3, 4>(pr, reps);
  return;
#endif
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 2, 3>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 3, 3>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 16, 4, 2>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 3, 1>(pr, reps);
  run_variant<2, 2, 2, 2, AL, BL, EPI, 32, 2, 2>(pr, reps);
  if constexpr (EPI != EPI_GEGLU) {
    run_variant<2, 2, 1, 1, AL, BL, EPI, 16, 3, 4>(pr, reps);
    run_variant<2, 2, 1, 1, AL, BL, EPI, 32, 3, 4>(pr, reps);
    run_variant<2, 2, 2, 1, AL, BL, EPI, 16, 3, 3>(pr, reps);
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const char* only = argc > 2 ? argv[2] : nullptr;
  const int M = 12576;
  Shape shapes[] = {
      {"NT 4096^3 store", MT_OP_NT, 4096, 4096, 4096, EPI_STORE, 1},
      {"NT qkv 12576x1536x512", MT_OP_NT, M, 1536, 512, EPI_STORE, 1},
      {"NT ff1+geglu 12576x4096x512", MT_OP_NT, M, 4096, 512, EPI_GEGLU, 1},
      {"NT outproj+res 12576x512x512", MT_OP_NT, M, 512, 512, EPI_BIAS_RES, 1},
      {"NT ff2 splitk5 12576x512x2048", MT_OP_NT, M, 512, 2048, EPI_ATOMIC, 5},
      {"NN geglu_bwd 12576x2048x512", MT_OP_NN, M, 2048, 512, EPI_GEGLU_BWD, 1},
      {"NN ff1 dgrad sk4 12576x512x4096", MT_OP_NN, M, 512, 4096, EPI_ATOMIC, 4},
      {"NN ff1 dgrad sk1 12576x512x4096 store", MT_OP_NN, M, 512, 4096, EPI_STORE, 1},
      {"NT ff1 dgrad-as-NT 12576x512x4096 store", MT_OP_NT, M, 512, 4096, EPI_STORE, 1},
      {"NN qkv dgrad sk1 12576x512x1536 store", MT_OP_NN, M, 512, 1536, EPI_STORE, 1},
      {"NT qkv dgrad-as-NT 12576x512x1536 store", MT_OP_NT, M, 512, 1536, EPI_STORE, 1},
      {"NN qkv dgrad sk3 12576x512x1536", MT_OP_NN, M, 512, 1536, EPI_ATOMIC, 3},
      {"NN outproj dgrad 12576x512x512", MT_OP_NN, M, 512, 512, EPI_STORE, 1},
      {"TN ff2 wgrad 512x2048x12576", MT_OP_TN, 512, 2048, M, EPI_ATOMIC, 0},
      {"TN ff1 wgrad 4096x512x12576", MT_OP_TN, 4096, 512, M, EPI_ATOMIC, 0},
      {"TN qkv wgrad 1536x512x12576", MT_OP_TN, 1536, 512, M, EPI_ATOMIC, 0},
      {"TN out wgrad 512x512x12576", MT_OP_TN, 512, 512, M, EPI_ATOMIC, 0},
      {"NT ef expand 50176x480x80", MT_OP_NT, 50176, 480, 80, EPI_STORE, 1},
      {"NT ef head 12544x1280x320", MT_OP_NT, 12544, 1280, 320, EPI_STORE, 1},
      {"NT ef14 project 50176x80x480", MT_OP_NT, 50176, 80, 480, EPI_STORE, 1},
      {"NT ef14 project 50176x112x672", MT_OP_NT, 50176, 112, 672, EPI_STORE, 1},
      {"NN ef14 exp dgrad 50176x80x480", MT_OP_NN, 50176, 80, 480, EPI_STORE, 1},
      {"NN ef14 prj dgrad 50176x480x80", MT_OP_NN, 50176, 480, 80, EPI_STORE, 1},
      {"TN ef14 exp wgrad 480x80x50176 auto", MT_OP_TN, 480, 80, 50176, EPI_ATOMIC, 0},
      {"TN ef14 exp wgrad 480x80x50176 s32", MT_OP_TN, 480, 80, 50176, EPI_ATOMIC, 32},
      {"TN ef14 exp wgrad 480x80x50176 s64", MT_OP_TN, 480, 80, 50176, EPI_ATOMIC, 64},
      {"NT ef7 expand 12544x1152x192", MT_OP_NT, 12544, 1152, 192, EPI_STORE, 1},
      {"NT ef7 project 12544x192x1152", MT_OP_NT, 12544, 192, 1152, EPI_STORE, 1},
      {"NN ef7 exp dgrad 12544x192x1152", MT_OP_NN, 12544, 192, 1152, EPI_STORE, 1},
      {"NN ef7 prj dgrad 12544x1152x192", MT_OP_NN, 12544, 1152, 192, EPI_STORE, 1},
      {"TN ef7 wgrad 1152x192x12544 auto", MT_OP_TN, 1152, 192, 12544, EPI_ATOMIC, 0},
      {"TN ef7 wgrad 1152x192x12544 s8", MT_OP_TN, 1152, 192, 12544, EPI_ATOMIC, 8},
      {"TN ef7 wgrad 1152x192x12544 s16", MT_OP_TN, 1152, 192, 12544, EPI_ATOMIC, 16},
      {"TN ef7 wgrad 1152x192x12544 s28", MT_OP_TN, 1152, 192, 12544, EPI_ATOMIC, 28},
  };
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    Problem pr = make_problem(s);
    run_baseline(pr, reps);
    if (s.op == MT_OP_NT && s.epi == EPI_STORE) sweep<KC, KC, EPI_STORE>(pr, reps);
    if (s.op == MT_OP_NT && s.epi == EPI_GEGLU) sweep<KC, KC, EPI_GEGLU>(pr, reps);
    if (s.op == MT_OP_NT && s.epi == EPI_BIAS_RES) sweep<KC, KC, EPI_BIAS_RES>(pr, reps);
    if (s.op == MT_OP_NT && s.epi == EPI_ATOMIC) sweep<KC, KC, EPI_ATOMIC>(pr, reps);
    if (s.op == MT_OP_NN && s.epi == EPI_GEGLU_BWD) sweep<KC, KM, EPI_GEGLU_BWD>(pr, reps);
    if (s.op == MT_OP_NN && s.epi == EPI_ATOMIC) sweep<KC, KM, EPI_ATOMIC>(pr, reps);
    if (s.op == MT_OP_NN && s.epi == EPI_STORE) sweep<KC, KM, EPI_STORE>(pr, reps);
    if (s.op == MT_OP_TN) sweep<KM, KM, EPI_ATOMIC>(pr, reps);
    free_problem(pr);
  }
  return 0;
}
