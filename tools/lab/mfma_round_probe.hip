// How does v_mfma_f32_32x32x16_bf16 round  C + sum_k a_k b_k ?  Probe with exactly representable products far below ulp(C).
// Build: hipcc --offload-arch=gfx950 -O2 mfma_round_probe.hip -o mfma_round_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// every lane: a[e] = av (k slots 0..nk-1 of its half, rest 0), b[e] = bv; C = c0  -> D[0]
__global__ void probe(float av, float bv, int nk, float c0, float* out) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(e < nk ? av : 0.f); b[e] = (__bf16)(e < nk ? bv : 0.f); }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void probe_f32(float av, float bv, float c0, float* out) {
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

// general form: lane-half 0 supplies products pa[e]*pb[e], e = 0..7; lane-half 1 supplies zero
__global__ void probe_v(const float* pa, const float* pb, float c0, float* out) {
  bf16x8 a, b;
  const bool lo = threadIdx.x < 32;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(lo ? pa[e] : 0.f); b[e] = (__bf16)(lo ? pb[e] : 0.f); }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

static void run_v(const char* what, const double (&prod)[8], float c0) {
  // each product = m * 2^e with m exactly representable in bf16: factor as (m) * (2^e)
  float ha[8], hb[8];
  double exact = c0;
  for (int e = 0; e < 8; ++e) {
    int ex; double m = frexp(prod[e], &ex);      // prod = m * 2^ex, 0.5 <= |m| < 1
    ha[e] = (float)m; hb[e] = prod[e] == 0 ? 0.f : ldexpf(1.f, ex);
    exact += prod[e];
  }
  float *da, *db, *d; hipMalloc(&da, 32); hipMalloc(&db, 32); hipMalloc(&d, 4);
  hipMemcpy(da, ha, 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_v, dim3(1), dim3(64), 0, 0, da, db, c0, d);
  float r; hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
  const float rn = (float)exact;
  const double u = nextafterf(fabsf(rn), INFINITY) - fabsf(rn);
  printf("%-86s -> %a  (exact %.10a, RN %a, %+.2f ulp from exact)\n", what, r, exact, rn, (r - exact) / u);
  hipFree(da); hipFree(db); hipFree(d);
}

int main() {
  const double U = ldexp(1.0, -23);      // ulp(1.0)
  printf("--- per-product granularity / rounding against C = +-1 (ulp = 2^-23) ---\n");
  for (int sgn = 1; sgn >= -1; sgn -= 2)
    for (double c0 : {1.0, -1.0})
      for (double f : {0.75, 0.5, 0.375, 0.25, 0.1875, 0.125, 0.09375, 0.0625}) {
        char w[128]; snprintf(w, sizeof w, "C=%+.0f, one product of %+g ulp", c0, sgn * f);
        double pr[8] = {sgn * f * U, 0, 0, 0, 0, 0, 0, 0};
        run_v(w, pr, (float)c0);
      }
  printf("--- sums of in-range pieces: final rounding ---\n");
  for (int sgn = 1; sgn >= -1; sgn -= 2)
    for (double c0 : {1.0, -1.0})
      for (int n : {2, 3, 4, 5, 6, 7}) {
        char w[128]; snprintf(w, sizeof w, "C=%+.0f, %d products of %+g ulp (sum %+g ulp)", c0, n, sgn * 0.125, sgn * 0.125 * n);
        double pr[8] = {0}; for (int e = 0; e < n; ++e) pr[e] = sgn * 0.125 * U;
        run_v(w, pr, (float)c0);
      }
  printf("--- 8 equal products of f granules (granule = ulp/8) against C = +-1: floor / toward-zero / nearest per addend? ---\n");
  for (double c0 : {1.0, -1.0})
    for (double f : {1.25, 1.75, -1.25, -1.75, 0.75, -0.75, 2.5, -2.5}) {
      char w[128]; snprintf(w, sizeof w, "C=%+.0f, 8 products of %+g granules (exact sum %+g ulp)", c0, f, f);
      double pr[8]; for (int e = 0; e < 8; ++e) pr[e] = f * 0.125 * U;
      run_v(w, pr, (float)c0);
    }
  printf("--- the same with C = 0 and one product of 1.0 as the alignment reference ---\n");
  for (double f : {1.25, 1.75, -1.25, -1.75, 0.75, -0.75}) {
    char w[128]; snprintf(w, sizeof w, "C=0, products {1, 7 x %+g granules} (exact sum %+g ulp)", f, f * 7 / 8);
    double pr[8]; pr[0] = 1.0; for (int e = 1; e < 8; ++e) pr[e] = f * 0.125 * U;
    run_v(w, pr, 0.0f);
  }
  for (double f : {1.25, 1.75, -1.25, -1.75, 0.75, -0.75}) {
    char w[128]; snprintf(w, sizeof w, "C=0, products {-1, 7 x %+g granules} (exact sum %+g ulp)", f, f * 7 / 8);
    double pr[8]; pr[0] = -1.0; for (int e = 1; e < 8; ++e) pr[e] = f * 0.125 * U;
    run_v(w, pr, 0.0f);
  }
  printf("--- alignment reference: C = 1 next to one product of 2^10 ---\n");
  for (double f : {64.0, 128.0, 256.0, 512.0, 1024.0}) {
    char w[128]; snprintf(w, sizeof w, "C=1, products {2^10, %g ulp(1)}: ulp(2^10) = 1024 ulp(1)", f);
    double pr[8] = {1024.0, f * U, 0, 0, 0, 0, 0, 0};
    run_v(w, pr, 1.0f);
  }
  printf("--- C = 0, products only: {1, f ulp} ---\n");
  for (double f : {1.0, 0.5, 0.25, 0.125, 0.0625, 0.75, 1.5}) {
    char w[128]; snprintf(w, sizeof w, "C=0, products {1, %g ulp(1)}", f);
    double pr[8] = {1.0, f * U, 0, 0, 0, 0, 0, 0};
    run_v(w, pr, 0.0f);
  }
  printf("--- original cases ---\n");
  float* d; hipMalloc(&d, 4);
  struct Case { const char* what; float a, b; int nk; float c; };
  // with nk slots per half-wave, the contraction has 2*nk non-zero products (both lane halves contribute)
  Case cases[] = {
      {"C=1, 2 products of -2^-31 (sum -2^-30)", -ldexpf(1, -16), ldexpf(1, -15), 1, 1.0f},
      {"C=1, 2 products of +2^-31", ldexpf(1, -16), ldexpf(1, -15), 1, 1.0f},
      {"C=-1, 2 products of +2^-31", ldexpf(1, -16), ldexpf(1, -15), 1, -1.0f},
      {"C=-1, 2 products of -2^-31", -ldexpf(1, -16), ldexpf(1, -15), 1, -1.0f},
      {"C=1, 16 products of +2^-26 (sum 2^-22 = 2 ulp)", ldexpf(1, -13), ldexpf(1, -13), 8, 1.0f},
      {"C=1, 16 products of +2^-27 (sum 2^-23 = 1 ulp)", ldexpf(1, -14), ldexpf(1, -13), 8, 1.0f},
      {"C=1, 16 products of +2^-28 (sum 2^-24 = 1/2 ulp, tie)", ldexpf(1, -14), ldexpf(1, -14), 8, 1.0f},
      {"C=1, 16 products of +1.5*2^-28 (sum 0.75 ulp)", 1.5f * ldexpf(1, -14), ldexpf(1, -14), 8, 1.0f},
      {"C=1, 16 products of -1.5*2^-28 (sum -0.75 ulp(1) = -1.5 ulp below 1)", -1.5f * ldexpf(1, -14), ldexpf(1, -14), 8, 1.0f},
      {"C=1, 16 products of +2^-30 (sum 1/8 ulp)", ldexpf(1, -15), ldexpf(1, -15), 8, 1.0f},
      {"C=1, 16 products of -2^-30 (sum -1/8 ulp(1))", -ldexpf(1, -15), ldexpf(1, -15), 8, 1.0f},
      {"C=1, 16 products of -2^-40", -ldexpf(1, -20), ldexpf(1, -20), 8, 1.0f},
      {"C=1024, 16 products of +2^-18 (sum 2^-14 = 1/2 ulp(1024)... tie)", ldexpf(1, -9), ldexpf(1, -9), 8, 1024.0f},
      {"C=1024, 16 products of 1.25*2^-18", 1.25f * ldexpf(1, -9), ldexpf(1, -9), 8, 1024.0f},
  };
  for (const Case& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.a, c.b, c.nk, c.c, d);
    float r; hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
    double exact = (double)c.c + 2.0 * c.nk * (double)c.a * (double)c.b;
    printf("%-72s -> %a   (exact %a, RN %a, ulps off RN %+.1f)\n", c.what, r, exact, (float)exact, (r - (float)exact) / (nextafterf(fabsf((float)exact), INFINITY) - fabsf((float)exact)));
  }
  // the fp32 pipe for comparison: K = 2 products
  hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, 0, -ldexpf(1, -16), ldexpf(1, -15), 1.0f, d);
  float r; hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
  printf("fp32 pipe: C=1, 2 products of -2^-31 -> %a\n", r);
  return 0;
}
