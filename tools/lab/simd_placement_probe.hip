// Lab probe: are the wavefronts of small workgroups spread over a CU's four SIMDs?  A VALU-bound kernel (dependent fma chains,
// no memory) is launched with the same total number of wavefronts as 1-, 2-, 4- and 6-wavefront workgroups; if workgroups always
// started on SIMD 0 the 1- and 2-wavefront forms would take 4x / 2x as long.  Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/p probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-7f); }
  if (a + c + d + b == 12345.f) out[0] = a;
}
// the same with 168 VGPRs allocated (3 wavefronts per SIMD): does a second 6-wavefront workgroup fit beside the first?
__global__ void spin168(float* out, int iters) {
  asm volatile("v_mov_b32 v167, 0" ::: "v167");
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-7f); }
  if (a + c + d + b == 12345.f) out[0] = a;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int waves_total = 256 * 12 * 8;                   // 12 wavefronts per CU at a time would be 3 per SIMD, 8 rounds
  for (int wpb : {1, 2, 3, 4, 6, 8, 12}) {
    const int blocks = waves_total / wpb;
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(64 * wpb), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(64 * wpb), 0, 0, out, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("wavefronts per workgroup %2d: %8.3f ms\n", wpb, ms);
  }
  for (int wpb : {1, 2, 3, 4, 6, 12}) {
    const int blocks = waves_total / wpb;
    hipLaunchKernelGGL(spin168, dim3(blocks), dim3(64 * wpb), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin168, dim3(blocks), dim3(64 * wpb), 0, 0, out, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin168, 64 * wpb, 0);
    printf("168 VGPRs, wavefronts per workgroup %2d: %8.3f ms   (occupancy API: %d workgroups per CU)\n", wpb, ms, nb);
  }
  return 0;
}
