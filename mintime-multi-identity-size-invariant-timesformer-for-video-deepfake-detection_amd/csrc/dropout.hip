// nn.Dropout inside the TimeSformer (reference models/size_invariant_timesformer.py:66-70 -- between GEGLU and the second feed-forward
// Linear -- and :98-101 -- behind the attention's output projection), for attn-dropout / ff-dropout > 0 in train mode.  The shipped
// configs use 0, so these are plain streaming kernels beside the tuned path, which they leave untouched: with dropout on, the producers
// store fp32 and the passes below apply the multiplier m = keep / (1 - p) (a caller-drawn fp32 tensor) on the way to the operand planes.
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "gemm_core.hpp"
#include "planes.hpp"

using namespace mt;

namespace {

// planes of x * m (and, optionally, the fp32 product).  One wavefront per 32 x 16 block, lane = (row, 8-column half).
__global__ __launch_bounds__(256) void mul_planes_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ out,
                                                         int R, int C, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c + e < C) {
        v[e] = x[(int64_t)r * C + c + e] * m[(int64_t)r * C + c + e];
        if (out) out[(int64_t)r * C + c + e] = v[e];
      }
  }
  planes_store8(o, r, c, v);
}

// out = r + y * m
__global__ __launch_bounds__(256) void mul_add_kernel(const float* __restrict__ y, const float* __restrict__ m, const float* __restrict__ r,
                                                      float* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(y)[i], b = reinterpret_cast<const float4*>(m)[i], c = reinterpret_cast<const float4*>(r)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
  }
}

// GEGLU backward as a pass: h = a * gelu(g), u = (a_0, g_0, a_1, g_1, ...) interleaved as the forward epilogue stored it;
// du = [dh m gelu(g) | dh m a gelu'(g)]  ([rows][2 n_half]) as planes and, optionally, fp32 (the bias gradient's column sums).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ m, const float* __restrict__ u,
                                                        float* __restrict__ du, int R, int n_half, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R && c < 2 * n_half) {
    const bool gate = c >= n_half;                      // n_half % 8 == 0: the 8 columns sit in one half
    const int j0 = gate ? c - n_half : c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = j0 + e;
      float d = dh[(int64_t)r * n_half + j];
      if (m) d *= m[(int64_t)r * n_half + j];
      const float2 ag = *reinterpret_cast<const float2*>(u + (int64_t)r * 2 * n_half + 2 * j);
      float gl, gr;
      gelu_erf_both(ag.y, gl, gr);
      v[e] = gate ? d * ag.x * gr : d * gl;
      if (du) du[(int64_t)r * 2 * n_half + c + e] = v[e];
    }
  }
  planes_store8(o, r, c, v);
}

PlaneRef plane_ref(void* planes, int rows, int cols) {
  const int cb16 = (cols + 15) >> 4, rp = (rows + 31) & ~31;
  return PlaneRef{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
}

}  // namespace

extern "C" int mt_mul_planes(const float* x, const float* m, void* planes, float* out, int rows, int cols, void* stream) {
  if (!x || !m || !planes || rows <= 0 || cols <= 0 || ((uintptr_t)planes & 15)) return fail(MT_ERR_ARG, "mt_mul_planes: bad arguments");
  const PlaneRef o = plane_ref(planes, rows, cols);
  hipLaunchKernelGGL(mul_planes_kernel, dim3((o.cb16 + 3) / 4, o.rows_pad / 32), dim3(256), 0, (hipStream_t)stream, x, m, out, rows, cols, o);
  return check_launch("mt_mul_planes");
}

extern "C" int mt_mul_add(const float* y, const float* m, const float* r, float* out, int64_t n, void* stream) {
  if (!y || !m || !r || !out || n <= 0 || (n & 3)) return fail(MT_ERR_ARG, "mt_mul_add: bad arguments (n %% 4 == 0)");
  if (((uintptr_t)y | (uintptr_t)m | (uintptr_t)r | (uintptr_t)out) & 15) return fail(MT_ERR_ARG, "mt_mul_add: 16-byte alignment");
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mul_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, m, r, out, n4);
  return check_launch("mt_mul_add");
}

extern "C" int mt_geglu_bwd(const float* dh, const float* m, const float* u, void* du_planes, float* du, int rows, int n_half, void* stream) {
  if (!dh || !u || !du_planes || rows <= 0 || n_half <= 0 || (n_half & 7) || ((uintptr_t)du_planes & 15) || ((uintptr_t)u & 7))
    return fail(MT_ERR_ARG, "mt_geglu_bwd: bad arguments (n_half %% 8 == 0)");
  const PlaneRef o = plane_ref(du_planes, rows, 2 * n_half);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((o.cb16 + 3) / 4, o.rows_pad / 32), dim3(256), 0, (hipStream_t)stream, dh, m, u, du, rows, n_half, o);
  return check_launch("mt_geglu_bwd");
}
