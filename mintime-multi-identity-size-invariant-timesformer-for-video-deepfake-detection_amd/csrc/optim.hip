// Next-row f4: the two ends of a training step that sit outside the networks -- BCE-with-logits on the [B,1] logits (forward value
// and gradient in one launch; reference train.py:261,367-368 does it on the CPU after a D2H copy) and the SGD update over every
// parameter in ONE multi-tensor launch (reference train.py:186 torch.optim.SGD(lr, weight_decay), :378).
#include "common.hpp"
#include <stdint.h>

namespace {
using namespace mt;

// loss = mean_i (1 - y) x + (1 + (pw - 1) y) softplus(-x);   dloss/dx_i = ((1 - y) - (1 + (pw - 1) y) sigmoid(-x)) / n
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ y, float pos_weight,
                                                         float* __restrict__ loss, float* __restrict__ dx, int n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float xi = x[i], yi = y[i];
    const float lw = 1.0f + (pos_weight - 1.0f) * yi;
    const float sp = log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.f);            // softplus(-x), the stable form torch uses
    acc += (1.0f - yi) * xi + lw * sp;
    if (dx) dx[i] = ((1.0f - yi) - lw / (1.0f + expf(xi))) / (float)n;        // sigmoid(-x) = 1 / (1 + e^x)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

struct SgdItem { float* p; const float* g; int64_t n; int64_t block0; };     // block0 = first 4096-element block of this tensor
constexpr int SGD_CHUNK = 4096;

__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdItem* __restrict__ items, int count, float lr, float wd) {
  // which tensor does this block belong to: binary search over the (sorted) first-block table
  int lo = 0, hi = count - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const SgdItem it = items[lo];
  const int64_t off = (b - it.block0) * SGD_CHUNK;
  float* p = it.p + off;
  const float* g = it.g + off;
  const int64_t left = it.n - off;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g) & 15) == 0);
#pragma unroll
  for (int i = 0; i < SGD_CHUNK / 1024; ++i) {
    const int e = (i * 256 + threadIdx.x) * 4;
    if (vec && e + 3 < left) {
      float4 pv = *reinterpret_cast<float4*>(p + e);
      const float4 gv = *reinterpret_cast<const float4*>(g + e);
      pv.x -= lr * fmaf(wd, pv.x, gv.x); pv.y -= lr * fmaf(wd, pv.y, gv.y);
      pv.z -= lr * fmaf(wd, pv.z, gv.z); pv.w -= lr * fmaf(wd, pv.w, gv.w);
      *reinterpret_cast<float4*>(p + e) = pv;
    } else {
      for (int k = 0; k < 4; ++k)
        if (e + k < left) p[e + k] -= lr * fmaf(wd, p[e + k], g[e + k]);
    }
  }
}

// Transposes of many small matrices in ONE launch (the TimeSformer's 54 Linear weights, once per step: their data gradients run in
// the k-contiguous NT form over W^T, tsf_engine.py).  One 32 x 32 tile per block through LDS; the table maps blocks to matrices
// like sgd_multi's.  (As 54 torch copy kernels this cost the host ~2 ms per step and the side queue 0.6 ms of launch gaps.)
struct TransposeItem { const float* src; float* dst; int64_t rows; int64_t cols; int64_t tile0; };    // dst [cols, rows] = src [rows, cols]^T

__global__ __launch_bounds__(256) void transpose_multi_kernel(const TransposeItem* __restrict__ items, int count) {
  __shared__ float tile[32][33];
  int lo = 0, hi = count - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const TransposeItem it = items[lo];
  const int64_t tcols = (it.cols + 31) / 32, t = b - it.tile0;
  const int64_t r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < it.rows && c < it.cols) tile[ty + 8 * i][tx] = it.src[r * it.cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < it.rows && c < it.cols) it.dst[c * it.rows + r] = tile[tx][ty + 8 * i];
  }
}

// torch.optim.Adam / AdamW (train.py:187-190; amsgrad off, maximize off), same per-element operation order as torch's
// single-tensor path: [AdamW: p *= 1 - lr*wd] [Adam: g += wd*p]; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The bias corrections come in precomputed (host doubles -> float).
struct AdamItem { float* p; const float* g; float* m; float* v; int64_t n; int64_t block0; };

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamItem* __restrict__ items, int count, float lr, float wd, float omb1,
                                                         float b2, float omb2, float eps, float step_size, float bc2_sqrt, int decoupled) {
  int lo = 0, hi = count - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const AdamItem it = items[lo];
  const int64_t off = (b - it.block0) * SGD_CHUNK;
  const int64_t left = it.n - off;
#pragma unroll
  for (int i = 0; i < SGD_CHUNK / 256; ++i) {
    const int e = i * 256 + threadIdx.x;          // tensors are not 16-byte aligned in general (flat views): scalar, coalesced
    if (e < left) {
      float p = it.p[off + e], g = it.g[off + e], m = it.m[off + e], v = it.v[off + e];
      if (decoupled) p *= 1.0f - lr * wd;
      else if (wd != 0.f) g = fmaf(wd, p, g);
      m = m + (g - m) * omb1;                     // torch: exp_avg.lerp_(grad, 1 - beta1)   (1 - beta rounded from double, like torch)
      v = v * b2 + omb2 * g * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
      const float denom = sqrtf(v) / bc2_sqrt + eps;
      p -= step_size * (m / denom);
      it.p[off + e] = p; it.m[off + e] = m; it.v[off + e] = v;
    }
  }
}

}  // namespace

extern "C" int mt_adam_multi(const void* items, int count, int64_t total_blocks, float lr, float weight_decay, double beta1, double beta2,
                             float eps, float step_size, float bias_correction2_sqrt, int decoupled_weight_decay, void* stream) {
  if (!items || count <= 0 || total_blocks <= 0) return fail(MT_ERR_ARG, "mt_adam_multi: empty table");
  if (total_blocks > 0x7fffffff) return fail(MT_ERR_ARG, "mt_adam_multi: too many blocks");
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const AdamItem*>(items), count, lr, weight_decay, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), eps, step_size,
                     bias_correction2_sqrt, decoupled_weight_decay);
  return check_launch("mt_adam_multi");
}

extern "C" int mt_bce_logits(const float* logits, const float* labels, float pos_weight, float* loss, float* dlogits, int n,
                             void* stream) {
  if (!logits || !labels || !loss || n <= 0) return fail(MT_ERR_ARG, "mt_bce_logits: null pointer / empty batch");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, pos_weight, loss, dlogits, n);
  return check_launch("mt_bce_logits");
}

extern "C" int mt_transpose_multi(const void* items, int count, int64_t total_tiles, void* stream) {
  if (!items || count <= 0 || total_tiles <= 0) return fail(MT_ERR_ARG, "mt_transpose_multi: empty table");
  hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const TransposeItem*)items, count);
  return check_launch("mt_transpose_multi");
}

extern "C" int mt_sgd_multi(const void* items, int count, int64_t total_blocks, float lr, float weight_decay, void* stream) {
  if (!items || count <= 0 || total_blocks <= 0) return fail(MT_ERR_ARG, "mt_sgd_multi: empty table");
  if (total_blocks > 0x7fffffff) return fail(MT_ERR_ARG, "mt_sgd_multi: too many blocks");
  hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const SgdItem*>(items), count, lr, weight_decay);
  return check_launch("mt_sgd_multi");
}
