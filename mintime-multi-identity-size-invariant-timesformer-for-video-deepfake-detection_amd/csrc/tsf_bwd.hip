// Size-Invariant TimeSformer backward kernels other than the dense contractions (gfx950).
//
// The reference has no hand-written backward: it is whatever torch autograd derives from
// models/size_invariant_timesformer.py (:80-87 attn, :109-144 Attention.forward, :18-26 PreNorm, :231-248
// embeddings, :270-276 head).  These kernels are the analytic adjoints of csrc/tsf_fwd.hip, on the same layouts:
//   mt_head_bwd        LayerNorm + Linear on the cls row
//   mt_layernorm_bwd   dx += LN'(dy), dgamma/dbeta accumulated
//   mt_colsum          bias gradients (column sums with optional row map)
//   mt_attn_bwd        divided attention core: recomputes the probabilities from q,k (nothing but qkv was saved)
//   mt_embed_bwd       cls / pos_emb / size_emb scatter-adds
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include <stdlib.h>
#include <float.h>

using namespace mt;

#include "planes.hpp"
#include "det.hpp"
namespace {

constexpr int DH = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---------------------------------------------------------------------------------------- LayerNorm backward
// one wavefront per row, grid-strided so each wave keeps dgamma/dbeta partials in registers.  NI = float4 slots per lane
// (ceil(D / 256)): sized to the row width, the register arrays cost 100 instead of 200 VGPRs at D = 512, which lets these
// wavefronts start on SIMDs that the weight-gradient stream's GEMM waves have mostly filled.
template <int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int rows, int D, int accumulate,
                                                            float* __restrict__ dxsum, int skip_period, const float* dx_in, const DetLog det) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nq = D >> 2;
  float4 dg[NI], db[NI], g[NI], ds[NI];     // ds: column sums of the UPDATED dx (the bias gradient of whoever consumes dx next)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; ds[i] = dg[i];
    const int q = lane + i * 64;
    g[i] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // two rows per trip: a wave walks ~12 rows and each row is a dependent chain of load -> wave reduction -> store, so the
  // second row's loads ride under the first row's reductions
  for (int row0 = wave_global; row0 < rows; row0 += 2 * nwaves) {
    const int row1 = row0 + nwaves;
    const bool has1 = row1 < rows;
    const int r1 = has1 ? row1 : row0;
    const float mean0 = stats[2 * (int64_t)row0], rstd0 = stats[2 * (int64_t)row0 + 1];
    const float mean1 = stats[2 * (int64_t)r1], rstd1 = stats[2 * (int64_t)r1 + 1];
    const float4* xr0 = reinterpret_cast<const float4*>(x + (int64_t)row0 * D);
    const float4* dyr0 = reinterpret_cast<const float4*>(dy + (int64_t)row0 * D);
    const float4* xr1 = reinterpret_cast<const float4*>(x + (int64_t)r1 * D);
    const float4* dyr1 = reinterpret_cast<const float4*>(dy + (int64_t)r1 * D);
    float4 xh0[NI], gy0[NI], xh1[NI], gy1[NI];
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    const float w1 = has1 ? 1.f : 0.f;          // the duplicated tail row must not count twice in dgamma / dbeta
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) {
        const float4 xv0 = xr0[q], d0 = dyr0[q], xv1 = xr1[q], d1 = dyr1[q];
        xh0[i] = make_float4((xv0.x - mean0) * rstd0, (xv0.y - mean0) * rstd0, (xv0.z - mean0) * rstd0, (xv0.w - mean0) * rstd0);
        xh1[i] = make_float4((xv1.x - mean1) * rstd1, (xv1.y - mean1) * rstd1, (xv1.z - mean1) * rstd1, (xv1.w - mean1) * rstd1);
        gy0[i] = make_float4(d0.x * g[i].x, d0.y * g[i].y, d0.z * g[i].z, d0.w * g[i].w);
        gy1[i] = make_float4(d1.x * g[i].x, d1.y * g[i].y, d1.z * g[i].z, d1.w * g[i].w);
        a1 += gy0[i].x + gy0[i].y + gy0[i].z + gy0[i].w;
        a2 += gy0[i].x * xh0[i].x + gy0[i].y * xh0[i].y + gy0[i].z * xh0[i].z + gy0[i].w * xh0[i].w;
        b1 += gy1[i].x + gy1[i].y + gy1[i].z + gy1[i].w;
        b2 += gy1[i].x * xh1[i].x + gy1[i].y * xh1[i].y + gy1[i].z * xh1[i].z + gy1[i].w * xh1[i].w;
        dg[i].x += d0.x * xh0[i].x + w1 * d1.x * xh1[i].x; dg[i].y += d0.y * xh0[i].y + w1 * d1.y * xh1[i].y;
        dg[i].z += d0.z * xh0[i].z + w1 * d1.z * xh1[i].z; dg[i].w += d0.w * xh0[i].w + w1 * d1.w * xh1[i].w;
        db[i].x += d0.x + w1 * d1.x; db[i].y += d0.y + w1 * d1.y; db[i].z += d0.z + w1 * d1.z; db[i].w += d0.w + w1 * d1.w;
      }
    }
    a1 = wave_sum(a1) / (float)D; a2 = wave_sum(a2) / (float)D;
    b1 = wave_sum(b1) / (float)D; b2 = wave_sum(b2) / (float)D;
    float4* dxr0 = reinterpret_cast<float4*>(dx + (int64_t)row0 * D);
    float4* dxr1 = reinterpret_cast<float4*>(dx + (int64_t)r1 * D);
    const float4* pin0 = reinterpret_cast<const float4*>(dx_in + (int64_t)row0 * D);     // == dx when accumulating in place
    const float4* pin1 = reinterpret_cast<const float4*>(dx_in + (int64_t)r1 * D);
    // rows with row % skip_period == 0 (the cls rows of the token matrix) are left out of dxsum when skip_period > 0
    const float c0 = (skip_period > 0 && row0 % skip_period == 0) ? 0.f : 1.f;
    const float c1 = (!has1 || (skip_period > 0 && row1 % skip_period == 0)) ? 0.f : 1.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) {
        float4 o;
        o.x = rstd0 * (gy0[i].x - a1 - xh0[i].x * a2); o.y = rstd0 * (gy0[i].y - a1 - xh0[i].y * a2);
        o.z = rstd0 * (gy0[i].z - a1 - xh0[i].z * a2); o.w = rstd0 * (gy0[i].w - a1 - xh0[i].w * a2);
        if (accumulate) { const float4 p = pin0[q]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        dxr0[q] = o;
        ds[i].x += c0 * o.x; ds[i].y += c0 * o.y; ds[i].z += c0 * o.z; ds[i].w += c0 * o.w;
        if (has1) {
          float4 t;
          t.x = rstd1 * (gy1[i].x - b1 - xh1[i].x * b2); t.y = rstd1 * (gy1[i].y - b1 - xh1[i].y * b2);
          t.z = rstd1 * (gy1[i].z - b1 - xh1[i].z * b2); t.w = rstd1 * (gy1[i].w - b1 - xh1[i].w * b2);
          if (accumulate) { const float4 p = pin1[q]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
          dxr1[q] = t;
          ds[i].x += c1 * t.x; ds[i].y += c1 * t.y; ds[i].z += c1 * t.z; ds[i].w += c1 * t.w;
        }
      }
    }
  }
  // block reduction over the 4 wavefronts, then one atomic per column per block.  One 16 KB buffer reused for the three sums:
  // the former 48 KB kept this kernel's blocks off CUs whose LDS the weight-gradient stream's GEMM blocks had filled (in-step the
  // launch stretched from 76 to 157 us).
  __shared__ float red[4][1024];
  const int wv = threadIdx.x >> 6;
#pragma unroll 1
  for (int which = 0; which < 3; ++which) {
    if (which == 2 && !dxsum) break;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) *reinterpret_cast<float4*>(&red[wv][4 * q]) = which == 0 ? dg[i] : (which == 1 ? db[i] : ds[i]);
    }
    __syncthreads();
    float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    for (int c = threadIdx.x; c < D; c += 256) {
      const float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      if (det.vals) det_put(det, which, blockIdx.x, c, v);        // deterministic mode: one log row per block, summed in block order
      else atomicAdd(dst + c, v);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------- LayerNorm backward, split by queue
// The fused kernel above keeps three [D] column sums per wavefront (100+ VGPRs) and meets in LDS: in-step it is the most stretched
// kernel of the main queue (35 us serialised, ~120 us next to a weight-gradient GEMM whose three blocks per CU leave 74 VGPRs per
// SIMD lane).  Only dx is on the critical path; the column sums are PARAMETER gradients (gamma, beta, the next Linear's bias).  So:
//   rows kernel (main stream): dx_out = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)) + dx_in, one row per
//     wavefront and trip, no LDS, few registers -- its wavefronts fit next to the weight-gradient blocks;
//   cols kernel (weight-gradient stream): dgamma += sum_r dy*xhat, dbeta += sum_r dy, dxsum += sum_r dx_out (rows with
//     r % skip_period == 0 left out) -- re-reads dy, x, dx_out (75 MB at B = 32) off the critical path.
template <int NI, bool KEEP = false>
__device__ __forceinline__ void layernorm_bwd_rows_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        float* __restrict__ dx, const float* __restrict__ dx_in, int rows, int D,
                                                        const PlaneRef dxp) {
  // register diet (56 VGPRs are what three weight-gradient waves leave on a SIMD lane): pass 1 keeps only the two row sums, pass 2
  // re-reads x / dy / gamma (the row is 2 x 2 KB and still in L1 / L2) one quad at a time
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int nq = D >> 2;
  const float inv_d = 1.0f / (float)D;
  if (dxp.p && blockIdx.x == 0)                      // padding rows of the plane tensor's last row block: zeros
    for (int row = rows + (threadIdx.x >> 6); row < dxp.rows_pad; row += 4)
      for (int q = lane; q < nq; q += 64) planes_store4(dxp, row, q * 4, 0.f, 0.f, 0.f, 0.f);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const float mean = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* dyr = reinterpret_cast<const float4*>(dy + (int64_t)row * D);
    const float4* pin = reinterpret_cast<const float4*>(dx_in + (int64_t)row * D);
    float4* dxr = reinterpret_cast<float4*>(dx + (int64_t)row * D);
    if constexpr (KEEP) {
      // the row's operands stay in registers between the two passes (every load of the row is in flight at once): the weight-gradient
      // GEMMs this kernel runs next to take 116-128 VGPRs per wave now, three blocks per CU -- 128 registers per lane are left over
      float4 xv[NI], d[NI], g[NI], pv[NI];
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int q = min(lane + i * 64, nq - 1);
        xv[i] = xr[q]; d[i] = dyr[q]; g[i] = reinterpret_cast<const float4*>(gamma)[q]; pv[i] = pin[q];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float w = lane + i * 64 < nq ? 1.f : 0.f;
        const float gx = d[i].x * g[i].x, gy_ = d[i].y * g[i].y, gz = d[i].z * g[i].z, gw = d[i].w * g[i].w;
        a1 += w * (gx + gy_ + gz + gw);
        a2 += w * rstd * (gx * (xv[i].x - mean) + gy_ * (xv[i].y - mean) + gz * (xv[i].z - mean) + gw * (xv[i].w - mean));
      }
      a1 = wave_sum(a1) * inv_d; a2 = wave_sum(a2) * inv_d;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int q = lane + i * 64;
        if (q < nq) {
          float4 o;
          o.x = rstd * (d[i].x * g[i].x - a1 - (xv[i].x - mean) * rstd * a2) + pv[i].x; o.y = rstd * (d[i].y * g[i].y - a1 - (xv[i].y - mean) * rstd * a2) + pv[i].y;
          o.z = rstd * (d[i].z * g[i].z - a1 - (xv[i].z - mean) * rstd * a2) + pv[i].z; o.w = rstd * (d[i].w * g[i].w - a1 - (xv[i].w - mean) * rstd * a2) + pv[i].w;
          dxr[q] = o;
          if (dxp.p) planes_store4(dxp, row, q * 4, o.x, o.y, o.z, o.w);
        }
      }
      continue;
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll 1
    for (int i = 0; i < NI; ++i) {
      const int q = min(lane + i * 64, nq - 1);             // clamped: every lane loads (D = 512: all 128 quads are live)
      const float w = lane + i * 64 < nq ? 1.f : 0.f;
      const float4 xv = xr[q], d = dyr[q], g = reinterpret_cast<const float4*>(gamma)[q];
      const float gx = d.x * g.x, gy_ = d.y * g.y, gz = d.z * g.z, gw = d.w * g.w;
      a1 += w * (gx + gy_ + gz + gw);
      a2 += w * rstd * (gx * (xv.x - mean) + gy_ * (xv.y - mean) + gz * (xv.z - mean) + gw * (xv.w - mean));
    }
    a1 = wave_sum(a1) * inv_d; a2 = wave_sum(a2) * inv_d;
#pragma unroll 1
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) {
        const float4 xv = xr[q], d = dyr[q], g = reinterpret_cast<const float4*>(gamma)[q], p = pin[q];
        float4 o;
        o.x = rstd * (d.x * g.x - a1 - (xv.x - mean) * rstd * a2) + p.x; o.y = rstd * (d.y * g.y - a1 - (xv.y - mean) * rstd * a2) + p.y;
        o.z = rstd * (d.z * g.z - a1 - (xv.z - mean) * rstd * a2) + p.z; o.w = rstd * (d.w * g.w - a1 - (xv.w - mean) * rstd * a2) + p.w;
        dxr[q] = o;
        if (dxp.p) planes_store4(dxp, row, q * 4, o.x, o.y, o.z, o.w);
      }
    }
  }
}

// D <= 512: capped at the 56 VGPRs that are free next to three weight-gradient waves (the allocator rounds their 146 up to 152)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56))) void layernorm_bwd_rows_kernel2(
    const float* dy, const float* x, const float* stats, const float* gamma, float* dx, const float* dx_in, int rows, int D, PlaneRef dxp) {
  layernorm_bwd_rows_body<2>(dy, x, stats, gamma, dx, dx_in, rows, D, dxp);
}
// D <= 512, operands kept in registers (the default next to the plane-operand weight gradients; MT_LN_ROWS_KEEP=0: the 56-VGPR form)
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel2k(
    const float* dy, const float* x, const float* stats, const float* gamma, float* dx, const float* dx_in, int rows, int D, PlaneRef dxp) {
  layernorm_bwd_rows_body<2, true>(dy, x, stats, gamma, dx, dx_in, rows, D, dxp);
}
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel4(const float* dy, const float* x, const float* stats, const float* gamma,
                                                                   float* dx, const float* dx_in, int rows, int D, PlaneRef dxp) {
  layernorm_bwd_rows_body<4>(dy, x, stats, gamma, dx, dx_in, rows, D, dxp);
}

// Rows kernel + per-block column partial sums (round 5).  The split above made the column sums a second pass on the weight-gradient
// queue: 75 MB re-read per LayerNorm, 2.3 ms of that queue per step.  Here the wavefront that holds a row's dy / x / dx_out in registers
// also adds them into three [D] partial sums (24 more VGPRs: 108, still next to three 120-VGPR weight-gradient waves), the block's
// four wavefronts meet in 8 KB of LDS once at the end, and the block stores ONE row of partials[blocks][3][D] -- no atomics, no
// re-read.  layernorm_cols_reduce_kernel (weight-gradient queue) adds the block rows in block order into dgamma / dbeta / dx_colsum:
// 6 MB read instead of 75, and a fixed summation order (nothing for the deterministic mode to replace).
__global__ __launch_bounds__(256) void layernorm_bwd_rows_sums_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
    float* __restrict__ dx, const float* __restrict__ dx_in, int rows, int D, PlaneRef dxp, float* __restrict__ partials, int skip_period) {
  constexpr int NI = 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  const int nq = D >> 2;
  const float inv_d = 1.0f / (float)D;
  if (dxp.p && blockIdx.x == 0)                      // padding rows of the plane tensor's last row block: zeros
    for (int row = rows + wv; row < dxp.rows_pad; row += 4)
      for (int q = lane; q < nq; q += 64) planes_store4(dxp, row, q * 4, 0.f, 0.f, 0.f, 0.f);
  float4 dg[NI], db[NI], ds[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) { dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; ds[i] = dg[i]; }
  for (int row = blockIdx.x * 4 + wv; row < rows; row += nwaves) {
    const float mean = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
    const float keep = (skip_period > 0 && row % skip_period == 0) ? 0.f : 1.f;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* dyr = reinterpret_cast<const float4*>(dy + (int64_t)row * D);
    const float4* pin = reinterpret_cast<const float4*>(dx_in + (int64_t)row * D);
    float4* dxr = reinterpret_cast<float4*>(dx + (int64_t)row * D);
    float4 xv[NI], d[NI], g[NI], pv[NI];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = min(lane + i * 64, nq - 1);
      xv[i] = xr[q]; d[i] = dyr[q]; g[i] = reinterpret_cast<const float4*>(gamma)[q]; pv[i] = pin[q];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float w = lane + i * 64 < nq ? 1.f : 0.f;
      xv[i].x = (xv[i].x - mean) * rstd; xv[i].y = (xv[i].y - mean) * rstd; xv[i].z = (xv[i].z - mean) * rstd; xv[i].w = (xv[i].w - mean) * rstd;
      const float gx = d[i].x * g[i].x, gy_ = d[i].y * g[i].y, gz = d[i].z * g[i].z, gw = d[i].w * g[i].w;
      a1 += w * (gx + gy_ + gz + gw);
      a2 += w * (gx * xv[i].x + gy_ * xv[i].y + gz * xv[i].z + gw * xv[i].w);
    }
    a1 = wave_sum(a1) * inv_d; a2 = wave_sum(a2) * inv_d;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) {
        float4 o;
        o.x = rstd * (d[i].x * g[i].x - a1 - xv[i].x * a2) + pv[i].x; o.y = rstd * (d[i].y * g[i].y - a1 - xv[i].y * a2) + pv[i].y;
        o.z = rstd * (d[i].z * g[i].z - a1 - xv[i].z * a2) + pv[i].z; o.w = rstd * (d[i].w * g[i].w - a1 - xv[i].w * a2) + pv[i].w;
        dxr[q] = o;
        if (dxp.p) planes_store4(dxp, row, q * 4, o.x, o.y, o.z, o.w);
        dg[i].x += d[i].x * xv[i].x; dg[i].y += d[i].y * xv[i].y; dg[i].z += d[i].z * xv[i].z; dg[i].w += d[i].w * xv[i].w;
        db[i].x += d[i].x; db[i].y += d[i].y; db[i].z += d[i].z; db[i].w += d[i].w;
        ds[i].x += keep * o.x; ds[i].y += keep * o.y; ds[i].z += keep * o.z; ds[i].w += keep * o.w;
      }
    }
  }
  __shared__ float red[4][512];
  auto meet = [&](const float4 (&v)[NI], int which) {        // (three explicit calls: a dynamic pick among dg / db / ds goes to scratch)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) *reinterpret_cast<float4*>(&red[wv][4 * q]) = v[i];
    }
    __syncthreads();
    float* dst = partials + ((int64_t)blockIdx.x * 3 + which) * D;
    for (int c = threadIdx.x; c < D; c += 256) dst[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    __syncthreads();
  };
  meet(dg, 0);
  meet(db, 1);
  meet(ds, 2);
}

// out_which[c] += sum over blocks of partials[b][which][c], in a FIXED order: 64 columns x 16 phases per block (1024 threads), phase p
// takes blocks p, p + 16, ... in four independent chains (the first version -- 4 phases, one chain -- was 256 dependent loads per
// thread: 198 us in the step for 6 MB), chains then phases added in index order
__global__ __launch_bounds__(1024) void layernorm_cols_reduce_kernel(const float* __restrict__ partials, int blocks, int D,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     float* __restrict__ dxsum) {
  const int which = blockIdx.y;
  float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
  if (!dst) return;
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < D) {
    const float* src = partials + (int64_t)which * D + c;
    const int64_t stride = (int64_t)3 * D;
    int b = ph;
    for (; b + 48 < blocks; b += 64) {
      s0 += src[(int64_t)b * stride]; s1 += src[(int64_t)(b + 16) * stride];
      s2 += src[(int64_t)(b + 32) * stride]; s3 += src[(int64_t)(b + 48) * stride];
    }
    for (; b < blocks; b += 16) s0 += src[(int64_t)b * stride];
  }
  red[ph][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < D) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    dst[c] += t;
  }
}

template <int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ stats, const float* __restrict__ dxn,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ dxsum, int skip_period, int rows, int D, const DetLog det) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int nq = D >> 2;
  float4 dg[NI], db[NI], ds[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) { dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; ds[i] = dg[i]; }
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const float mean = stats[2 * (int64_t)row], rstd = stats[2 * (int64_t)row + 1];
    const float c = (skip_period > 0 && row % skip_period == 0) ? 0.f : 1.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) {
        const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)row * D)[q];
        const float4 d = reinterpret_cast<const float4*>(dy + (int64_t)row * D)[q];
        dg[i].x += d.x * ((xv.x - mean) * rstd); dg[i].y += d.y * ((xv.y - mean) * rstd);
        dg[i].z += d.z * ((xv.z - mean) * rstd); dg[i].w += d.w * ((xv.w - mean) * rstd);
        db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        if (dxsum) {
          const float4 o = reinterpret_cast<const float4*>(dxn + (int64_t)row * D)[q];
          ds[i].x += c * o.x; ds[i].y += c * o.y; ds[i].z += c * o.z; ds[i].w += c * o.w;
        }
      }
    }
  }
  __shared__ float red[4][1024];
  const int wv = threadIdx.x >> 6;
#pragma unroll 1
  for (int which = 0; which < 3; ++which) {
    if (which == 2 && !dxsum) break;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = lane + i * 64;
      if (q < nq) *reinterpret_cast<float4*>(&red[wv][4 * q]) = which == 0 ? dg[i] : (which == 1 ? db[i] : ds[i]);
    }
    __syncthreads();
    float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
    for (int cidx = threadIdx.x; cidx < D; cidx += 256) {
      const float v = red[0][cidx] + red[1][cidx] + red[2][cidx] + red[3][cidx];
      if (det.vals) det_put(det, which, blockIdx.x, cidx, v);     // deterministic mode: one log row per block, summed in block order
      else atomicAdd(dst + cidx, v);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------- column sums (bias grads)
// out[n] += sum_m A[map(m)*lda + n];  block = 64 columns x 4 row-lanes, grid (ceil(N/64), row chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int64_t lda, int gin, int gout, int off,
                                                     int M, int N, float* __restrict__ out, int rows_per_block, const DetLog det) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl;
  const int m0 = blockIdx.y * rows_per_block;
  const int m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (n < N) {
    for (int m = m0 + rl; m < m1; m += 4) {
      int64_t r = m;
      if (gin) { const int g = m / gin; r = (int64_t)g * gout + off + (m - g * gin); }
      s += A[r * lda + n];
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && n < N) {
    const float v = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (det.vals) {                                               // deterministic mode: group = 64-column block, rank = row chunk
      det_put(det, blockIdx.x, blockIdx.y, cl, v);
      if (blockIdx.y == 0 && cl == 0) det_base(det, blockIdx.x, (int64_t)blockIdx.x * 64);
    } else atomicAdd(out + n, v);
  }
}

// ---------------------------------------------------------------------------------------- head backward
// one block (D threads) loops over clips; classes C small (1).
__global__ void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ x, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ w, float* __restrict__ dx,
                                float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dw,
                                float* __restrict__ dbias, int B, int N, int D, int C, float eps) {
  __shared__ float red[2][16];
  const int i = threadIdx.x;             // feature index, blockDim.x == D (<= 1024)
  const int lane = i & 63, wv = i >> 6, nw = blockDim.x >> 6;
  auto block_sum2 = [&](float a, float b, float& oa, float& ob) {
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if (lane == 0) { red[0][wv] = a; red[1][wv] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
    for (int k = 0; k < nw; ++k) { sa += red[0][k]; sb += red[1][k]; }
    oa = sa; ob = sb;
  };
  float dg = 0.f, db = 0.f;
  const float gi = gamma[i], bi = beta[i];
  for (int b = 0; b < B; ++b) {
    const float xv = x[(int64_t)b * N * D + i];
    float s, dummy;
    block_sum2(xv, 0.f, s, dummy);
    const float mean = s / (float)D;
    float ss;
    block_sum2((xv - mean) * (xv - mean), 0.f, ss, dummy);
    const float rstd = rsqrtf(ss / (float)D + eps);
    const float xh = (xv - mean) * rstd;
    const float xn = xh * gi + bi;
    float dxn = 0.f;
    for (int c = 0; c < C; ++c) {
      const float dl = dlogits[b * C + c];
      dxn += dl * w[(int64_t)c * D + i];
      dw[(int64_t)c * D + i] += dl * xn;          // single block: plain accumulation
    }
    dg += dxn * xh; db += dxn;
    const float gy = dxn * gi;
    float c1, c2;
    block_sum2(gy, gy * xh, c1, c2);
    c1 /= (float)D; c2 /= (float)D;
    dx[(int64_t)b * N * D + i] = rstd * (gy - c1 - xh * c2);
  }
  dgamma[i] += dg; dbeta[i] += db;
  if (i < C) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[b * C + i];
    dbias[i] += s;
  }
}

// The same with one block per clip (default mode): the single block above walks the B clips one after the other with six block barriers
// each -- 70 us at B = 32 on the head of the backward pass, where nothing else can run yet.  Parameter gradients meet in atomics.
__global__ void head_bwd_clips_kernel(const float* __restrict__ dlogits, const float* __restrict__ x, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ w, float* __restrict__ dx,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dw,
                                      float* __restrict__ dbias, int B, int N, int D, int C, float eps) {
  __shared__ float red[2][16];
  const int i = threadIdx.x, b = blockIdx.x;
  const int lane = i & 63, wv = i >> 6, nw = blockDim.x >> 6;
  auto block_sum2 = [&](float a, float c, float& oa, float& oc) {
    a = wave_sum(a); c = wave_sum(c);
    __syncthreads();
    if (lane == 0) { red[0][wv] = a; red[1][wv] = c; }
    __syncthreads();
    float sa = 0.f, sc = 0.f;
    for (int k = 0; k < nw; ++k) { sa += red[0][k]; sc += red[1][k]; }
    oa = sa; oc = sc;
  };
  const float gi = gamma[i], bi = beta[i];
  const float xv = x[(int64_t)b * N * D + i];
  float s, dummy;
  block_sum2(xv, 0.f, s, dummy);
  const float mean = s / (float)D;
  float ss;
  block_sum2((xv - mean) * (xv - mean), 0.f, ss, dummy);
  const float rstd = rsqrtf(ss / (float)D + eps);
  const float xh = (xv - mean) * rstd;
  const float xn = xh * gi + bi;
  float dxn = 0.f;
  for (int c = 0; c < C; ++c) {
    const float dl = dlogits[b * C + c];
    dxn += dl * w[(int64_t)c * D + i];
    atomicAdd(dw + (int64_t)c * D + i, dl * xn);
  }
  atomicAdd(dgamma + i, dxn * xh);
  atomicAdd(dbeta + i, dxn);
  const float gy = dxn * gi;
  float c1, c2;
  block_sum2(gy, gy * xh, c1, c2);
  c1 /= (float)D; c2 /= (float)D;
  dx[(int64_t)b * N * D + i] = rstd * (gy - c1 - xh * c2);
  if (i < C) atomicAdd(dbias + i, dlogits[b * C + i]);
}

// ---------------------------------------------------------------------------------------- embeddings backward
// dcls += sum_b dx[b,0]; dpos[positions[b,t]] += dx[b,t]; dsize[size idx] += dx[b,t]   (atomics; tables pre-zeroed)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dcls,
                                                        float* __restrict__ dpos, float* __restrict__ dsize,
                                                        const int64_t* __restrict__ positions, const int* __restrict__ sizes,
                                                        int B, int N, int n, int F, int D, int pos_rows, int size_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * N) return;
  const int b = row / N, t = row - b * N;
  int64_t pi = positions ? positions[row] : (int64_t)t;
  int si = 0;
  if (t > 0 && sizes) si = sizes[b * F + (t - 1) / n];
  pi = pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi);          // same clamp as the forward (which raised the flag)
  si = si < 0 ? 0 : (si >= size_rows ? size_rows - 1 : si);
  const float* dr = dx + (int64_t)row * D;
  for (int i = lane; i < D; i += 64) {
    const float v = dr[i];
    if (t == 0 && dcls) atomicAdd(dcls + i, v);
    if (dpos) atomicAdd(dpos + pi * D + i, v);
    if (dsize) atomicAdd(dsize + (int64_t)si * D + i, v);
  }
}

// One block per (clip, frame slot): the n tokens of a slot share their size row, so the block sums them in registers and issues ONE
// atomic per column for it (the row-per-wavefront kernel above sends B N D atomics at ~21 live rows: 600 adds per address); position
// rows differ per token and keep one atomic each.  Block (b, 0) also carries the cls row.
__global__ __launch_bounds__(256) void embed_bwd_slots_kernel(const float* __restrict__ dx, float* __restrict__ dcls,
                                                              float* __restrict__ dpos, float* __restrict__ dsize,
                                                              const int64_t* __restrict__ positions, const int* __restrict__ sizes,
                                                              int B, int N, int n, int F, int D, int pos_rows, int size_rows) {
  const int b = blockIdx.x / F, f = blockIdx.x - b * F;
  int si = sizes ? sizes[b * F + f] : 0;
  si = si < 0 ? 0 : (si >= size_rows ? size_rows - 1 : si);
  const int row0 = b * N + 1 + f * n;
  for (int i = threadIdx.x; i < D; i += 256) {
    float acc = 0.f;
#pragma unroll 7
    for (int t = 0; t < n; ++t) {
      const int row = row0 + t;
      const float v = dx[(int64_t)row * D + i];
      acc += v;
      if (dpos) {
        int64_t pi = positions ? positions[row] : (int64_t)(1 + f * n + t);
        pi = pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi);
        atomicAdd(dpos + pi * D + i, v);
      }
    }
    if (dsize) atomicAdd(dsize + (int64_t)si * D + i, acc);
    if (f == 0) {                                      // the cls token of this clip: position row as given, size row 0
      const int row = b * N;
      const float v = dx[(int64_t)row * D + i];
      if (dcls) atomicAdd(dcls + i, v);
      if (dpos) {
        int64_t pi = positions ? positions[row] : 0;
        pi = pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi);
        atomicAdd(dpos + pi * D + i, v);
      }
      if (dsize) atomicAdd(dsize + i, v);
    }
  }
}

// Deterministic mode: one wavefront per (table, 64 columns), one lane per column, walks the token rows in order and adds each RUN of
// rows that share a table row to it (tokens of one frame share their position / size row, so runs are long): every element of the
// tables has one writer and a fixed summation order.  ~B N sequential, pipelined 4-byte loads per lane.
__global__ __launch_bounds__(64) void embed_bwd_det_kernel(const float* __restrict__ dx, float* __restrict__ dcls,
                                                           float* __restrict__ dpos, float* __restrict__ dsize,
                                                           const int64_t* __restrict__ positions, const int* __restrict__ sizes,
                                                           int B, int N, int n, int F, int D, int pos_rows, int size_rows) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int kind = blockIdx.y;                           // 0: cls, 1: positions, 2: sizes
  if (col >= D) return;
  if (kind == 0) {
    if (!dcls) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dx[(int64_t)b * N * D + col];
    dcls[col] += acc;
    return;
  }
  float* dst = kind == 1 ? dpos : dsize;
  if (!dst) return;
  int cur = -1;
  float run = 0.f;
  int b = 0, t = 0;
  for (int row = 0; row < B * N; ++row) {
    int idx;
    if (kind == 1) {
      int64_t pi = positions ? positions[row] : (int64_t)t;
      idx = (int)(pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi));
    } else {
      int si = 0;
      if (t > 0 && sizes) si = sizes[b * F + (t - 1) / n];
      idx = si < 0 ? 0 : (si >= size_rows ? size_rows - 1 : si);
    }
    const float v = dx[(int64_t)row * D + col];
    if (idx != cur) {
      if (cur >= 0) dst[(int64_t)cur * D + col] += run;
      cur = idx;
      run = 0.f;
    }
    run += v;
    if (++t == N) { t = 0; ++b; }
  }
  if (cur >= 0) dst[(int64_t)cur * D + col] += run;
}

// Deterministic mode since round 5: the run walk above is one wavefront per 64 columns over all B N token rows -- 5.9 ms per step at
// B = 32, on the main queue.  An embedding gradient is an index_add; what makes it order-dependent is floating-point addition.  So the
// token rows scatter into two-limb INTEGER accumulators (common.hpp stat_add: trunc(v) and round(frac * 2^44) as int64 atomics --
// associative, hence the same bits whatever order the wavefronts arrive in), one wavefront per token row like the default kernel, and
// a second pass decodes the table-shaped accumulator and adds it to the gradient.  ~0.3 ms for both tables.
__global__ __launch_bounds__(256) void embed_bwd_limb_scatter_kernel(const float* __restrict__ dx, const int64_t* __restrict__ positions,
                                                                    const int* __restrict__ sizes, int B, int N, int n, int F, int D,
                                                                    int rows, int kind, double* __restrict__ acc, int64_t limb) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * N) return;
  const int b = row / N, t = row - b * N;
  int64_t idx;
  if (kind == 1) idx = positions ? positions[row] : (int64_t)t;
  else idx = (t > 0 && sizes) ? sizes[b * F + (t - 1) / n] : 0;
  idx = idx < 0 ? 0 : (idx >= rows ? rows - 1 : idx);              // same clamp as the forward (which raised the flag)
  const float* dr = dx + (int64_t)row * D;
  for (int i = lane; i < D; i += 64) stat_add(acc + idx * D + i, limb, dr[i]);
}

__global__ __launch_bounds__(256) void embed_bwd_limb_decode_kernel(const double* __restrict__ acc, int64_t limb, float* __restrict__ dst,
                                                                   int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const long long hi = *reinterpret_cast<const long long*>(acc + i), lo = *reinterpret_cast<const long long*>(acc + limb + i);
    if (hi | lo) dst[i] += (float)stat_get(acc + i, limb);
  }
}

// ---------------------------------------------------------------------------------------- attention backward: cls query
// one block of CLS_W wavefronts per (b,h).  Writes dq (row 0) and INITIALISES dk, dv for every key row of this head with the cls
// query's contribution; the patch kernels then accumulate on top.  Keys are spread over all lanes of the block for the per-key
// work and over its wavefronts for the dq reduction.
#ifndef MT_CLS_W
#define MT_CLS_W 16
#endif
constexpr int CLS_W = MT_CLS_W;          // (16 wavefronts: the N keys in two trips of 256 -- with 4 the kernel was a chain of 2 x 7 exposed round trips)

__device__ __forceinline__ float block_reduce(float v, float* red, int wave, int lane, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < CLS_W; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// The "factored" hand-over of the cls query's contribution to the patch keys (plane output only): per key and head the two scalars
// dS_j and p_j of dk_j += dS_j (s q_cls), dv_j += p_j dO_cls -- the kernel that owns the key forms the rank-1 terms itself.  They live
// in a part of the fp32 dqkv working buffer nothing else uses with plane output: the q-section (columns [0, inner)) of the clip's patch
// rows 1 .., vector (which, h) in R = ceil(N / inner) consecutive rows.  Offsets are relative to the clip's first row.
__device__ __forceinline__ int64_t cls_fact_off(int which, int h, int H, int j, int N, int inner, int ld) {
  const int R = (N + inner - 1) / inner;
  return (int64_t)(1 + (which * H + h) * R + j / inner) * ld + (j % inner);
}

__device__ __forceinline__ float mul_unfused(float a, float b) {
#pragma clang fp contract(off)
  return a * b;             // (a rounded product: the caller's addition must not contract it into an fma)
}

// Lane mapping: 16 lanes per key (lane & 15 = which float4 of the 64-wide head), 16 keys per pass of the block, so every K / V /
// dK / dV row is one coalesced 256-byte access (one key per lane made each lane walk its own 6 KB-strided row: 105 us per launch).
__global__ __launch_bounds__(CLS_W * 64) void attn_cls_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, const uint8_t* __restrict__ mask,
                                                                 int B, int H, int F, int n, float scale, int factored) {
  // factored: the patch keys receive only the two scalars of the rank-1 contribution (cls_fact_off); the patch kernel that owns the key
  // multiplies them with the cls query / cls dO rows itself.  Saves writing and re-reading two 64-float rows per key and head (52 + 52
  // MB per call at B = 32).  The cls key's own row (j = 0) stays full: the patch kernels add to it atomically.
  extern __shared__ __attribute__((aligned(16))) float lds[];   // p[N], dS[N], CLS_W reduction slots, CLS_W x 64 partial dq
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 15, grp = tid >> 4;                     // float4 index inside the head, key slot inside a pass
  constexpr int KPP = CLS_W * 4;                                // keys per pass
  const int bh = blockIdx.x, h = bh % H, b = bh / H;
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  float* pl_ = lds;
  float* ds_ = lds + N;
  float* red = ds_ + N;
  float* part = lds + ((2 * N + CLS_W + 3) & ~3);               // 16-byte aligned (float4 slots)
  const float* base = qkv + (int64_t)b * N * ld + h * DH + sub * 4;
  float* dbase = dqkv + (int64_t)b * N * ld + h * DH + sub * 4;
  float4 q = *reinterpret_cast<const float4*>(base);
  q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
  const float4 dO = *reinterpret_cast<const float4*>(dout + (int64_t)b * N * inner + h * DH + sub * 4);   // row 0
  // CLS_U passes per trip with all their K / V rows requested up front: one block per (b, h) means one wave per SIMD, so a pass that
  // loads, reduces and stores before the next pass's loads are issued is a chain of N / 16 exposed round trips (46 us per launch).
  constexpr int CLS_U = 4;
  for (int j0 = grp; j0 < N; j0 += KPP * CLS_U) {
    float4 kk[CLS_U], vv[CLS_U];
#pragma unroll
    for (int u = 0; u < CLS_U; ++u) {
      const int j = min(j0 + u * KPP, N - 1);
      kk[u] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + inner);
      vv[u] = *reinterpret_cast<const float4*>(base + (int64_t)j * ld + 2 * inner);
    }
#pragma unroll
    for (int u = 0; u < CLS_U; ++u) {
      const int j = j0 + u * KPP;
      float a = fmaf(q.x, kk[u].x, fmaf(q.y, kk[u].y, fmaf(q.z, kk[u].z, q.w * kk[u].w)));
      float dp = fmaf(dO.x, vv[u].x, fmaf(dO.y, vv[u].y, fmaf(dO.z, vv[u].z, dO.w * vv[u].w)));
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o); dp += __shfl_xor(dp, o); }
      if (sub == 0 && j < N) {
        if (j > 0 && mask && !mask[b * F + (j - 1) / n]) a = -FLT_MAX;
        pl_[j] = a; ds_[j] = dp;
      }
    }
  }
  __syncthreads();
  float mx = -FLT_MAX;
  for (int j = tid; j < N; j += CLS_W * 64) mx = fmaxf(mx, pl_[j]);
  mx = block_reduce(mx, red, wave, lane, true);
  float sum = 0.f;
  for (int j = tid; j < N; j += CLS_W * 64) { const float e = expf(pl_[j] - mx); pl_[j] = e; sum += e; }
  sum = block_reduce(sum, red, wave, lane, false);
  const float inv = 1.0f / sum;
  float delta = 0.f;
  for (int j = tid; j < N; j += CLS_W * 64) { const float p = pl_[j] * inv; pl_[j] = p; delta += p * ds_[j]; }
  delta = block_reduce(delta, red, wave, lane, false);
  __syncthreads();
  // per key: dS; dk_j = dS * q_scaled, dv_j = p * dO (stores), dq += dS * k_j
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = grp; j0 < N; j0 += KPP * CLS_U) {
    float4 kk[CLS_U];
#pragma unroll
    for (int u = 0; u < CLS_U; ++u)
      kk[u] = *reinterpret_cast<const float4*>(base + (int64_t)min(j0 + u * KPP, N - 1) * ld + inner);
#pragma unroll
    for (int u = 0; u < CLS_U; ++u) {
      const int j = j0 + u * KPP;
      if (j < N) {
        const float p = pl_[j];
        const float dS = p * (ds_[j] - delta);
        if (factored && j > 0) {
          if (sub == 0) {
            float* dclip = dqkv + (int64_t)b * N * ld;
            dclip[cls_fact_off(0, h, H, j, N, inner, ld)] = dS;
            dclip[cls_fact_off(1, h, H, j, N, inner, ld)] = p;
          }
        } else {
          *reinterpret_cast<float4*>(dbase + (int64_t)j * ld + inner) = make_float4(dS * q.x, dS * q.y, dS * q.z, dS * q.w);
          *reinterpret_cast<float4*>(dbase + (int64_t)j * ld + 2 * inner) = make_float4(p * dO.x, p * dO.y, p * dO.z, p * dO.w);
        }
        dq.x = fmaf(dS, kk[u].x, dq.x); dq.y = fmaf(dS, kk[u].y, dq.y); dq.z = fmaf(dS, kk[u].z, dq.z); dq.w = fmaf(dS, kk[u].w, dq.w);
      }
    }
  }
  // sum over the key slots: the 4 groups of a wavefront, then the wavefronts
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    dq.x += __shfl_xor(dq.x, o); dq.y += __shfl_xor(dq.y, o); dq.z += __shfl_xor(dq.z, o); dq.w += __shfl_xor(dq.w, o);
  }
  if (lane < 16) *reinterpret_cast<float4*>(part + wave * 64 + sub * 4) = dq;
  __syncthreads();
  if (tid < 16) {
    float4 t = *reinterpret_cast<const float4*>(part + sub * 4);
#pragma unroll
    for (int w = 1; w < CLS_W; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(part + w * 64 + sub * 4);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(dbase) = make_float4(scale * t.x, scale * t.y, scale * t.z, scale * t.w);
  }
}

// ---------------------------------------------------------------------------------------- space attention backward on the matrix cores
// One wavefront per (b, h, frame), same transposed formulation as attn_space_fwd_mfma_kernel (tsf_fwd.hip): every product is
// arranged so that its B operand is an accumulator of the previous product used in place, and its A operand is read from global
// memory either by rows (128 contiguous bytes per lane) or by columns (coalesced across lanes).  No LDS except 512 B of
// per-query softmax statistics handed from phase A to phase B.
//   phase A, lane = query:  S^T = K (sQ)^T, dP^T = V dO^T  ->  P^T, delta, dS^T (all in-lane)  ->  dQ^T = K^T dS^T
//   phase B, lane = key  :  S = (sQ) K^T,  dP = dO V^T     ->  P, dS (statistics from phase A)  ->  dV^T = dO^T P,  dK^T = (sQ)^T dS
// 896 v_mfma_f32_32x32x2_f32 per group.  dq is stored; dk / dv are added onto the cls query's contribution written by
// attn_cls_bwd_kernel (read-modify-write: a patch key belongs to exactly one group; fp32 atomics for the shared cls key).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mfma_slot_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__device__ __forceinline__ void load_row32(const float* __restrict__ p, float (&dst)[32], float mul) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * t);
    dst[4 * t] = v.x * mul; dst[4 * t + 1] = v.y * mul; dst[4 * t + 2] = v.z * mul; dst[4 * t + 3] = v.w * mul;
  }
}

// XP (round 6, default): phase B does not recompute P and dS (256 of the 896 MFMAs, plus the row loads of Q and dO) -- phase A leaves
// P^T and dS^T in LDS as [key][query] with a 65-float row stride (33 KB per wavefront, dynamic LDS: the block's four wavefronts take
// 133 KB, one block per CU as before), written along rows and read back along columns, both conflict-free.  640 MFMAs per group.
constexpr int XP_LD = 65;
template <int WPB, bool FACT = false, bool XP = false>
__global__ __launch_bounds__(WPB * 64) void attn_space_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                      float* __restrict__ dqkv, int B, int H, int F, int n,
                                                                      float scale, const PlaneRef dp, const DetLog det) {
  __shared__ float2 stat_all[XP ? 1 : WPB][64];             // (logsumexp, delta) per query of the wavefront's group
  extern __shared__ __attribute__((aligned(16))) float xp_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane & 31, hf = lane >> 5;
  float2* stat = stat_all[XP ? 0 : wave];
  float* Pm = xp_lds + (XP ? wave * 2 * 64 * XP_LD : 0);    // P^T [key slot][query slot]
  float* Sm = Pm + 64 * XP_LD;                              // dS^T
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * F) return;                    // wave-uniform (no block-level barrier below)
  const int f = (int)(wid % F);
  const int bh = (int)(wid / F);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  float* dbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float* dobase = dout + (int64_t)b * N * inner + h * DH;
  const int t0 = 1 + f * n;
  auto tok_k = [&](int key) { return key == 0 ? 0 : t0 + min(key, n) - 1; };
  auto tok_q = [&](int q) { return t0 + min(q, n - 1); };

  // ---------------------------------------------------------------- phase A: lane = query (32 j + c)
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const int q = 32 * j + c;
    float qb[32], dob[32];
    load_row32(base + (int64_t)tok_q(q) * ld + hf * 32, qb, scale);
    load_row32(dobase + (int64_t)tok_q(q) * inner + hf * 32, dob, 1.0f);
    f32x16 st[2], dpt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float ka[32], va[32];
      load_row32(base + (int64_t)tok_k(32 * i + c) * ld + inner + hf * 32, ka, 1.0f);
      load_row32(base + (int64_t)tok_k(32 * i + c) * ld + 2 * inner + hf * 32, va, 1.0f);
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[i][r] = 0.f; dpt[i][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        st[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[ks], qb[ks], st[i], 0, 0, 0);
        dpt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[ks], dob[ks], dpt[i], 0, 0, 0);
      }
    }
    // K by columns in the order the dS^T accumulators are consumed; in flight during the softmax
    float kc[2][32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const float* kr = base + (int64_t)tok_k(32 * (ks >> 4) + mfma_slot_row(ks & 15, hf)) * ld + inner;
      kc[0][ks] = kr[c];
      kc[1][ks] = kr[32 + c];
    }
    float mx = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * i + mfma_slot_row(r, hf) <= n) mx = fmaxf(mx, st[i][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = (32 * i + mfma_slot_row(r, hf) <= n) ? __expf(st[i][r] - mx) : 0.f;
        st[i][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    float delta = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[i][r] *= inv; delta = fmaf(st[i][r], dpt[i][r], delta); }
    delta += __shfl_xor(delta, 32);
    if constexpr (XP) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) Pm[(32 * i + mfma_slot_row(r, hf)) * XP_LD + q] = st[i][r];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[i][r] *= dpt[i][r] - delta;          // dS^T (zero on padded keys: P^T is zero there)
    if constexpr (XP) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) Sm[(32 * i + mfma_slot_row(r, hf)) * XP_LD + q] = st[i][r];
    } else if (hf == 0) stat[q] = make_float2(mx + __logf(sum), delta);
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) dq[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[i][ks], st[ks >> 4][ks & 15], dq[i], 0, 0, 0);
    if (q < n) {
      float* drow = dbase + (int64_t)(t0 + q) * ld;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = make_float4(scale * dq[i][4 * g], scale * dq[i][4 * g + 1], scale * dq[i][4 * g + 2], scale * dq[i][4 * g + 3]);
          if (dp.p) planes_store4(dp, b * N + t0 + q, h * DH + 32 * i + 8 * g + 4 * hf, v.x, v.y, v.z, v.w);   // final: nothing else writes a patch row's dq
          else *reinterpret_cast<float4*>(drow + 32 * i + 8 * g + 4 * hf) = v;
        }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the statistics are in LDS
  __builtin_amdgcn_wave_barrier();

  // ---------------------------------------------------------------- phase B: lane = key (32 j + c)
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const int key = 32 * j + c;
    f32x16 pp[2], ds[2];                      // P and dS tiles [query tile]
    if constexpr (!XP) {
      float kb[32], vb[32];
      load_row32(base + (int64_t)tok_k(key) * ld + inner + hf * 32, kb, 1.0f);
      load_row32(base + (int64_t)tok_k(key) * ld + 2 * inner + hf * 32, vb, 1.0f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float qa[32], da[32];
        load_row32(base + (int64_t)tok_q(32 * i + c) * ld + hf * 32, qa, scale);
        load_row32(dobase + (int64_t)tok_q(32 * i + c) * inner + hf * 32, da, 1.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) { pp[i][r] = 0.f; ds[i][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
          pp[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[ks], kb[ks], pp[i], 0, 0, 0);
          ds[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[ks], vb[ks], ds[i], 0, 0, 0);
        }
      }
    }
    // dO and scaled Q by columns, in the query order the P / dS accumulators are consumed in
    float dc[2][32], qc[2][32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const int tq = tok_q(32 * (ks >> 4) + mfma_slot_row(ks & 15, hf));
      const float* dr = dobase + (int64_t)tq * inner;
      const float* qr = base + (int64_t)tq * ld;
      dc[0][ks] = dr[c]; dc[1][ks] = dr[32 + c];
      qc[0][ks] = qr[c] * scale; qc[1][ks] = qr[32 + c] * scale;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = 32 * i + mfma_slot_row(r, hf);
        if constexpr (XP) {                   // phase A's P^T / dS^T, read along a column (rows of padded keys hold zeros; padded
          const bool ok = q < n;              // query slots hold copies of the last query: masked here)
          pp[i][r] = ok ? Pm[key * XP_LD + q] : 0.f;
          ds[i][r] = ok ? Sm[key * XP_LD + q] : 0.f;
        } else {
          const float2 sd = stat[q];
          const float p = (q < n && key <= n) ? __expf(pp[i][r] - sd.x) : 0.f;
          ds[i][r] = p * (ds[i][r] - sd.y);
          pp[i][r] = p;
        }
      }
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv[i][r] = 0.f; dk[i][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 32; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[i][ks], pp[ks >> 4][ks & 15], dv[i], 0, 0, 0);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[i][ks], ds[ks >> 4][ks & 15], dk[i], 0, 0, 0);
      }
    if (key <= n) {
      float* krow = dbase + (int64_t)tok_k(key) * ld + inner;
      float* vrow = dbase + (int64_t)tok_k(key) * ld + 2 * inner;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = 32 * i + 8 * g + 4 * hf;
          if (key == 0) {                       // the cls key is shared by every frame of the clip
            if (det.vals) {                     // deterministic mode: one log row per frame, summed in frame order (det.hpp)
#pragma unroll
              for (int e = 0; e < 4; ++e) { det_put(det, 2 * bh, f, d0 + e, dk[i][4 * g + e]); det_put(det, 2 * bh + 1, f, d0 + e, dv[i][4 * g + e]); }
              if (f == 0 && d0 == 0) {
                det_base(det, 2 * bh, (int64_t)b * N * ld + h * DH + inner);
                det_base(det, 2 * bh + 1, (int64_t)b * N * ld + h * DH + 2 * inner);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) { atomicAdd(krow + d0 + e, dk[i][4 * g + e]); atomicAdd(vrow + d0 + e, dv[i][4 * g + e]); }
            }
          } else {
            float4 a, v;
            if constexpr (FACT) {               // the cls query's rank-1 contribution from its two scalars (attn_cls_bwd_kernel, factored)
              // (the same two roundings as the full rows the non-factored kernel stores: dS * (q * scale), p * dO; no contraction)
              const float* dclip = dqkv + (int64_t)b * N * ld;
              const float dsc = dclip[cls_fact_off(0, h, H, tok_k(key), N, inner, ld)];
              const float pc = dclip[cls_fact_off(1, h, H, tok_k(key), N, inner, ld)];
              const float4 qc = *reinterpret_cast<const float4*>(base + d0), dc = *reinterpret_cast<const float4*>(dobase + d0);
              a = make_float4(mul_unfused(dsc, mul_unfused(qc.x, scale)), mul_unfused(dsc, mul_unfused(qc.y, scale)),
                              mul_unfused(dsc, mul_unfused(qc.z, scale)), mul_unfused(dsc, mul_unfused(qc.w, scale)));
              v = make_float4(mul_unfused(pc, dc.x), mul_unfused(pc, dc.y), mul_unfused(pc, dc.z), mul_unfused(pc, dc.w));
            } else {
              a = *reinterpret_cast<const float4*>(krow + d0); v = *reinterpret_cast<const float4*>(vrow + d0);
            }
            a.x += dk[i][4 * g]; a.y += dk[i][4 * g + 1]; a.z += dk[i][4 * g + 2]; a.w += dk[i][4 * g + 3];
            v.x += dv[i][4 * g]; v.y += dv[i][4 * g + 1]; v.z += dv[i][4 * g + 2]; v.w += dv[i][4 * g + 3];
            if (dp.p) {                           // the group owns its patch keys: these are the final dk / dv values
              const int row = b * N + tok_k(key);
              planes_store4(dp, row, inner + h * DH + d0, a.x, a.y, a.z, a.w);
              planes_store4(dp, row, 2 * inner + h * DH + d0, v.x, v.y, v.z, v.w);
            } else {
              *reinterpret_cast<float4*>(krow + d0) = a;
              *reinterpret_cast<float4*>(vrow + d0) = v;
            }
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------- attention backward: patch queries
// Same decomposition as the forward kernel (one lane per query; MODE 0 time / 1 space).
// pass 1 (lane = query i): s_ij, p_ij, dP_ij = dO_i.v_j, delta_i, dS_ij = p_ij(dP_ij - delta_i), dq_i = scale*sum_j dS_ij k_j
// pass 2 (lane = key  j): dk_j = sum_i dS_ij (scale q_i),  dv_j = sum_i p_ij dO_i   -- roles transposed through LDS
// Per-wavefront LDS: tile A [ROWS][STRIDE] (K, later Q.scale), tile B [ROWS][STRIDE] (V, later dO), P and dS [64][SP].
template <int MODE, int NKEYS, int PPW, int STRIDE, int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_patch_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, const uint8_t* __restrict__ mask,
                                                                 const uint8_t* __restrict__ ident, int B, int H, int F, int n,
                                                                 float scale, const PlaneRef dp, const DetLog det) {
  constexpr int ROWS = MODE == 0 ? 1 + PPW * (NKEYS - 1) : NKEYS;
  constexpr int QROWS = 64;                                  // pass-2 tiles hold one row per query lane
  constexpr int TROWS = ROWS > QROWS ? ROWS : QROWS;
  constexpr int SP = (NKEYS % 2 == 0) ? NKEYS + 1 : NKEYS;   // odd stride: conflict-free writer lanes
  constexpr int WAVE_LDS = 2 * TROWS * STRIDE + 2 * 64 * SP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* tA = lds + wave * WAVE_LDS;
  float* tB = tA + TROWS * STRIDE;
  float* Pm = tB + TROWS * STRIDE;
  float* Sm = Pm + 64 * SP;

  const int N = 1 + F * n;
  const int inner = H * DH;
  const int ld = 3 * inner;
  const int chunks = MODE == 0 ? (n + PPW - 1) / PPW : F;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * chunks) return;
  const int c = (int)(wid % chunks);
  const int bh = (int)(wid / chunks);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  float* dbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float* dob = dout + (int64_t)b * N * inner + h * DH;

  int qtok = -1, pl = 0, fq = 0;
  if (MODE == 0) {
    pl = lane / (NKEYS - 1); fq = lane % (NKEYS - 1);
    const int p = c * PPW + pl;
    if (pl < PPW && p < n) qtok = 1 + fq * n + p;
    if (pl >= PPW) pl = 0;
  } else {
    if (lane < n) qtok = 1 + c * n + lane;
  }
  const int row1 = MODE == 0 ? 1 + pl * (NKEYS - 1) : 1;

  auto row_token = [&](int r) -> int {
    if (r == 0) return 0;
    if (MODE == 0) {
      const int pl2 = (r - 1) / (NKEYS - 1), f2 = (r - 1) % (NKEYS - 1);
      const int p = c * PPW + pl2;
      return p < n ? 1 + f2 * n + p : -1;
    }
    return 1 + c * n + (r - 1);
  };
  auto stage = [&](float* tile, int which) {
    for (int r0 = 0; r0 < ROWS; r0 += 4) {
      const int r = r0 + (lane >> 4), c4 = lane & 15;
      if (r < ROWS) {
        const int tok = row_token(r);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok >= 0) v = *reinterpret_cast<const float4*>(base + (int64_t)tok * ld + which * inner + c4 * 4);
        *reinterpret_cast<float4*>(tile + r * STRIDE + c4 * 4) = v;
      }
    }
  };

  stage(tA, 1);
  stage(tB, 2);
  float q[DH];
  {
    const float* qp = base + (int64_t)(qtok >= 0 ? qtok : 0) * ld;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(qp + i * 4);
      q[4 * i] = v.x * scale; q[4 * i + 1] = v.y * scale; q[4 * i + 2] = v.z * scale; q[4 * i + 3] = v.w * scale;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- pass 1a: scores -> Pm[lane][j]
  float* myP = Pm + lane * SP;
  float* myS = Sm + lane * SP;
  float mx = -FLT_MAX;
#pragma unroll 2
  for (int j = 0; j < NKEYS; ++j) {
    const float* kr = tA + (j == 0 ? 0 : row1 + j - 1) * STRIDE;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) {
      const float4 k0 = *reinterpret_cast<const float4*>(kr + i * 8);
      const float4 k1 = *reinterpret_cast<const float4*>(kr + i * 8 + 4);
      a0 = fmaf(q[8 * i], k0.x, a0); a0 = fmaf(q[8 * i + 1], k0.y, a0);
      a0 = fmaf(q[8 * i + 2], k0.z, a0); a0 = fmaf(q[8 * i + 3], k0.w, a0);
      a1 = fmaf(q[8 * i + 4], k1.x, a1); a1 = fmaf(q[8 * i + 5], k1.y, a1);
      a1 = fmaf(q[8 * i + 6], k1.z, a1); a1 = fmaf(q[8 * i + 7], k1.w, a1);
    }
    float a = a0 + a1;
    if (MODE == 0 && j > 0) {
      const bool ok = mask[b * F + (j - 1)] && ident[(b * F + fq) * F + (j - 1)];
      if (!ok) a = -FLT_MAX;
    }
    myP[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
  for (int j = 0; j < NKEYS; ++j) { const float e = expf(myP[j] - mx); myP[j] = e; sum += e; }
  const float inv = qtok >= 0 ? 1.0f / sum : 0.f;            // idle lanes contribute nothing in pass 2

  // ---- pass 1b: dP_j = dO . v_j ; delta
  float dO[DH];
  {
    const float* dp = dob + (int64_t)(qtok >= 0 ? qtok : 0) * inner;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(dp + i * 4);
      dO[4 * i] = v.x; dO[4 * i + 1] = v.y; dO[4 * i + 2] = v.z; dO[4 * i + 3] = v.w;
    }
  }
  float delta = 0.f;
#pragma unroll 2
  for (int j = 0; j < NKEYS; ++j) {
    const float* vr = tB + (j == 0 ? 0 : row1 + j - 1) * STRIDE;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < DH / 8; ++i) {
      const float4 v0 = *reinterpret_cast<const float4*>(vr + i * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(vr + i * 8 + 4);
      a0 = fmaf(dO[8 * i], v0.x, a0); a0 = fmaf(dO[8 * i + 1], v0.y, a0);
      a0 = fmaf(dO[8 * i + 2], v0.z, a0); a0 = fmaf(dO[8 * i + 3], v0.w, a0);
      a1 = fmaf(dO[8 * i + 4], v1.x, a1); a1 = fmaf(dO[8 * i + 5], v1.y, a1);
      a1 = fmaf(dO[8 * i + 6], v1.z, a1); a1 = fmaf(dO[8 * i + 7], v1.w, a1);
    }
    const float dp = a0 + a1;
    const float p = myP[j] * inv;
    myP[j] = p;
    myS[j] = dp;
    delta += p * dp;
  }
  // ---- pass 1c: dS, dq
  float dq[DH];
#pragma unroll
  for (int i = 0; i < DH; ++i) dq[i] = 0.f;
#pragma unroll 2
  for (int j = 0; j < NKEYS; ++j) {
    const float* kr = tA + (j == 0 ? 0 : row1 + j - 1) * STRIDE;
    const float dS = myP[j] * (myS[j] - delta);
    myS[j] = dS;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 kk = *reinterpret_cast<const float4*>(kr + i * 4);
      dq[4 * i] = fmaf(dS, kk.x, dq[4 * i]); dq[4 * i + 1] = fmaf(dS, kk.y, dq[4 * i + 1]);
      dq[4 * i + 2] = fmaf(dS, kk.z, dq[4 * i + 2]); dq[4 * i + 3] = fmaf(dS, kk.w, dq[4 * i + 3]);
    }
  }
  if (qtok >= 0) {
    float* dqr = dbase + (int64_t)qtok * ld;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      if (dp.p) planes_store4(dp, b * N + qtok, h * DH + i * 4, dq[4 * i] * scale, dq[4 * i + 1] * scale, dq[4 * i + 2] * scale, dq[4 * i + 3] * scale);
      else *reinterpret_cast<float4*>(dqr + i * 4) =
          make_float4(dq[4 * i] * scale, dq[4 * i + 1] * scale, dq[4 * i + 2] * scale, dq[4 * i + 3] * scale);
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- pass 2 set-up: tA[lane] = scaled q_lane, tB[lane] = dO_lane (rows indexed by QUERY lane now)
  {
    float* qa = tA + lane * STRIDE;
    float* da = tB + lane * STRIDE;
    const float live = qtok >= 0 ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      *reinterpret_cast<float4*>(qa + i * 4) = make_float4(q[4 * i] * live, q[4 * i + 1] * live, q[4 * i + 2] * live, q[4 * i + 3] * live);
      *reinterpret_cast<float4*>(da + i * 4) = make_float4(dO[4 * i] * live, dO[4 * i + 1] * live, dO[4 * i + 2] * live, dO[4 * i + 3] * live);
    }
    if (qtok < 0) { for (int j = 0; j < NKEYS; ++j) { myP[j] = 0.f; myS[j] = 0.f; } }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- pass 2a: patch keys.  lane = key token (same lane->token map as the queries)
  {
    float dk[DH], dv[DH];
#pragma unroll
    for (int i = 0; i < DH; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    // queries that see my token as a key: time -> the F lanes of my patch group; space -> all n lanes
    const int i0 = MODE == 0 ? pl * (NKEYS - 1) : 0;
    const int cnt = MODE == 0 ? (NKEYS - 1) : n;
    const int jme = MODE == 0 ? 1 + fq : 1 + lane;           // my key index inside those queries' score rows
#pragma unroll 2
    for (int t = 0; t < cnt; ++t) {
      const int iq = i0 + t;
      const float dS = Sm[iq * SP + jme];
      const float p = Pm[iq * SP + jme];
      const float* qr = tA + iq * STRIDE;
      const float* dr = tB + iq * STRIDE;
#pragma unroll
      for (int i = 0; i < DH / 4; ++i) {
        const float4 qq = *reinterpret_cast<const float4*>(qr + i * 4);
        const float4 dd = *reinterpret_cast<const float4*>(dr + i * 4);
        dk[4 * i] = fmaf(dS, qq.x, dk[4 * i]); dk[4 * i + 1] = fmaf(dS, qq.y, dk[4 * i + 1]);
        dk[4 * i + 2] = fmaf(dS, qq.z, dk[4 * i + 2]); dk[4 * i + 3] = fmaf(dS, qq.w, dk[4 * i + 3]);
        dv[4 * i] = fmaf(p, dd.x, dv[4 * i]); dv[4 * i + 1] = fmaf(p, dd.y, dv[4 * i + 1]);
        dv[4 * i + 2] = fmaf(p, dd.z, dv[4 * i + 2]); dv[4 * i + 3] = fmaf(p, dd.w, dv[4 * i + 3]);
      }
    }
    if (qtok >= 0) {   // exclusive owner of this token's k/v rows for this head: accumulate on the cls kernel's values
      float* dkr = dbase + (int64_t)qtok * ld + inner;
      float* dvr = dbase + (int64_t)qtok * ld + 2 * inner;
#pragma unroll
      for (int i = 0; i < DH / 4; ++i) {
        float4 a = *reinterpret_cast<float4*>(dkr + i * 4);
        a.x += dk[4 * i]; a.y += dk[4 * i + 1]; a.z += dk[4 * i + 2]; a.w += dk[4 * i + 3];
        float4 v = *reinterpret_cast<float4*>(dvr + i * 4);
        v.x += dv[4 * i]; v.y += dv[4 * i + 1]; v.z += dv[4 * i + 2]; v.w += dv[4 * i + 3];
        if (dp.p) {
          planes_store4(dp, b * N + qtok, inner + h * DH + i * 4, a.x, a.y, a.z, a.w);
          planes_store4(dp, b * N + qtok, 2 * inner + h * DH + i * 4, v.x, v.y, v.z, v.w);
        } else {
          *reinterpret_cast<float4*>(dkr + i * 4) = a;
          *reinterpret_cast<float4*>(dvr + i * 4) = v;
        }
      }
    }
  }
  // ---- pass 2b: the cls key (j = 0) is shared by every group of this head: lane = d, atomics into row 0
  {
    float ak = 0.f, av = 0.f;
    for (int iq = 0; iq < 64; ++iq) {
      ak = fmaf(Sm[iq * SP], tA[iq * STRIDE + lane], ak);
      av = fmaf(Pm[iq * SP], tB[iq * STRIDE + lane], av);
    }
    if (det.vals) {                             // deterministic mode: one log row per chunk of this (b, h), summed in chunk order
      det_put(det, 2 * bh, c, lane, ak);
      det_put(det, 2 * bh + 1, c, lane, av);
      if (c == 0 && lane == 0) {
        det_base(det, 2 * bh, (int64_t)b * N * ld + h * DH + inner);
        det_base(det, 2 * bh + 1, (int64_t)b * N * ld + h * DH + 2 * inner);
      }
    } else {
      atomicAdd(dbase + inner + lane, ak);
      atomicAdd(dbase + 2 * inner + lane, av);
    }
  }
}

// ---------------------------------------------------------------------------------------- time attention backward, one patch per wavefront
// One wavefront per (b, h, patch): F queries (the patch in every frame) x (cls + F) keys.  Global traffic is by whole 256-byte head rows
// with lane = d (4F + 2 coalesced loads up front, kept in registers for the rank-1 updates at the end); the rows are mirrored in
// 9 KB of LDS (F = 8) for the score phase, where lane = (query f, key j): two 64-long dot products per lane, the softmax and delta
// across the F lanes of a query by shuffles.  dq / dk / dv are formed with lane = d from the registers and the F x (F+1) P / dS
// matrices in LDS (broadcast reads), parked in the row tiles and written out eight columns per lane (plane blocks or fp32).
// Against attn_patch_bwd_kernel<0,...> (7 patches per wavefront, 39 KB of LDS each, 4 wavefronts per CU): 16+ wavefronts per CU.
template <int F, int WPB, bool FACT>
__global__ __launch_bounds__(WPB * 64, F == 8 ? 4 : 2) void attn_time_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, const uint8_t* __restrict__ mask,
                                                                const uint8_t* __restrict__ ident, int B, int H, int n, float scale,
                                                                const PlaneRef dp, const DetLog det) {
  constexpr int NK = F + 1, ST = 68, SPP = (NK + 3) & ~3;
  constexpr int WAVE_LDS = (4 * F + 2) * ST + 2 * F * SPP;
  constexpr int LPF = 64 / F;                                // lanes per query in the cls-column pass (F dims each)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* qT = lds + wave * WAVE_LDS;                         // [F][ST] scaled q, later dq
  float* dT = qT + F * ST;                                   // [F][ST] dO
  float* kT = dT + F * ST;                                   // [NK][ST] k (row 0 = cls), later dk
  float* vT = kT + NK * ST;                                  // [NK][ST] v, later dv
  float* Pm = vT + NK * ST;                                  // [F][SPP] P   (column 0 = cls key)
  float* Sm = Pm + F * SPP;                                  // [F][SPP] dS
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * n) return;                     // wave-uniform; no block-level barrier below
  const int p = (int)(wid % n);
  const int bh = (int)(wid / n);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  float* dbase = dqkv + (int64_t)b * N * ld + h * DH;
  const float* dob = dout + (int64_t)b * N * inner + h * DH;

  float q[F], dO[F], k[NK], v[NK];
  k[0] = base[inner + lane];
  v[0] = base[2 * inner + lane];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const int64_t tok = 1 + f * n + p;
    q[f] = base[tok * ld + lane];
    k[f + 1] = base[tok * ld + inner + lane];
    v[f + 1] = base[tok * ld + 2 * inner + lane];
    dO[f] = dob[tok * inner + lane];
  }
#pragma unroll
  for (int f = 0; f < F; ++f) {
    q[f] *= scale;
    qT[f * ST + lane] = q[f];
    dT[f * ST + lane] = dO[f];
  }
#pragma unroll
  for (int j = 0; j < NK; ++j) { kT[j * ST + lane] = k[j]; vT[j * ST + lane] = v[j]; }
  __builtin_amdgcn_wave_barrier();

  // ---- the cls key's column: s_f0 = q_f . k_0, dP_f0 = dO_f . v_0 (LPF lanes per query)
  {
    const int f = lane / LPF, c = lane % LPF;
    const float* qr = qT + f * ST + c * F;
    const float* dr = dT + f * ST + c * F;
    const float* kr = kT + c * F;
    const float* vr = vT + c * F;
    float a = 0.f, d2 = 0.f;
#pragma unroll
    for (int i = 0; i < F; i += 4) {
      const float4 qq = *reinterpret_cast<const float4*>(qr + i), kk = *reinterpret_cast<const float4*>(kr + i);
      const float4 dd = *reinterpret_cast<const float4*>(dr + i), vv = *reinterpret_cast<const float4*>(vr + i);
      a = fmaf(qq.x, kk.x, a); a = fmaf(qq.y, kk.y, a); a = fmaf(qq.z, kk.z, a); a = fmaf(qq.w, kk.w, a);
      d2 = fmaf(dd.x, vv.x, d2); d2 = fmaf(dd.y, vv.y, d2); d2 = fmaf(dd.z, vv.z, d2); d2 = fmaf(dd.w, vv.w, d2);
    }
#pragma unroll
    for (int o = 1; o < LPF; o <<= 1) { a += __shfl_xor(a, o); d2 += __shfl_xor(d2, o); }
    if (c == 0) { Pm[f * SPP] = a; Sm[f * SPP] = d2; }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- scores, softmax, delta, dS: lane = (f, jj), key j = jj + 1 (frame jj)
#pragma unroll 1
  for (int pass = 0; pass < F * F / 64; ++pass) {
    const int pi = pass * 64 + lane;
    const int f = pi / F, jj = pi % F, j = jj + 1;
    const float* qr = qT + f * ST;
    const float* dr = dT + f * ST;
    const float* kr = kT + j * ST;
    const float* vr = vT + j * ST;
    float s_a = 0.f, s_b = 0.f, d_a = 0.f, d_b = 0.f;
#pragma unroll 2
    for (int i = 0; i < DH / 8; ++i) {
      const float4 q0 = *reinterpret_cast<const float4*>(qr + 8 * i), q1 = *reinterpret_cast<const float4*>(qr + 8 * i + 4);
      const float4 k0 = *reinterpret_cast<const float4*>(kr + 8 * i), k1 = *reinterpret_cast<const float4*>(kr + 8 * i + 4);
      s_a = fmaf(q0.x, k0.x, s_a); s_a = fmaf(q0.y, k0.y, s_a); s_a = fmaf(q0.z, k0.z, s_a); s_a = fmaf(q0.w, k0.w, s_a);
      s_b = fmaf(q1.x, k1.x, s_b); s_b = fmaf(q1.y, k1.y, s_b); s_b = fmaf(q1.z, k1.z, s_b); s_b = fmaf(q1.w, k1.w, s_b);
      const float4 o0 = *reinterpret_cast<const float4*>(dr + 8 * i), o1 = *reinterpret_cast<const float4*>(dr + 8 * i + 4);
      const float4 v0 = *reinterpret_cast<const float4*>(vr + 8 * i), v1 = *reinterpret_cast<const float4*>(vr + 8 * i + 4);
      d_a = fmaf(o0.x, v0.x, d_a); d_a = fmaf(o0.y, v0.y, d_a); d_a = fmaf(o0.z, v0.z, d_a); d_a = fmaf(o0.w, v0.w, d_a);
      d_b = fmaf(o1.x, v1.x, d_b); d_b = fmaf(o1.y, v1.y, d_b); d_b = fmaf(o1.z, v1.z, d_b); d_b = fmaf(o1.w, v1.w, d_b);
    }
    float s = s_a + s_b;
    const float dpv = d_a + d_b;
    if (!(mask[b * F + jj] && ident[(b * F + f) * F + jj])) s = -FLT_MAX;        // masked_fill_(~mask, -finfo.max)
    const float s0 = Pm[f * SPP], dp0 = Sm[f * SPP];
    float mx = fmaxf(s, s0);
#pragma unroll
    for (int o = 1; o < F; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = expf(s - mx), e0 = expf(s0 - mx);
    float sum = e;
#pragma unroll
    for (int o = 1; o < F; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / (sum + e0);
    const float pj = e * inv, p0 = e0 * inv;
    float delta = pj * dpv;
#pragma unroll
    for (int o = 1; o < F; o <<= 1) delta += __shfl_xor(delta, o);
    delta = fmaf(p0, dp0, delta);
    Pm[f * SPP + j] = pj;
    Sm[f * SPP + j] = pj * (dpv - delta);
    if (jj == 0) { Pm[f * SPP] = p0; Sm[f * SPP] = p0 * (dp0 - delta); }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- lane = d: dq_f = scale sum_j dS_fj k_j;  dk_j = sum_f dS_fj (scale q_f);  dv_j = sum_f P_fj dO_f
  // (outer loops rolled: unrolled, the compiler hoists all 2 F (F+1) broadcast reads into registers)
#pragma unroll 2
  for (int f = 0; f < F; ++f) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) a = fmaf(Sm[f * SPP + j], k[j], a);
    qT[f * ST + lane] = a * scale;
  }
#pragma unroll 1
  for (int j = 0; j < NK; ++j) {
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) { ak = fmaf(Sm[f * SPP + j], q[f], ak); av = fmaf(Pm[f * SPP + j], dO[f], av); }
    if (j == 0) {                                            // the cls key is shared by every patch of this (b, h)
      if (det.vals) {                                        // deterministic mode: one log row per patch, summed in patch order
        det_put(det, 2 * bh, p, lane, ak);
        det_put(det, 2 * bh + 1, p, lane, av);
        if (p == 0 && lane == 0) {
          det_base(det, 2 * bh, (int64_t)b * N * ld + h * DH + inner);
          det_base(det, 2 * bh + 1, (int64_t)b * N * ld + h * DH + 2 * inner);
        }
      } else {
        atomicAdd(dbase + inner + lane, ak);
        atomicAdd(dbase + 2 * inner + lane, av);
      }
    } else {
      kT[j * ST + lane] = ak;
      vT[j * ST + lane] = av;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- write-out, eight columns per lane.  The wavefront owns its patch tokens' k / v rows of this head: their fp32 values hold the
  // cls query's contribution (attn_cls_bwd_kernel) and are completed here.
  for (int it = lane; it < 3 * F * 8; it += 64) {
    const int r = it >> 3, seg = it & 7;
    const int kind = r / F, f = r % F;                       // 0 dq, 1 dk, 2 dv
    const int tok = 1 + f * n + p;
    const float* src = (kind == 0 ? qT + f * ST : kind == 1 ? kT + (f + 1) * ST : vT + (f + 1) * ST) + seg * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
    float o[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    float* g = dbase + (int64_t)tok * ld + kind * inner + seg * 8;
    if (kind) {
      if constexpr (FACT) {                 // the cls query's rank-1 contribution from its two scalars (attn_cls_bwd_kernel, factored)
        // (the same two roundings as the full rows the non-factored kernel stores: dS * (q * scale), p * dO; no contraction)
        const float sc = dqkv[(int64_t)b * N * ld + cls_fact_off(kind - 1, h, H, tok, N, inner, ld)];
        const float mul = kind == 1 ? scale : 1.0f;
        const float* rc = (kind == 1 ? base : dob) + seg * 8;          // row 0: the cls query / its dO
        const float4 e0 = *reinterpret_cast<const float4*>(rc), e1 = *reinterpret_cast<const float4*>(rc + 4);
        const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += mul_unfused(sc, mul_unfused(e[i], mul));
      } else {
        const float4 e0 = *reinterpret_cast<const float4*>(g), e1 = *reinterpret_cast<const float4*>(g + 4);
        o[0] += e0.x; o[1] += e0.y; o[2] += e0.z; o[3] += e0.w; o[4] += e1.x; o[5] += e1.y; o[6] += e1.z; o[7] += e1.w;
      }
    }
    if (dp.p) planes_store8(dp, b * N + tok, kind * inner + h * DH + seg * 8, o);
    else {
      *reinterpret_cast<float4*>(g) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(g + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

template <int F, int WPB>
int launch_time_bwd(const float* qkv, const float* dout, float* dqkv, const uint8_t* mask, const uint8_t* ident, int B, int H, int n,
                    float scale, const PlaneRef& dp, bool fact, const DetLog& det, hipStream_t s) {
  constexpr int NK = F + 1, SPP = (NK + 3) & ~3;
  const size_t lds = (size_t)WPB * ((4 * F + 2) * 68 + 2 * F * SPP) * sizeof(float);
  const int64_t waves = (int64_t)B * H * n;
  auto k = fact ? attn_time_bwd_kernel<F, WPB, true> : attn_time_bwd_kernel<F, WPB, false>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_attn_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(WPB * 64), lds, s, qkv, dout, dqkv, mask, ident, B, H, n, scale, dp, det);
  return check_launch("mt_attn_bwd(time)");
}

// plane output of mt_attn_bwd: the cls row of every clip (its dk / dv collect fp32 atomics from every group and are only final when
// the patch kernels have finished) and the zero padding rows
__global__ __launch_bounds__(256) void attn_bwd_cls_planes_kernel(const float* __restrict__ dqkv, int B, int N, int ld, const PlaneRef dp) {
  const int b = blockIdx.x;
  if (b < B) {
    const float* row = dqkv + (int64_t)b * N * ld;
    for (int q = threadIdx.x; q < ld >> 2; q += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + q * 4);
      planes_store4(dp, b * N, q * 4, v.x, v.y, v.z, v.w);
    }
  } else {
    planes_zero_pad(dp, B * N, threadIdx.x, 256);
  }
}

template <int MODE, int NKEYS, int PPW, int STRIDE, int WPB>
int launch_patch_bwd(const float* qkv, const float* dout, float* dqkv, const uint8_t* mask, const uint8_t* ident, int B, int H,
                     int F, int n, float scale, const PlaneRef& dp, const DetLog& det, hipStream_t s) {
  constexpr int ROWS = MODE == 0 ? 1 + PPW * (NKEYS - 1) : NKEYS;
  constexpr int TROWS = ROWS > 64 ? ROWS : 64;
  constexpr int SP = (NKEYS % 2 == 0) ? NKEYS + 1 : NKEYS;
  const int chunks = MODE == 0 ? (n + PPW - 1) / PPW : F;
  const int64_t waves = (int64_t)B * H * chunks;
  const size_t lds = (size_t)WPB * (2 * TROWS * STRIDE + 2 * 64 * SP) * sizeof(float);
  auto k = attn_patch_bwd_kernel<MODE, NKEYS, PPW, STRIDE, WPB>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_attn_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(WPB * 64), lds, s, qkv, dout, dqkv, mask, ident, B, H, F,
                     n, scale, dp, det);
  return check_launch("mt_attn_bwd(patch)");
}

}  // namespace

extern "C" int mt_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                                float* dgamma, float* dbeta, int rows, int dim, int accumulate, float* dx_colsum, int skip_period,
                                const float* dx_in, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dgamma || !dbeta) return fail(MT_ERR_ARG, "mt_layernorm_bwd: null pointer");
  if (dim <= 0 || (dim & 3) || dim > 1024) return fail(MT_ERR_ARG, "mt_layernorm_bwd: dim %d unsupported", dim);
  if (rows <= 0) return 0;
  int blocks = (rows + 3) / 4;
  static const int cap = getenv("MT_LN_BWD_BLOCKS") ? atoi(getenv("MT_LN_BWD_BLOCKS")) : 256;     // tuning knob
  if (blocks > cap) blocks = cap;
  DetScope det((hipStream_t)stream, 3, blocks, dim, true, true);     // deterministic mode: the three column sums by block order
  if (dim <= 512)
    hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, dx, dgamma, dbeta,
                       rows, dim, accumulate, dx_colsum, skip_period, dx_in ? dx_in : dx, det.log);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, dx, dgamma, dbeta,
                       rows, dim, accumulate, dx_colsum, skip_period, dx_in ? dx_in : dx, det.log);
  int rc = check_launch("mt_layernorm_bwd");
  if (!rc) rc = det.reduce_f32(dgamma, 0, 1);
  if (!rc) rc = det.reduce_f32(dbeta, 1, 1);
  if (!rc && dx_colsum) rc = det.reduce_f32(dx_colsum, 2, 1);
  return rc;
}

static int ln_rows_blocks(int rows);

extern "C" int mt_layernorm_bwd_rows(const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                                     const float* dx_in, int rows, int dim, void* dx_planes, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dx_in) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows: null pointer");
  if (dim <= 0 || (dim & 3) || dim > 1024) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows: dim %d unsupported", dim);
  if (dx_planes && ((dim & 15) || ((uintptr_t)dx_planes & 15))) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows: plane output needs dim %% 16 == 0 and 16-byte alignment");
  if (rows <= 0) return 0;
  const int rp = (rows + 31) & ~31;
  const PlaneRef dxp{reinterpret_cast<__bf16*>(dx_planes), (int64_t)rp * dim, dim >> 4, rp};
  const int blocks = ln_rows_blocks(rows);
  static const int keep = getenv("MT_LN_ROWS_KEEP") ? atoi(getenv("MT_LN_ROWS_KEEP")) : 1;
  if (dim <= 512 && keep)
    hipLaunchKernelGGL(layernorm_bwd_rows_kernel2k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, dx, dx_in, rows, dim, dxp);
  else if (dim <= 512)
    hipLaunchKernelGGL(layernorm_bwd_rows_kernel2, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, dx, dx_in, rows, dim, dxp);
  else
    hipLaunchKernelGGL(layernorm_bwd_rows_kernel4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, dx, dx_in, rows, dim, dxp);
  return check_launch("mt_layernorm_bwd_rows");
}

static int ln_rows_blocks(int rows) {
  int blocks = (rows + 3) / 4;
  static const int cap = getenv("MT_LN_ROWS_BLOCKS") ? atoi(getenv("MT_LN_ROWS_BLOCKS")) : 1024;    // tuning knob
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

extern "C" int mt_layernorm_bwd_rows_blocks(int rows) { return ln_rows_blocks(rows); }

extern "C" int mt_layernorm_bwd_rows_sums(const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                                          const float* dx_in, int rows, int dim, void* dx_planes, float* partials, int skip_period,
                                          void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dx_in || !partials) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows_sums: null pointer");
  if (dim <= 0 || (dim & 3) || dim > 512) return fail(MT_ERR_UNSUPPORTED, "mt_layernorm_bwd_rows_sums: dim %d unsupported (<= 512, %% 4 == 0)", dim);
  if (dx_planes && ((dim & 15) || ((uintptr_t)dx_planes & 15))) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows_sums: plane output needs dim %% 16 == 0 and 16-byte alignment");
  if (rows <= 0) return fail(MT_ERR_ARG, "mt_layernorm_bwd_rows_sums: no rows");
  const int rp = (rows + 31) & ~31;
  const PlaneRef dxp{reinterpret_cast<__bf16*>(dx_planes), (int64_t)rp * dim, dim >> 4, rp};
  hipLaunchKernelGGL(layernorm_bwd_rows_sums_kernel, dim3(ln_rows_blocks(rows)), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma,
                     dx, dx_in, rows, dim, dxp, partials, skip_period);
  return check_launch("mt_layernorm_bwd_rows_sums");
}

extern "C" int mt_layernorm_bwd_cols_reduce(const float* partials, int blocks, int dim, float* dgamma, float* dbeta, float* dx_colsum,
                                            void* stream) {
  if (!partials || !dgamma || !dbeta) return fail(MT_ERR_ARG, "mt_layernorm_bwd_cols_reduce: null pointer");
  if (blocks <= 0 || dim <= 0) return fail(MT_ERR_ARG, "mt_layernorm_bwd_cols_reduce: bad shape");
  hipLaunchKernelGGL(layernorm_cols_reduce_kernel, dim3((dim + 63) / 64, 3), dim3(1024), 0, (hipStream_t)stream, partials, blocks, dim,
                     dgamma, dbeta, dx_colsum);
  return check_launch("mt_layernorm_bwd_cols_reduce");
}

extern "C" int mt_layernorm_bwd_cols(const float* dy, const float* x, const float* stats, const float* dx_new, float* dgamma,
                                     float* dbeta, float* dx_colsum, int skip_period, int rows, int dim, void* stream) {
  if (!dy || !x || !stats || !dgamma || !dbeta || (dx_colsum && !dx_new)) return fail(MT_ERR_ARG, "mt_layernorm_bwd_cols: null pointer");
  if (dim <= 0 || (dim & 3) || dim > 1024) return fail(MT_ERR_ARG, "mt_layernorm_bwd_cols: dim %d unsupported", dim);
  if (rows <= 0) return 0;
  int blocks = (rows + 3) / 4;
  if (blocks > 256) blocks = 256;
  DetScope det((hipStream_t)stream, 3, blocks, dim, true, true);     // deterministic mode: the three column sums by block order
  if (dim <= 512)
    hipLaunchKernelGGL(layernorm_bwd_cols_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, dx_new, dgamma, dbeta,
                       dx_colsum, skip_period, rows, dim, det.log);
  else
    hipLaunchKernelGGL(layernorm_bwd_cols_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, stats, dx_new, dgamma, dbeta,
                       dx_colsum, skip_period, rows, dim, det.log);
  int rc = check_launch("mt_layernorm_bwd_cols");
  if (!rc) rc = det.reduce_f32(dgamma, 0, 1);
  if (!rc) rc = det.reduce_f32(dbeta, 1, 1);
  if (!rc && dx_colsum) rc = det.reduce_f32(dx_colsum, 2, 1);
  return rc;
}

extern "C" int mt_colsum(const float* A, int64_t lda, mt_rowmap map, int M, int N, float* out, void* stream) {
  if (!A || !out) return fail(MT_ERR_ARG, "mt_colsum: null pointer");
  const int rpb = 256;
  DetScope det((hipStream_t)stream, (N + 63) / 64, (M + rpb - 1) / rpb, 64);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, A, lda, map.gin,
                     map.gout, map.off, M, N, out, rpb, det.log);
  const int rc = check_launch("mt_colsum");
  return rc ? rc : det.reduce_f32(out);
}

extern "C" int mt_head_bwd(const float* dlogits, const float* x, const float* gamma, const float* beta, const float* w,
                           float* dx, float* dgamma, float* dbeta, float* dw, float* dbias, int B, int N, int dim, int classes,
                           float eps, void* stream) {
  if (!dlogits || !x || !gamma || !beta || !w || !dx || !dgamma || !dbeta || !dw || !dbias)
    return fail(MT_ERR_ARG, "mt_head_bwd: null pointer");
  if (dim > 1024 || (dim & 63)) return fail(MT_ERR_ARG, "mt_head_bwd: dim %d unsupported (multiple of 64, <= 1024)", dim);
  if (classes > dim) return fail(MT_ERR_ARG, "mt_head_bwd: classes > dim");
  if (det_enabled() || B < 2)
    hipLaunchKernelGGL(head_bwd_kernel, dim3(1), dim3(dim), 0, (hipStream_t)stream, dlogits, x, gamma, beta, w, dx, dgamma, dbeta, dw,
                       dbias, B, N, dim, classes, eps);
  else
    hipLaunchKernelGGL(head_bwd_clips_kernel, dim3(B), dim3(dim), 0, (hipStream_t)stream, dlogits, x, gamma, beta, w, dx, dgamma, dbeta,
                       dw, dbias, B, N, dim, classes, eps);
  return check_launch("mt_head_bwd");
}

extern "C" int mt_embed_bwd(const float* dx, float* dcls, float* dpos_emb, float* dsize_emb, const int64_t* positions,
                            const int32_t* sizes, int B, int F, int n, int dim, int pos_rows, int size_rows, void* stream) {
  if (!dx) return fail(MT_ERR_ARG, "mt_embed_bwd: null pointer");
  if (pos_rows <= 0 || (dsize_emb && size_rows <= 0)) return fail(MT_ERR_ARG, "mt_embed_bwd: empty embedding table");
  const int N = 1 + F * n;
  if (det_enabled()) {
    hipStream_t s = (hipStream_t)stream;
    static const bool walk = getenv("MT_DET_EMBED_WALK") != nullptr;        // A/B aid: round 4's run walk for all three gradients
    const int srows = dsize_emb ? size_rows : 1;
    // cls token: B ordered adds per column (kind 0 of the walk kernel); the tables: integer-limb scatter + decode
    hipLaunchKernelGGL(embed_bwd_det_kernel, dim3((dim + 63) / 64, walk ? 3 : 1), dim3(64), 0, s, dx, dcls,
                       dpos_emb, dsize_emb, positions, sizes, B, N, n, F, dim, pos_rows, srows);
    int rc = check_launch("mt_embed_bwd(deterministic)");
    if (rc || walk) return rc;
    for (int kind = 1; kind <= 2; ++kind) {
      float* dst = kind == 1 ? dpos_emb : dsize_emb;
      if (!dst) continue;
      const int rows = kind == 1 ? pos_rows : srows;
      const int64_t total = (int64_t)rows * dim;
      double* acc = reinterpret_cast<double*>(det_arena(s, (size_t)total * 2 * sizeof(double), 1));
      if (!acc) return fail(MT_ERR_LAUNCH, "deterministic mode: no workspace for the embedding gradient (%lld MB)", (long long)(total * 16 >> 20));
      if (hipMemsetAsync(acc, 0, (size_t)total * 2 * sizeof(double), s) != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_embed_bwd: memset");
      hipLaunchKernelGGL(embed_bwd_limb_scatter_kernel, dim3((B * N + 3) / 4), dim3(256), 0, s, dx, positions, sizes, B, N, n, F, dim, rows,
                         kind, acc, total);
      int64_t nb = (total + 255) / 256;
      if (nb > 4096) nb = 4096;
      hipLaunchKernelGGL(embed_bwd_limb_decode_kernel, dim3((unsigned)nb), dim3(256), 0, s, acc, total, dst, total);
      rc = check_launch("mt_embed_bwd(deterministic, limbs)");
      if (rc) return rc;
    }
    return 0;
  }
  static const bool rows_form = getenv("MT_EMBED_BWD_ROWS") != nullptr;        // A/B aid: the row-per-wavefront kernel
  if (rows_form || F * n + 1 != N)
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((B * N + 3) / 4), dim3(256), 0, (hipStream_t)stream, dx, dcls, dpos_emb, dsize_emb,
                       positions, sizes, B, N, n, F, dim, pos_rows, dsize_emb ? size_rows : 1);
  else
    hipLaunchKernelGGL(embed_bwd_slots_kernel, dim3(B * F), dim3(256), 0, (hipStream_t)stream, dx, dcls, dpos_emb, dsize_emb,
                       positions, sizes, B, N, n, F, dim, pos_rows, dsize_emb ? size_rows : 1);
  return check_launch("mt_embed_bwd");
}

extern "C" int mt_attn_bwd(const float* qkv, const float* dout, float* dqkv, const uint8_t* mask, const uint8_t* ident,
                           int B, int H, int F, int n, int mode, float scale, void* dqkv_planes, void* stream) {
  if (!qkv || !dout || !dqkv) return fail(MT_ERR_ARG, "mt_attn_bwd: null pointer");
  if (mode == 0 && (!mask || !ident)) return fail(MT_ERR_ARG, "mt_attn_bwd: time attention needs mask and identities_mask");
  if (n != 49) return fail(MT_ERR_UNSUPPORTED, "mt_attn_bwd: num-patches %d unsupported (49)", n);
  if (dqkv_planes && (mode == 2 || ((uintptr_t)dqkv_planes & 15))) return fail(MT_ERR_ARG, "mt_attn_bwd: plane output needs mode 0 / 1 and 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  const int N = 1 + F * n;
  const int rp = (B * N + 31) & ~31, ld = 3 * H * DH;
  const PlaneRef dp{reinterpret_cast<__bf16*>(dqkv_planes), (int64_t)rp * ld, ld / 16, rp};
  static const bool time_old = getenv("MT_ATTN_TIME_OLD") != nullptr;    // A/B aid: the 7-patches-per-wavefront kernel
  static const bool valu = getenv("MT_ATTN_VALU") != nullptr;            // A/B aid: the one-lane-per-query space kernel
  static const bool fact_on = !getenv("MT_ATTN_CLS_FACT") || atoi(getenv("MT_ATTN_CLS_FACT")) != 0;
  // with plane output the fp32 dqkv is working memory: the cls query's contribution to the patch keys crosses to the patch kernel as two
  // scalars per key and head (kernels that understand it: the MFMA space kernel, the one-patch-per-wavefront time kernel)
  const bool fact = fact_on && dqkv_planes && ((mode == 1 && !valu) || (mode == 0 && !time_old && (F == 8 || F == 16)));
  hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3(B * H), dim3(CLS_W * 64), (2 * N + CLS_W + 4 + CLS_W * 64) * sizeof(float), s, qkv, dout, dqkv, mask, B, H, F, n, scale, fact ? 1 : 0);
  int rc = check_launch("mt_attn_bwd(cls)");
  if (rc || mode == 2) return rc;                 // mode 2: the cls query's adjoint only (dk / dv of every key, dq of the cls row)
  // deterministic mode: the cls key's dk / dv (one row per (b, head), shared by every frame / patch group) go through a log with one
  // rank per group of the launched kernel, and are added to the cls kernel's values in rank order (det.hpp)
  const int det_ranks = mode == 1 ? F : (!time_old && (F == 8 || F == 16) ? n : (F == 8 ? (n + 6) / 7 : (F == 16 ? (n + 3) / 4 : (n + 1) / 2)));
  DetScope det(s, 2 * B * H, det_ranks, DH);
  if (mode == 1) {
    if (valu) rc = launch_patch_bwd<1, 50, 1, 64, 2>(qkv, dout, dqkv, mask, ident, B, H, F, n, scale, dp, det.log, s);
    else {
      const int64_t waves = (int64_t)B * H * F;
      static const bool xp = !(getenv("MT_ATTN_SPACE_XP") && atoi(getenv("MT_ATTN_SPACE_XP")) == 0);    // 0: phase B recomputes P / dS (round 5)
      const dim3 grid((unsigned)((waves + 3) / 4));
      if (xp) {
        constexpr size_t lds = (size_t)4 * 2 * 64 * XP_LD * sizeof(float);
        auto k = fact ? attn_space_bwd_mfma_kernel<4, true, true> : attn_space_bwd_mfma_kernel<4, false, true>;
        if (ensure_dynamic_lds((const void*)k, lds) != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_attn_bwd(space): %zu bytes of LDS refused", lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, qkv, dout, dqkv, B, H, F, n, scale, dp, det.log);
      } else if (fact) hipLaunchKernelGGL((attn_space_bwd_mfma_kernel<4, true, false>), grid, dim3(256), 0, s, qkv, dout, dqkv, B, H, F, n, scale, dp, det.log);
      else hipLaunchKernelGGL((attn_space_bwd_mfma_kernel<4, false, false>), grid, dim3(256), 0, s, qkv, dout, dqkv, B, H, F, n, scale, dp, det.log);
      rc = check_launch("mt_attn_bwd(space, mfma)");
    }
  } else if (!time_old) {
    switch (F) {
      case 8: rc = launch_time_bwd<8, 4>(qkv, dout, dqkv, mask, ident, B, H, n, scale, dp, fact, det.log, s); break;
      case 16: rc = launch_time_bwd<16, 2>(qkv, dout, dqkv, mask, ident, B, H, n, scale, dp, fact, det.log, s); break;
      case 32: rc = launch_patch_bwd<0, 33, 2, 68, 2>(qkv, dout, dqkv, mask, ident, B, H, F, n, scale, dp, det.log, s); break;   // (130 row registers: spills)
      default: return fail(MT_ERR_UNSUPPORTED, "mt_attn_bwd: num-frames %d unsupported (8/16/32)", F);
    }
  } else {
    switch (F) {
      case 8: rc = launch_patch_bwd<0, 9, 7, 68, 2>(qkv, dout, dqkv, mask, ident, B, H, F, n, scale, dp, det.log, s); break;
      case 16: rc = launch_patch_bwd<0, 17, 4, 68, 2>(qkv, dout, dqkv, mask, ident, B, H, F, n, scale, dp, det.log, s); break;
      case 32: rc = launch_patch_bwd<0, 33, 2, 68, 2>(qkv, dout, dqkv, mask, ident, B, H, F, n, scale, dp, det.log, s); break;
      default: return fail(MT_ERR_UNSUPPORTED, "mt_attn_bwd: num-frames %d unsupported (8/16/32)", F);
    }
  }
  if (rc) return rc;
  if ((rc = det.reduce_f32(dqkv))) return rc;
  if (!dqkv_planes) return rc;
  // the cls rows (final only now) and the padding rows of the plane tensor; with plane output the patch rows of dqkv (fp32) hold
  // the cls query's contribution only -- the planes are the result
  hipLaunchKernelGGL(attn_bwd_cls_planes_kernel, dim3(B + (rp > B * N ? 1 : 0)), dim3(256), 0, s, dqkv, B, N, ld, dp);
  return check_launch("mt_attn_bwd(cls planes)");
}
