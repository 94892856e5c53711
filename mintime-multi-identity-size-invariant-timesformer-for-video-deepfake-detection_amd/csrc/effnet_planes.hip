// Operand-plane producers for EfficientNet-B0's late stages (14 x 14 and 7 x 7: 12.5-50 K pixel rows, 80-320 block channels,
// 480-1152 expanded channels; reference efficientnet_pytorch/model.py:89-128).
//
// At these sizes the 1x1 convolutions are neither HBM- nor MFMA-bound on the fp32 pipe's kernels: one 128 x 128 tile per CU, a
// serial K loop whose operand prologues (BatchNorm + swish + gate, BatchNorm-backward affine) and in-kernel bf16 split run in the
// MFMA wavefronts (55-110 us per launch for 20-60 MB and 4-8 GFLOP; tools/lab/ef_planes_lab.py, profiles/r05_ef_planes_lab.txt).
// The recipe of the TimeSformer / Xception (gemm_planes.hpp): the operand is written ONCE as blocked bf16 planes by the kernel that
// has it in registers anyway, and the GEMMs that read it (forward, data gradient, weight gradient) only DMA.
//   mt_bn_act_fwd_planes      block output y = bn2(z_p) [* drop-connect gate] [+ y_in] as fp32 AND as planes (the next block's
//                             expand-conv operand, forward and weight gradient)
//   mt_bn_swish_gate_planes   project-conv operand a = swish(bn1(z_d)) * gate as planes (forward and weight gradient)
// (the BatchNorm-backward affine dz = ka du + kb z + kc as planes is mt_bn_bwd_apply_planes, gemm_planes.hip)
// One wavefront per 32 x 16 block of the plane tensor (1 KB per plane), lane = (row, 8 columns); padding rows / columns are zeros.
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include "planes.hpp"

using namespace mt;

namespace {

__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   /* v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division: the swish kernels are VALU-bound */
__device__ __forceinline__ float swish_(float x) { return x * sigmoid_(x); }      // (the formula of effnet_fwd.hip / gemm_core.hpp)

__global__ __launch_bounds__(256) void bn_act_planes_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ res,
                                                            float* __restrict__ y, int R, int C, int act,
                                                            const float* __restrict__ rowscale, int rows_per_group, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    const float g = rowscale ? rowscale[r / rows_per_group] : 1.f;
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      if (c + e + 4 <= C) {
        const int64_t at = (int64_t)r * C + c + e;
        const float4 v = *reinterpret_cast<const float4*>(z + at);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c + e), sh = *reinterpret_cast<const float4*>(shift + c + e);
        float t[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (act == 1) t[q] = swish_(t[q]);
          else if (act == 2) t[q] = fmaxf(t[q], 0.f);
          if (rowscale) t[q] *= g;                       // (same operation order as bn_act_kernel: the two agree to the bit)
        }
        if (res) {
          const float4 rr = *reinterpret_cast<const float4*>(res + at);
          t[0] += rr.x; t[1] += rr.y; t[2] += rr.z; t[3] += rr.w;
        }
        *reinterpret_cast<float4*>(y + at) = make_float4(t[0], t[1], t[2], t[3]);
        x[e] = t[0]; x[e + 1] = t[1]; x[e + 2] = t[2]; x[e + 3] = t[3];
      }
    }
  }
  planes_store8(o, r, c, x);
}

// The same for NARROW tensors (block outputs: 80-320 channels): the kernel above gives each lane 32 bytes of a row, 64 bytes per row and
// wavefront -- with 320-1280-byte rows a wavefront's 32 row segments are 32 separate half lines (27-32 us per launch in the step
// where mt_bn_act_fwd takes 10).  Here a block owns a 32-row block: its rows are ONE contiguous run of memory (read and written
// with consecutive float4s), the results meet in LDS, and each wavefront then emits whole 1 KB plane blocks from there.
__global__ __launch_bounds__(256) void bn_act_planes_rows_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ res,
                                                                 float* __restrict__ y, int R, int C, int act,
                                                                 const float* __restrict__ rowscale, int rows_per_group, PlaneRef o) {
  extern __shared__ float tile_[];                   // [32][C + 4]
  const int pitch = C + 4, CQ = C >> 2;
  const int r0 = blockIdx.x * 32;
  for (int i = threadIdx.x; i < 32 * CQ; i += 256) {
    const int rl = i / CQ, cq = i - rl * CQ, r = r0 + rl;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const int64_t at = (int64_t)r * C + 4 * cq;
      const float4 v = *reinterpret_cast<const float4*>(z + at);
      const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
      t[0] = fmaf(v.x, sc.x, sh.x); t[1] = fmaf(v.y, sc.y, sh.y); t[2] = fmaf(v.z, sc.z, sh.z); t[3] = fmaf(v.w, sc.w, sh.w);
      const float g = rowscale ? rowscale[r / rows_per_group] : 1.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (act == 1) t[q] = swish_(t[q]);
        else if (act == 2) t[q] = fmaxf(t[q], 0.f);
        if (rowscale) t[q] *= g;
      }
      if (res) {
        const float4 rr = *reinterpret_cast<const float4*>(res + at);
        t[0] += rr.x; t[1] += rr.y; t[2] += rr.z; t[3] += rr.w;
      }
      *reinterpret_cast<float4*>(y + at) = make_float4(t[0], t[1], t[2], t[3]);
    }
    *reinterpret_cast<float4*>(tile_ + rl * pitch + 4 * cq) = make_float4(t[0], t[1], t[2], t[3]);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rl = lane >> 1;
  for (int cb = wave; cb < o.cb16; cb += 4) {
    const int c = cb * 16 + (lane & 1) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = c + e < C ? tile_[rl * pitch + c + e] : 0.f;
    planes_store8(o, r0 + rl, c, x);
  }
}

__global__ __launch_bounds__(256) void bn_swish_gate_planes_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ gate,
                                                                   int hw, int R, int C, PlaneRef o) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = blockIdx.y, cb = blockIdx.x * 4 + wave;
  if (cb >= o.cb16) return;
  const int r = rb * 32 + (lane >> 1), c = cb * 16 + (lane & 1) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < R) {
    const float* grow = gate + (int64_t)(r / hw) * C;
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      if (c + e + 4 <= C) {
        const float4 v = *reinterpret_cast<const float4*>(z + (int64_t)r * C + c + e);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c + e), sh = *reinterpret_cast<const float4*>(shift + c + e);
        const float4 g = *reinterpret_cast<const float4*>(grow + c + e);
        // the project convolution's operand as mt_gemm's PRO_BN_SWISH_GATE prologue forms it: swish(z * scale + shift) * gate
        x[e] = swish_(fmaf(v.x, sc.x, sh.x)) * g.x; x[e + 1] = swish_(fmaf(v.y, sc.y, sh.y)) * g.y;
        x[e + 2] = swish_(fmaf(v.z, sc.z, sh.z)) * g.z; x[e + 3] = swish_(fmaf(v.w, sc.w, sh.w)) * g.w;
      }
    }
  }
  planes_store8(o, r, c, x);
}

}  // namespace

extern "C" int mt_bn_act_fwd_planes(const float* z, const float* scale, const float* shift, const float* res, float* y, int rows,
                                    int C, int act, const float* rowscale, int rows_per_group, void* y_planes, void* stream) {
  if (!z || !scale || !shift || !y || !y_planes) return fail(MT_ERR_ARG, "mt_bn_act_fwd_planes: null pointer");
  if (rows <= 0 || C <= 0 || (C & 3)) return fail(MT_ERR_ARG, "mt_bn_act_fwd_planes: C %% 4 != 0 or empty");
  if ((((uintptr_t)z | (uintptr_t)y | (uintptr_t)res | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)y_planes) & 15))
    return fail(MT_ERR_ARG, "mt_bn_act_fwd_planes: 16-byte alignment");
  const int cb16 = (C + 15) >> 4, rp = (rows + 31) & ~31;
  const PlaneRef o{reinterpret_cast<__bf16*>(y_planes), (int64_t)rp * cb16 * 16, cb16, rp};
  if (C <= 512) {                                     // narrow tensors: one block per 32-row block, coalesced rows, planes from LDS
    const size_t lds = (size_t)32 * (C + 4) * sizeof(float);
    (void)ensure_dynamic_lds((const void*)bn_act_planes_rows_kernel, lds);
    hipLaunchKernelGGL(bn_act_planes_rows_kernel, dim3(rp / 32), dim3(256), lds, (hipStream_t)stream, z, scale, shift, res, y, rows, C, act,
                       rowscale, rows_per_group > 0 ? rows_per_group : 1, o);
    return check_launch("mt_bn_act_fwd_planes");
  }
  hipLaunchKernelGGL(bn_act_planes_kernel, dim3((cb16 + 3) / 4, rp / 32), dim3(256), 0, (hipStream_t)stream, z, scale, shift, res, y,
                     rows, C, act, rowscale, rows_per_group > 0 ? rows_per_group : 1, o);
  return check_launch("mt_bn_act_fwd_planes");
}

extern "C" int mt_bn_swish_gate_planes(const float* z, const float* scale, const float* shift, const float* gate, int hw,
                                       void* planes, int rows, int C, void* stream) {
  if (!z || !scale || !shift || !gate || !planes) return fail(MT_ERR_ARG, "mt_bn_swish_gate_planes: null pointer");
  if (rows <= 0 || C <= 0 || (C & 3) || hw <= 0) return fail(MT_ERR_ARG, "mt_bn_swish_gate_planes: bad shape");
  if ((((uintptr_t)z | (uintptr_t)gate | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)planes) & 15))
    return fail(MT_ERR_ARG, "mt_bn_swish_gate_planes: 16-byte alignment");
  const int cb16 = (C + 15) >> 4, rp = (rows + 31) & ~31;
  const PlaneRef o{reinterpret_cast<__bf16*>(planes), (int64_t)rp * cb16 * 16, cb16, rp};
  hipLaunchKernelGGL(bn_swish_gate_planes_kernel, dim3((cb16 + 3) / 4, rp / 32), dim3(256), 0, (hipStream_t)stream, z, scale, shift,
                     gate, hw, rows, C, o);
  return check_launch("mt_bn_swish_gate_planes");
}
