// Blocked bf16 plane tensors: the operand format of the plane-operand GEMM loop (gemm_planes.hpp), and the helpers the PRODUCER
// kernels use to write it (LayerNorm forward / backward, attention, the GEGLU epilogues, mt_split_planes_blk).
//
//   planes[3][Rp/32][Cp/16][32][16] bf16,  Rp = rows rounded up to 32, Cp = columns rounded up to 16;  x = p0 + p1 + p2 exactly
//   (round-to-nearest bf16 at each level), padding rows / columns hold zeros (a ragged contraction end reads them).
// A 32 x 16 block is 1 KB: one LDS-DMA piece of a k-contiguous operand tile; eight 128-byte runs of a k-major one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mt {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

struct PlaneRef {          // one plane tensor
  __bf16* p;               // plane 0
  int64_t pstride;         // elements between planes = Rp * Cp
  int cb16;                // Cp / 16
  int rows_pad;            // Rp
};

__host__ __device__ __forceinline__ int64_t planes_blk_off(int64_t r, int64_t c, int64_t cb16) {
  return ((r >> 5) * cb16 + (c >> 4)) * 512 + (r & 31) * 16 + (c & 15);
}

// exact three-piece split of two values
__device__ __forceinline__ void split2(f32x2_t v, bf16x2_t& a, bf16x2_t& b, bf16x2_t& c) {
  // no contraction across this boundary: with -ffp-contract=fast a caller's final multiply (o * inv) would fuse into the residual
  // subtraction below, and the planes would hold the pieces of the UNROUNDED product instead of the fp32 value
#pragma clang fp contract(off)
  a = __builtin_convertvector(v, bf16x2_t);
  const f32x2_t r = v - __builtin_convertvector(a, f32x2_t);
  b = __builtin_convertvector(r, bf16x2_t);
  const f32x2_t t = r - __builtin_convertvector(b, f32x2_t);
  c = __builtin_convertvector(t, bf16x2_t);
}

// four consecutive columns (c % 4 == 0) of row r
__device__ __forceinline__ void planes_store4(const PlaneRef& o, int r, int c, float v0, float v1, float v2, float v3) {
  bf16x2_t a0, b0, c0, a1, b1, c1;
  split2(f32x2_t{v0, v1}, a0, b0, c0);
  split2(f32x2_t{v2, v3}, a1, b1, c1);
  __bf16* dst = o.p + planes_blk_off(r, c, o.cb16);
  *reinterpret_cast<bf16x4_t*>(dst) = bf16x4_t{a0[0], a0[1], a1[0], a1[1]};
  *reinterpret_cast<bf16x4_t*>(dst + o.pstride) = bf16x4_t{b0[0], b0[1], b1[0], b1[1]};
  *reinterpret_cast<bf16x4_t*>(dst + 2 * o.pstride) = bf16x4_t{c0[0], c0[1], c1[0], c1[1]};
}

// eight consecutive columns (c % 8 == 0) of row r
__device__ __forceinline__ void planes_store8(const PlaneRef& o, int r, int c, const float (&v)[8]) {
  bf16x8_t x0, x1, x2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bf16x2_t a, b, cc;
    split2(f32x2_t{v[2 * q], v[2 * q + 1]}, a, b, cc);
    x0[2 * q] = a[0]; x0[2 * q + 1] = a[1];
    x1[2 * q] = b[0]; x1[2 * q + 1] = b[1];
    x2[2 * q] = cc[0]; x2[2 * q + 1] = cc[1];
  }
  __bf16* dst = o.p + planes_blk_off(r, c, o.cb16);
  *reinterpret_cast<bf16x8_t*>(dst) = x0;
  *reinterpret_cast<bf16x8_t*>(dst + o.pstride) = x1;
  *reinterpret_cast<bf16x8_t*>(dst + 2 * o.pstride) = x2;
}

// zeros into the padding rows [rows, rows_pad) of every column block (one caller block; D % 4 == 0 columns)
__device__ __forceinline__ void planes_zero_pad(const PlaneRef& o, int rows, int tid, int nthreads) {
  const int npad = o.rows_pad - rows, nq = o.cb16 * 4;
  for (int i = tid; i < npad * nq; i += nthreads) planes_store4(o, rows + i / nq, (i % nq) * 4, 0.f, 0.f, 0.f, 0.f);
}

}  // namespace mt
