// Shared host-side helpers for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <map>
#include <mutex>
#include <utility>

namespace mt {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int cur_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & 15;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, raised once per (kernel, device) instead of on every launch (a runtime call each on
// the host path whose enqueue time bench.py reports); launches come from the caller's thread AND the autograd thread.
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> raised;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  size_t& r = raised[std::make_pair(kernel, dev)];
  if (bytes <= r) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) r = bytes;
  return e;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-2, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

// skinny_wgrad.hip
int stem_wgrad_mfma(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N, int H, int W,
                    int Ho, int Wo, int pad0, hipStream_t st);

enum { MT_ERR_ARG = -1, MT_ERR_LAUNCH = -2, MT_ERR_UNSUPPORTED = -3 };

// BatchNorm partial sums [slots][2][C] (fp64 accumulators fed by one atomic per block and channel).  Default: fp64 atomics -- the
// order in which blocks arrive shows in the last bits.  Deterministic mode (the ABI's `slots` argument NEGATIVE: |slots| accumulators):
// every accumulator is two 64-bit INTEGER limbs, `limb` = |slots| * 2 * C doubles apart, and a block's fp32 partial v is added as
//   limb 0 += trunc(v)                      (exact)
//   limb 1 += round((v - trunc(v)) * 2^44)   (|error| <= 2^-45 per contribution; up to 2^19 contributions per accumulator)
// Integer addition is associative: the sums are the same bits whatever the arrival order, with no second pass over the tensor
// (round 4 re-read every tensor for this: 27 of the deterministic mode's 36 extra ms per step).
__device__ __forceinline__ void stat_add(double* p, int64_t limb, float v) {
  if (limb == 0) {
    atomicAdd(p, (double)v);
    return;
  }
  const float t = truncf(v);
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)(long long)t);
  atomicAdd(reinterpret_cast<unsigned long long*>(p + limb), (unsigned long long)__float2ll_rn((v - t) * 17592186044416.0f));
}
__device__ __forceinline__ double stat_get(const double* p, int64_t limb) {
  if (limb == 0) return *p;
  const long long hi = *reinterpret_cast<const long long*>(p), lo = *reinterpret_cast<const long long*>(p + limb);
  return (double)hi + (double)lo * 5.684341886080801486968994140625e-14;      // 2^-44
}
__host__ __device__ __forceinline__ int stat_slots(int slots) { return slots < 0 ? -slots : (slots > 0 ? slots : 1); }
__host__ __device__ __forceinline__ int64_t stat_limb(int slots, int C) { return slots < 0 ? (int64_t)(-slots) * 2 * C : 0; }

// Work assignment for the persistent, channel-chunked NHWC tile kernels (depthwise conv forward / dgrad / wgrad).
// A chunk of 16 channels is only 64 B of every pixel, so the chunks of ONE spatial tile must run on the same XCD at about
// the same time: its L2 then fetches each 128 B line once and serves the neighbouring chunk from cache.  Block b runs on XCD
// b % 8; within an XCD consecutive blocks take the chunks of one tile.  gridDim.x = 8 * chunks * tiles_per_xcd_in_flight.
__device__ __forceinline__ void xcd_chunk_tile(int chunks, int& chunk, int64_t& tile0, int64_t& stride) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  chunk = j % chunks;
  tile0 = (int64_t)(j / chunks) * 8 + xcd;
  stride = (int64_t)((gridDim.x >> 3) / chunks) * 8;
}

// (n, ty, tx) of a tile index in an [N][ty_n][tx_n] grid, in 32-bit unsigned arithmetic.  A 64-bit division by a run-time value is a
// ~100-instruction scalar sequence, and the tile kernels decompose a tile index twice per tile: 1300 of the 2200 instructions of the
// 3x3 depthwise data-gradient kernel were exactly that (round 6; the kernels are bound by instruction issue, not by bytes).  Tile
// counts are far below 2^31 (xcd_chunk_grid refuses more).
__device__ __forceinline__ void tile_nyx(int64_t tile, int ty_n, int tx_n, int& n, int& ty, int& tx) {
  const unsigned t = (unsigned)tile, t2 = t / (unsigned)tx_n;
  tx = (int)(t - t2 * (unsigned)tx_n);
  const unsigned nn = t2 / (unsigned)ty_n;
  ty = (int)(t2 - nn * (unsigned)ty_n);
  n = (int)nn;
}

// rows / hw for a uniform row index: 32-bit when it fits (a 64-bit division by a run-time value is a ~100-instruction scalar sequence,
// paid once per 64- / 128-row chunk by the streaming kernels)
__device__ __forceinline__ int64_t div_rows(int64_t r, int hw) {
  return (r >> 32) ? r / hw : (int64_t)((unsigned)r / (unsigned)hw);
}

// floor(n / d) for 0 <= n < 2^31 and a run-time d >= 1 as one v_mul_hi_u32 + one shift: with l = ceil(log2 d) and
// M = floor(2^(31+l) / d) + 1 (< 2^32),  n M / 2^(31+l) = n/d + e,  0 < e < 2^-l <= 1/d,  so the floor is exact.  hipcc expands a
// 32-bit division by a run-time value into ~35 VALU instructions; the im2col prologues (gemm_core.hpp) did six per 16-byte gather.
struct FastDiv {
  uint32_t mul;       // 0: d == 1
  int sh;             // l - 1
  __device__ __forceinline__ int div(int n) const { return mul ? (int)(__umulhi((uint32_t)n, mul) >> sh) : n; }
};
inline FastDiv fast_div_of(int d) {
  FastDiv f{0u, 0};
  if (d <= 1) return f;
  int l = 0;
  while (((int64_t)1 << l) < d) ++l;
  f.mul = (uint32_t)((((uint64_t)1 << (31 + l)) / (uint64_t)d) + 1);
  f.sh = l - 1;
  return f;
}

inline unsigned xcd_chunk_grid(int chunks, int64_t ntiles, int target_blocks) {
  if (ntiles >= ((int64_t)1 << 31)) return 0;      // (a zero grid makes the launch fail loudly: tile_nyx works in 32 bits)
  int64_t per_xcd = target_blocks / (8 * chunks);
  const int64_t need = (ntiles + 7) / 8;
  if (per_xcd > need) per_xcd = need;
  if (per_xcd < 1) per_xcd = 1;
  return (unsigned)(8 * chunks * per_xcd);
}


// Workgroup barrier that orders LDS traffic only.  __syncthreads() is preceded by s_waitcnt vmcnt(0): every global load in flight
// -- i.e. the next chunks' prefetch of a streaming kernel -- is drained at each barrier.  Here only the LDS counter is waited for
// (ds_writes visible to the other wavefronts), the loads stay in flight across the barrier and the compiler's own counted waits
// guard their first use.  Global STORES the other wavefronts must see still need __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace mt
