// Deterministic mode (mt_set_deterministic(1) / MT_DETERMINISTIC=1; reference train.py:110 asks cuDNN for deterministic kernels).
//
// Every floating-point atomic of the training step is a sum whose order the hardware picks.  With the switch on, a kernel that would
// add its partial sums with atomics instead WRITES them into a log -- vals[group][rank][P]: `group` names the slice of the output the
// partials belong to (a channel chunk, a weight row block, a (clip, head) ...), `rank` the contributor (a tile, a block, a wavefront),
// both derived from the launch geometry only -- and a reduce kernel launched right behind it on the same stream sums the ranks of
// every group in rank order (fp64 accumulator, one rounding) into the output.  Split-K GEMMs write whole partial tiles into
// [splits][M][N] slabs that are added in split order.  Same inputs, same state -> bit-identical gradients, run after run.
// The logs live in a per-stream arena owned by the library (grown on demand, never shrunk).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mt {

struct DetLog {
  float* vals;      // [G][R][P] partial sums (zero-filled before the launch); null = deterministic mode off, use atomics
  int* base;        // [G] element offset of the group's P outputs in the reduce target (-1 = group without contributor)
  int R, P;
};

// one partial sum: j-th output of `group`, contributed by `rank`; the group's output offset rides on rank 0, j == 0
__device__ __forceinline__ void det_put(const DetLog& L, int group, int rank, int j, float v) {
  L.vals[((int64_t)group * L.R + rank) * L.P + j] = v;
}
__device__ __forceinline__ void det_base(const DetLog& L, int group, int64_t off) { L.base[group] = (int)off; }

int det_enabled();                                       // the switch
// Arena of the stream: at least `bytes`, 256-byte aligned, valid until the next det_arena call on the same stream.
void* det_arena(hipStream_t s, size_t bytes, int slot = 0);      // slot 0: logs, slot 1: split-K slabs (a launch may hold one of each)

// Host side of one logged launch:   DetScope d(stream, G, R, P);  kernel<<<...>>>(..., d.log);  d.reduce_f32(out) / d.reduce_f64(out)
struct DetScope {
  DetLog log;
  hipStream_t s;
  int G;
  bool on;
  bool failed;        // the switch is on but the arena could not be had: the kernel fell back to atomics, reduce_* reports it
  // base_zero: every group's output offset is 0 (the kernel need not write it; reduce group ranges into different targets)
  DetScope(hipStream_t stream, int groups, int ranks, int p, bool enable = true, bool base_zero = false);
  // out[base[g] + j] += sum_r vals[g][r][j]   (r ascending, fp64 accumulator); groups [g0, g0 + count), count < 0 = all
  int reduce_f32(float* out, int g0 = 0, int count = -1);
  int reduce_f64(double* out, int g0 = 0, int count = -1);
};

// Split-K slabs: ws[splits][M][N] (dense, ldc == N) written by the GEMM's ATOMIC epilogue; C[m*ldc + n] += sum_s ws[s][m][n], s ascending.
int det_slab_reduce(float* C, int64_t ldc, const float* ws, int splits, int M, int N, hipStream_t s);

// The GEMM launchers' side of it.  At the point where a launcher has fixed its split count it calls det_gemm_setup, which (switch on,
// splits > 1) redirects C / ldc to the arena and sets det_slab; the C-ABI entry point calls det_gemm_finish after a successful launch
// (thread-local hand-over: the launch path has four levels and two translation units).  Returns nonzero on an unsupported form.
int det_gemm_setup(float*& C, int64_t& ldc, int64_t& det_slab, int M, int N, int splits, bool row_mapped, hipStream_t s);
int det_gemm_finish(hipStream_t s, bool launched);
// column sums taken in a GEMM epilogue (GEGLU_BWD's col_sum: the bias gradient): log with one group per 32 columns of [2][n_half],
// one rank per 32 rows; reduced into col_sum by det_gemm_finish
int det_gemm_colsum_setup(DetLog& log, int M, int n_half, float* col_sum, hipStream_t s);

}  // namespace mt
