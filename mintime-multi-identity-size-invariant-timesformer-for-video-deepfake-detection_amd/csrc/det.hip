// Deterministic mode: the switch, the per-stream log arena and the fixed-order reduce kernels (det.hpp has the scheme).
#include "common.hpp"
#include "det.hpp"
#include "../../include/mintime_hip.h"
#include <atomic>
#include <stdlib.h>

namespace mt {

static std::atomic<int>& det_flag() {
  static std::atomic<int> f{getenv("MT_DETERMINISTIC") ? atoi(getenv("MT_DETERMINISTIC")) != 0 : 0};
  return f;
}
int det_enabled() { return det_flag().load(std::memory_order_relaxed); }

struct Arena { void* p = nullptr; size_t bytes = 0; };
static std::mutex g_arena_mu;
static std::map<std::pair<int, hipStream_t>, Arena> g_arenas[2];

// Grown with synchronise + free + allocate on first use.  Never under stream capture: growing is illegal there and would free
// memory an already captured graph may hold -- the caller gets nullptr (an error at the entry point: run one eager step first so
// that the workspace has its size, like harness.GraphedEval's warm-up).  Arenas are keyed by (device, stream handle);
// mt_det_release(stream) frees those of a stream that is about to be destroyed.
void* det_arena(hipStream_t s, size_t bytes, int slot) {
  auto& arenas = g_arenas[slot & 1];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(g_arena_mu);
  Arena& a = arenas[std::make_pair(dev, s)];
  if (bytes > a.bytes) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    if (a.p) { (void)hipStreamSynchronize(s); (void)hipFree(a.p); a.p = nullptr; a.bytes = 0; }
    size_t want = bytes + bytes / 2;
    if (want < (size_t)(8 << 20)) want = (size_t)(8 << 20);
    if (hipMalloc(&a.p, want) != hipSuccess) { a.p = nullptr; return nullptr; }
    a.bytes = want;
  }
  return a.p;
}

int det_release(hipStream_t s) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(g_arena_mu);
  for (auto& arenas : g_arenas) {
    auto it = arenas.find(std::make_pair(dev, s));
    if (it == arenas.end()) continue;
    if (it->second.p) { (void)hipStreamSynchronize(s); (void)hipFree(it->second.p); }
    arenas.erase(it);
  }
  return 0;
}

// ---- log reduce: block (g, chunk of W outputs); 256 threads = (256 / W) rank segments x W outputs
template <typename OUT, int W>
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ vals, const int* __restrict__ base, int R, int P,
                                                         OUT* __restrict__ out, int g0) {
  __shared__ double part[256];
  const int g = g0 + blockIdx.x, j = blockIdx.y * W + (threadIdx.x % W), seg = threadIdx.x / W;
  constexpr int SEGS = 256 / W;
  const int b = base[g];
  if (b < 0) return;
  const int rs = (R + SEGS - 1) / SEGS;
  const int r0 = seg * rs, r1 = min(R, r0 + rs);
  double s = 0.0;
  if (j < P) {
    const float* v = vals + ((int64_t)g * R + r0) * P + j;
    int r = r0;
    for (; r + 4 <= r1; r += 4, v += 4 * (int64_t)P) {      // four loads in flight, added in rank order
      const float a = v[0], b = v[P], c = v[2 * (int64_t)P], d = v[3 * (int64_t)P];
      s += (double)a; s += (double)b; s += (double)c; s += (double)d;
    }
    for (; r < r1; ++r, v += P) s += (double)*v;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (seg == 0 && j < P) {
    double t = part[threadIdx.x];
#pragma unroll
    for (int q = 1; q < SEGS; ++q) t += part[q * W + threadIdx.x];
    out[(int64_t)b + j] += (OUT)t;
  }
}

template <typename OUT>
static int launch_reduce(const DetLog& L, int G, OUT* out, int g0, int count, hipStream_t s) {
  if (count < 0) count = G - g0;
  if (count <= 0 || !out) return 0;
  if (L.P <= 16) hipLaunchKernelGGL((det_reduce_kernel<OUT, 16>), dim3(count, (L.P + 15) / 16), dim3(256), 0, s, L.vals, L.base, L.R, L.P, out, g0);
  else hipLaunchKernelGGL((det_reduce_kernel<OUT, 64>), dim3(count, (L.P + 63) / 64), dim3(256), 0, s, L.vals, L.base, L.R, L.P, out, g0);
  return check_launch("det_reduce");
}

DetScope::DetScope(hipStream_t stream, int groups, int ranks, int p, bool enable, bool base_zero) : s(stream), G(groups), on(false), failed(false) {
  log.vals = nullptr; log.base = nullptr; log.R = ranks; log.P = p;
  if (!enable || !det_enabled() || groups <= 0 || ranks <= 0 || p <= 0) return;
  const size_t vbytes = ((size_t)groups * ranks * p * sizeof(float) + 255) & ~(size_t)255;
  const size_t bbytes = ((size_t)groups * sizeof(int) + 255) & ~(size_t)255;
  char* a = reinterpret_cast<char*>(det_arena(stream, vbytes + bbytes));
  if (!a) { failed = true; return; }                     // (the kernel falls back to atomics; reduce_* turns that into an error)
  log.vals = reinterpret_cast<float*>(a);
  log.base = reinterpret_cast<int*>(a + vbytes);
  (void)hipMemsetAsync(log.vals, 0, vbytes, stream);
  (void)hipMemsetAsync(log.base, base_zero ? 0 : 0xFF, bbytes, stream);
  on = true;
}
static int no_arena() { return fail(MT_ERR_LAUNCH, "deterministic mode: no workspace for the partial-sum log (hipMalloc failed, or the workspace would have to grow under stream capture)"); }
int DetScope::reduce_f32(float* out, int g0, int count) { return failed ? no_arena() : (on ? launch_reduce<float>(log, G, out, g0, count, s) : 0); }
int DetScope::reduce_f64(double* out, int g0, int count) { return failed ? no_arena() : (on ? launch_reduce<double>(log, G, out, g0, count, s) : 0); }

// ---- split-K slabs
__global__ __launch_bounds__(256) void det_slab_reduce_kernel(float* __restrict__ C, int64_t ldc, const float* __restrict__ ws, int splits,
                                                              int M, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
  float* c = C + (int64_t)m * ldc + n;
  float acc = *c;
  const float* w = ws + i;
  const int64_t slab = (int64_t)M * N;
  // eight slabs' loads in flight at a time, added in split order (the first version -- one load, one add -- was a chain of up to 40
  // dependent round trips per thread: 99 us per launch, 10.8 ms per step over the 109 split-K gradients)
  int sidx = 0;
  for (; sidx + 8 <= splits; sidx += 8, w += 8 * slab) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[j * slab];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; sidx < splits; ++sidx, w += slab) acc += *w;
  *c = acc;
}

int det_slab_reduce(float* C, int64_t ldc, const float* ws, int splits, int M, int N, hipStream_t s) {
  const int64_t total = (int64_t)M * N;
  hipLaunchKernelGGL(det_slab_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, C, ldc, ws, splits, M, N);
  return check_launch("det_slab_reduce");
}

struct GemmPending { float* C = nullptr; int64_t ldc = 0; const float* ws = nullptr; int splits = 0, M = 0, N = 0; };
static thread_local GemmPending tl_pending;
struct LogPending { DetLog log; int G = 0; float* out = nullptr; bool on = false; };
static thread_local LogPending tl_log;

int det_gemm_colsum_setup(DetLog& log, int M, int n_half, float* col_sum, hipStream_t s) {
  tl_log = LogPending();
  log.vals = nullptr; log.base = nullptr; log.R = 0; log.P = 0;
  if (!det_enabled() || !col_sum) return 0;
  if (n_half & 31) return fail(MT_ERR_UNSUPPORTED, "deterministic mode: col_sum needs n_half %% 32 == 0");
  DetScope sc(s, 2 * n_half / 32, (M + 255) / 256 * 8, 32);     // ranks up to the end of the last (<= 256-row) block tile: rows past M log zeros
  if (!sc.on) return fail(MT_ERR_LAUNCH, "deterministic mode: no workspace for the column-sum log");
  log = sc.log;
  tl_log.log = sc.log; tl_log.G = sc.G; tl_log.out = col_sum; tl_log.on = true;
  return 0;
}

int det_gemm_setup(float*& C, int64_t& ldc, int64_t& det_slab, int M, int N, int splits, bool row_mapped, hipStream_t s) {
  tl_pending = GemmPending();
  if (!det_enabled() || splits <= 1) return 0;           // one contributor per element: a single add is already deterministic
  if (row_mapped) return fail(MT_ERR_UNSUPPORTED, "deterministic mode: split-K into a row-mapped output");
  float* ws = reinterpret_cast<float*>(det_arena(s, (size_t)splits * M * N * sizeof(float), 1));
  if (!ws) return fail(MT_ERR_LAUNCH, "deterministic mode: no workspace for %d x %d x %d split-K slabs", splits, M, N);
  tl_pending.C = C; tl_pending.ldc = ldc; tl_pending.ws = ws; tl_pending.splits = splits; tl_pending.M = M; tl_pending.N = N;
  C = ws; ldc = N; det_slab = (int64_t)M * N;
  return 0;
}

int det_gemm_finish(hipStream_t s, bool launched) {
  const GemmPending q = tl_pending;
  const LogPending l = tl_log;
  tl_pending = GemmPending();
  tl_log = LogPending();
  if (!launched) return 0;
  if (l.on)
    if (int rc = launch_reduce<float>(l.log, l.G, l.out, 0, -1, s)) return rc;
  if (!q.ws) return 0;
  return det_slab_reduce(q.C, q.ldc, q.ws, q.splits, q.M, q.N, s);
}

}  // namespace mt

namespace mt {
// ---- BatchNorm sums in fixed order (deterministic mode): the producers' fused atomics are switched off and the sums are taken from the
// stored tensor.  mode 0: s1 = sum x, s2 = sum x^2 (forward batch statistics);  mode 1: s1 = sum d, s2 = sum d * (z - mean) * istd (backward).
// Stage 1: block (row chunk, 64-column chunk) -> fp64 partials [chunks][2][C]; stage 2: chunk order -> stats[0 .. 2C) (slot 0).
constexpr int kBnRowsPerBlock = 2048;
template <int W>
__global__ __launch_bounds__(256) void det_bn_sums_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                          const float* __restrict__ mean_istd, int64_t rows, int C, int mode,
                                                          double* __restrict__ part) {
  constexpr int RL = 256 / W;
  __shared__ double red[2][256];
  const int cl = threadIdx.x % W, rl = threadIdx.x / W;
  const int c = blockIdx.y * W + cl;
  const int64_t r0 = (int64_t)blockIdx.x * kBnRowsPerBlock, r1 = min(rows, r0 + kBnRowsPerBlock);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    float mean = 0.f, istd = 0.f;
    if (mode == 1) { mean = mean_istd[c]; istd = mean_istd[C + c]; }
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      const float v = x[r * C + c];
      s1 += (double)v;
      if (mode == 0) s2 += (double)v * (double)v;
      else s2 += (double)(v * ((z[r * C + c] - mean) * istd));
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int q = 0; q < RL; ++q) { a += red[0][q * W + cl]; b += red[1][q * W + cl]; }
    part[((int64_t)blockIdx.x * 2) * C + c] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

__global__ __launch_bounds__(256) void det_bn_sums_finish_kernel(const double* __restrict__ part, int chunks, int C, double* __restrict__ stats) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // (which, c)
  if (i >= 2 * C) return;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * 2 * C + i];
  stats[i] += s;
}
}  // namespace mt

extern "C" int mt_det_bn_sums(const float* x, const float* z, const float* mean_invstd, int64_t rows, int C, int mode, double* stats,
                              void* stream) {
  using namespace mt;
  if (!x || !stats || rows <= 0 || C <= 0) return fail(MT_ERR_ARG, "mt_det_bn_sums: bad arguments");
  if (mode == 1 && (!z || !mean_invstd)) return fail(MT_ERR_ARG, "mt_det_bn_sums: mode 1 needs z and mean_invstd");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (int)((rows + kBnRowsPerBlock - 1) / kBnRowsPerBlock);
  double* part = reinterpret_cast<double*>(det_arena(s, (size_t)chunks * 2 * C * sizeof(double)));
  if (!part) return fail(MT_ERR_LAUNCH, "mt_det_bn_sums: no workspace");
  if (C <= 16) hipLaunchKernelGGL((det_bn_sums_kernel<16>), dim3(chunks, (C + 15) / 16), dim3(256), 0, s, x, z, mean_invstd, rows, C, mode, part);
  else if (C <= 32) hipLaunchKernelGGL((det_bn_sums_kernel<32>), dim3(chunks, (C + 31) / 32), dim3(256), 0, s, x, z, mean_invstd, rows, C, mode, part);
  else hipLaunchKernelGGL((det_bn_sums_kernel<64>), dim3(chunks, (C + 63) / 64), dim3(256), 0, s, x, z, mean_invstd, rows, C, mode, part);
  int rc = check_launch("mt_det_bn_sums");
  if (rc) return rc;
  hipLaunchKernelGGL(det_bn_sums_finish_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, part, chunks, C, stats);
  return check_launch("mt_det_bn_sums(finish)");
}

extern "C" int mt_set_deterministic(int on) {
  mt::det_flag().store(on != 0, std::memory_order_relaxed);
  return 0;
}
extern "C" int mt_get_deterministic(void) { return mt::det_enabled(); }
extern "C" int mt_det_release(void* stream) { return mt::det_release((hipStream_t)stream); }
