// _conv_stem forward: dense 3x3 stride-2 TF-"SAME" convolution of the crops, 3 -> 32 channels (reference
// efficientnet_pytorch/model.py:173,276 with Conv2dStaticSamePadding utils.py:248-276), plus the BatchNorm batch statistics of its
// output (train mode).  x [N, H, W, 3] fp32 or uint8 (raw 0..255, next-row f2), z [N*Ho*Wo, 32], stats [slots][2][32] fp64.
//
// 864 FMAs per output pixel against 128 bytes written: HBM-bound if the arithmetic is cheap enough.  The im2col-prologue GEMM
// (K = 27 padded to 28, gathered per fragment) ran 360 us for a 256-crop batch = 1.7 TB/s; the first direct kernel (one pixel per
// thread, weights broadcast from LDS) was LDS-read bound at 1.15 ms.  Here one block takes one OUTPUT ROW of one image at a time
// (persistent, next row's input in flight): the three input rows it needs go to LDS once (zero columns left and right stand in
// for the padding), each of the four wavefronts owns 32 consecutive output pixels, and the 27-tap contraction is 14 steps of
// v_mfma_f32_32x32x2_f32 with the A operand gathered from the LDS rows (lane = pixel, stride 6 floats: conflict-free) and the
// [28][32] weight tile as B.  The accumulator layout (lane = channel) makes the stores full 128-byte pixel rows and leaves the
// per-channel sums as lane-private registers for the whole launch.
// Algorithmic bytes = N*H*W*3 (* 4 for fp32 input) + N*Ho*Wo*32*4.
#include "common.hpp"
#include <stdint.h>
#include <type_traits>

namespace {
using namespace mt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CO = 32;
constexpr int MAXW = 512;                  // widest crop row the LDS tile holds

struct StemArgs {
  const void* x; const float* w; float* z; double* stats;
  int slots, x_u8, N, H, W, Ho, Wo, pad0;
};

// LV = prefetch slots per thread per register set: 3 input rows of W pixels over 256 threads (8 up to W = 226, 18 up to MAXW)
template <int LV>
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemArgs p) {
  extern __shared__ float smem[];
  const int pitch = (p.W + 2) * 3;         // one zero pixel left and right of every row
  float* rows = smem;                      // [3][pitch]
  float* wl = rows + 3 * pitch;            // [28][32]  tap = (kh*3 + kw)*3 + ci; tap 27 = 0
  float* red = wl + 28 * CO;               // [4][2][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh2 = lane >> 5, cl = lane & 31;

  for (int i = tid; i < 3 * pitch; i += 256) rows[i] = 0.f;
  for (int i = tid; i < 28 * CO; i += 256) {
    const int co = i & 31, t = i >> 5;
    float v = 0.f;
    if (t < 27) {
      const int ci = t % 3, kk = t / 3, kw = kk % 3, kh = kk / 3;
      v = p.w[((co * 3 + ci) * 3 + kh) * 3 + kw];               // torch layout [co][ci][kh][kw]
    }
    wl[i] = v;
  }

  // this lane's A-operand offsets inside the LDS rows for the 14 k-steps (taps 2*ks + kh2): row kh, column (kw - pad0 + 1)*3 + ci,
  // relative to the pixel base 6*ow; the padded tap 27 reads the (always zero) first column of row 0 ... with a zero weight
  int toff[14];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks) {
    const int t = 2 * ks + kh2;
    const int tt = t < 27 ? t : 0;
    const int ci = tt % 3, kk = tt / 3, kw = kk % 3, kh = kk / 3;
    toff[ks] = kh * pitch + (kw - p.pad0 + 1) * 3 + ci;
  }
  const int ow = wave * 32 + cl;           // this lane's pixel (A operand) in the current 128-pixel span
  const int items = p.N * p.Ho;
  const int rowf = p.W * 3;                // input elements per row
  const int nslot = (3 * rowf + 255) / 256;
  float pre[2][LV];                         // two register sets: the rows of the next TWO items are in flight (one item's multiply is
                                           // far shorter than a load's latency; with one set every item waited for its loads)
  // loop-invariant placement of this thread's prefetch slots: input row r (0..2), element c of that row, LDS offset
  // (the divisions cost more than the convolution if they are redone per item)
  int sr[LV], sc[LV], sl[LV];
#pragma unroll
  for (int i = 0; i < LV; ++i) {
    const int e = tid + 256 * i, ec = min(e, 3 * rowf - 1);
    sr[i] = ec / rowf;
    sc[i] = ec - sr[i] * rowf;
    sl[i] = e < 3 * rowf ? sr[i] * pitch + 3 + sc[i] : -1;
  }

  auto fetch = [&](auto set_c, int item) {  // unconditional loads on clamped offsets (a predicated load de-pipelines), zeroed after
    constexpr int SET = decltype(set_c)::value;
    const int n = item / p.Ho, oh = item - n * p.Ho;
#pragma unroll
    for (int i = 0; i < LV; ++i) {
      if (i < nslot) {                       // uniform
        const int ih = 2 * oh + sr[i] - p.pad0;
        const int64_t off = ((int64_t)n * p.H + min(max(ih, 0), p.H - 1)) * rowf + sc[i];
        const float v = p.x_u8 ? (float)reinterpret_cast<const uint8_t*>(p.x)[off] : reinterpret_cast<const float*>(p.x)[off];
        pre[SET][i] = (ih >= 0 && ih < p.H) ? v : 0.f;
      }
    }
  };
  auto stage = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < LV; ++i)
      if (i < nslot && sl[i] >= 0) rows[sl[i]] = pre[SET][i];
  };

  float s1 = 0.f, s2 = 0.f;                // this lane's channel (cl), its half's pixels, all items of this block
  const int stride = gridDim.x;
  auto process = [&](auto set_c, int item) {
    stage(set_c);
    __syncthreads();
    if (item + 2 * stride < items) fetch(set_c, item + 2 * stride);      // in flight through this item and the next
    for (int span = 0; span < p.Wo; span += 128) {
      const int px = span + ow;            // A-operand pixel of this lane; lanes past the row are clamped (their results are dropped)
      const float* a_w = rows + 6 * min(px, p.Wo - 1);
      if (span + wave * 32 < p.Wo) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_w[toff[ks]], wl[(2 * ks + kh2) * CO + cl], acc, 0, 0, 0);
        float* zo = p.z + ((int64_t)item * p.Wo + span + wave * 32) * CO + cl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pr = (r & 3) + 8 * (r >> 2) + 4 * kh2;       // acc[r] = out[pixel pr][channel cl]
          if (span + wave * 32 + pr < p.Wo) {
            const float v = acc[r];
            zo[pr * CO] = v;
            s1 += v; s2 = fmaf(v, v, s2);
          }
        }
      }
    }
    __syncthreads();                       // the rows are free again
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int item = blockIdx.x;
  if (item < items) fetch(I0{}, item);
  if (item + stride < items) fetch(I1{}, item + stride);
  __syncthreads();                         // zero fill and weights in place
  for (; item < items; item += 2 * stride) {
    process(I0{}, item);
    if (item + stride < items) process(I1{}, item + stride);
  }
  if (p.stats) {
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (kh2 == 0) { red[(wave * 2 + 0) * CO + cl] = s1; red[(wave * 2 + 1) * CO + cl] = s2; }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid >> 5, c = tid & 31;
      const float v = red[(0 * 2 + which) * CO + c] + red[(1 * 2 + which) * CO + c] + red[(2 * 2 + which) * CO + c] + red[(3 * 2 + which) * CO + c];
      stat_add(p.stats + ((int64_t)(blockIdx.x % stat_slots(p.slots)) * 2 + which) * CO + c, stat_limb(p.slots, CO), v);
    }
  }
}

}  // namespace

static int stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* z, double* stats, int slots, int N, int H, int W, bool valid,
                         void* stream) {
  if (!x || !w || !z) return fail(MT_ERR_ARG, "mt_stem_conv_fwd: null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || W > MAXW || (int64_t)N * H > (1 << 30))
    return fail(MT_ERR_ARG, "mt_stem_conv_fwd: crops of 1..%d columns, N*H < 2^30", MAXW);
  if (valid && (H < 3 || W < 3)) return fail(MT_ERR_ARG, "mt_stem_conv_fwd_valid: crops of at least 3 x 3");
  // TF-SAME: ceil(H / 2) outputs, the missing rows / columns are zeros; valid (padding 0): floor((H - 3) / 2) + 1 outputs, no padding
  const int Ho = valid ? (H - 3) / 2 + 1 : (H + 1) / 2, Wo = valid ? (W - 3) / 2 + 1 : (W + 1) / 2;
  const int padt_h = valid ? 0 : max((Ho - 1) * 2 + 3 - H, 0), padt_w = valid ? 0 : max((Wo - 1) * 2 + 3 - W, 0);
  if (padt_h / 2 != padt_w / 2) return fail(MT_ERR_UNSUPPORTED, "mt_stem_conv_fwd: H and W must need the same leading padding");
  StemArgs a{x, w, z, stats, slots != 0 ? slots : 1, x_is_u8 ? 1 : 0, N, H, W, Ho, Wo, padt_h / 2};
  const size_t smem = ((size_t)3 * (W + 2) * 3 + 28 * CO + 8 * CO) * sizeof(float);
  const int64_t items = (int64_t)N * Ho;
  const int blocks = (int)(items < 2048 ? items : 2048);
  if (3 * W * 3 <= 8 * 256) hipLaunchKernelGGL(stem_mfma_kernel<8>, dim3(blocks), dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(stem_mfma_kernel<(MAXW * 9 + 255) / 256>, dim3(blocks), dim3(256), smem, (hipStream_t)stream, a);
  return check_launch("mt_stem_conv_fwd");
}

extern "C" int mt_stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* z, double* stats, int slots, int N, int H, int W,
                                void* stream) {
  return stem_conv_fwd(x, x_is_u8, w, z, stats, slots, N, H, W, false, stream);
}

extern "C" int mt_stem_conv_fwd_valid(const void* x, int x_is_u8, const float* w, float* z, double* stats, int slots, int N, int H, int W,
                                      void* stream) {
  return stem_conv_fwd(x, x_is_u8, w, z, stats, slots, N, H, W, true, stream);
}
