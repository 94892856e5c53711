// Input-sequence builder: expands the compact per-clip description a data loader produces (slots per
// identity, faces actually read, their video-frame numbers and face/frame area ratios) into the four side inputs of
// SizeInvariantTimeSformer.forward, directly in HBM, for a whole batch in one launch.
// Reference: deepfakes_dataset.py:259-287 (size buckets, padding, mask), :315-321 (identities_mask), :324-329 (positions);
// predict.py:285-309, 335-347 (same rules; its mask marks padded slots, the dataset's does not -- see mask_mode).
#include "common.hpp"
#include "../../include/mintime_hip.h"

using namespace mt;

namespace {

// one block (64 lanes) per clip; F <= 64 slots
__global__ __launch_bounds__(64) void build_clip_inputs_kernel(const int* __restrict__ slots, const int* __restrict__ valid,
                                                               const int* __restrict__ frames, const int* __restrict__ ratio,
                                                               unsigned char* __restrict__ mask, unsigned char* __restrict__ ident,
                                                               int* __restrict__ sizes, int64_t* __restrict__ positions, int F,
                                                               int n, int max_ids, int mask_mode) {
  __shared__ int s_frame[64], s_rank[64], s_valid[64], s_start[64], s_len[64];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t == 0) {
    int s = 0, run_max = 0;
    bool any = false;
    for (int i = 0; i < max_ids; ++i) {
      const int cnt = slots[b * max_ids + i], ok = valid[b * max_ids + i], start = s;
      for (int k = 0; k < cnt && s < F; ++k, ++s) {
        const bool real = k < ok;
        int fr;
        if (real) {
          fr = frames[b * F + s];
          run_max = any ? max(run_max, fr) : fr;
          any = true;
        } else {
          fr = any ? run_max : 0;            // padded slot: the largest frame number seen so far in the clip (dataset :271-275)
        }
        s_frame[s] = fr;
        s_valid[s] = real ? 1 : 0;
        s_start[s] = start;
        s_len[s] = cnt;
      }
    }
    for (; s < F; ++s) { s_frame[s] = 0; s_valid[s] = 0; s_start[s] = s; s_len[s] = 0; }
  }
  __syncthreads();
  if (t < F) {
    const int v = s_valid[t];
    mask[b * F + t] = (unsigned char)((mask_mode == 0 || v) ? 1 : 0);
    int bucket = 0;
    if (v) {
      const int r = ratio[b * F + t];
      bucket = r <= 5 ? 1 : (r - 1) / 5 + 1;       // SIZE_EMB_DICT: (0..5) -> 1, (6..10) -> 2, ... (96..100) -> 20
    }
    sizes[b * F + t] = bucket;
    for (int j = 0; j < F; ++j)
      ident[((int64_t)b * F + t) * F + j] = (unsigned char)((j >= s_start[t] && j < s_start[t] + s_len[t]) ? 1 : 0);
    // 1-based rank of this slot's frame number among the clip's distinct frame numbers
    const int x = s_frame[t];
    int rank = 1;
    for (int j = 0; j < F; ++j) {
      const int y = s_frame[j];
      if (y < x) {
        bool first = true;
        for (int k = 0; k < j; ++k) first = first && (s_frame[k] != y);
        rank += first ? 1 : 0;
      }
    }
    s_rank[t] = rank;
  }
  __syncthreads();
  int64_t* prow = positions + (int64_t)b * (1 + (int64_t)F * n);
  if (t == 0) prow[0] = 0;
  for (int i = t; i < F * n; i += 64) {
    const int s = i / n, j = i - s * n;
    prow[1 + i] = (int64_t)(s_rank[s] - 1) * n + 1 + j;
  }
}

}  // namespace

extern "C" int mt_build_clip_inputs(const int* slots, const int* valid, const int* frames, const int* ratio, unsigned char* mask,
                                    unsigned char* identities_mask, int* size_embedding, int64_t* positions, int B, int F,
                                    int num_patches, int max_identities, int mask_mode, void* stream) {
  if (!slots || !valid || !frames || !ratio || !mask || !identities_mask || !size_embedding || !positions)
    return fail(MT_ERR_ARG, "mt_build_clip_inputs: null pointer");
  if (B <= 0 || F <= 0 || F > 64 || num_patches <= 0 || max_identities <= 0 || max_identities > 64)
    return fail(MT_ERR_ARG, "mt_build_clip_inputs: bad sizes (B=%d F=%d patches=%d identities=%d)", B, F, num_patches, max_identities);
  hipLaunchKernelGGL(build_clip_inputs_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, slots, valid, frames, ratio, mask,
                     identities_mask, size_embedding, positions, F, num_patches, max_identities, mask_mode);
  return check_launch("mt_build_clip_inputs");
}
