// Size-Invariant TimeSformer forward kernels other than the dense contractions (gfx950).
//
//   mt_layernorm_fwd      PreNorm's nn.LayerNorm                (size_invariant_timesformer.py:18-26)
//   mt_embed_fwd          cls token + pos_emb + size_emb        (:231-248)
//   mt_attn_fwd           divided space/time attention core incl. the cls query (:80-87, :112-141)
//   mt_head_fwd           to_out = LayerNorm + Linear(dim,1) on the cls row (:270-276)
//
// Attention layout decisions (MI355X-first, not a translation of the reference's rearranges):
//   * q/k/v are consumed straight out of the QKV GEMM's [B, N, 3*H*64] buffer and the result is written
//     in merged-head [B, N, H*64] layout -> none of the reference's chunk/rearrange/cat/repeat copies exist.
//   * groups are tiny (9 keys for time, 50 for space, d=64): at fp32 the MFMA rate equals the VALU rate on
//     CDNA4, and a 32x32 tile would be mostly padding, so one LANE owns one query row: scores, softmax and
//     P.V are lane-local (no cross-lane reduction at all); K/V rows are staged once per wavefront in LDS
//     and read as broadcast / conflict-free ds_read_b128.
//   * masks are read from mask[B,F] / identities_mask[B,F,F] directly; the reference's materialised
//     [(B*H*49), F, F+1] frame mask (:252-255) never exists.  Masked logits are FILLED with -FLT_MAX (:82-85).
#include "../../include/mintime_hip.h"
#include "common.hpp"
#include <stdlib.h>
#include <float.h>

using namespace mt;

#include "planes.hpp"
namespace {

constexpr int DH = 64;   // dim_head (config: dim-head 64)

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

// ---------------------------------------------------------------------------------------- LayerNorm
// one wavefront per row; D % 4 == 0, D <= 1024
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ stats, int rows, int D, float eps, PlaneRef yp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) {
    if (yp.p && row < yp.rows_pad)                   // padding rows of the last row block: zeros
      for (int q = lane; q < D >> 2; q += 64) planes_store4(yp, row, q * 4, 0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
  const int nq = D >> 2;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + i * 64;
    v[i] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + i * 64;
    if (q < nq) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  float4* yr = y ? reinterpret_cast<float4*>(y + (int64_t)row * D) : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + i * 64;
    if (q < nq) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[q];
      const float4 b = reinterpret_cast<const float4*>(beta)[q];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (yr) yr[q] = o;
      if (yp.p) planes_store4(yp, row, q * 4, o.x, o.y, o.z, o.w);
    }
  }
  if (stats && lane == 0) {
    stats[2 * (int64_t)row] = mean;
    stats[2 * (int64_t)row + 1] = rstd;
  }
}

// ---------------------------------------------------------------------------------------- embeddings
// x[b,0,:]   = cls + pos_emb[positions[b,0]] + size_emb[0]
// x[b,1+t,:] += pos_emb[positions[b,1+t]] + size_emb[size[b, t / n]]        (x holds the patch-embedding GEMM output)
__global__ __launch_bounds__(256) void embed_fwd_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                        const float* __restrict__ pos_emb, const float* __restrict__ size_emb,
                                                        const int64_t* __restrict__ positions, const int* __restrict__ sizes,
                                                        int B, int N, int n, int F, int D, int pos_rows, int size_rows,
                                                        int* __restrict__ err) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * N) return;
  const int b = row / N, t = row - b * N;
  int64_t pi = positions ? positions[row] : (int64_t)t;
  int si = 0;
  if (t > 0 && sizes) si = sizes[b * F + (t - 1) / n];
  // nn.Embedding raises on an out-of-range index; here the access is clamped into the table and a sticky flag is raised
  if (pi < 0 || pi >= pos_rows || si < 0 || si >= size_rows) {
    if (err && lane == 0) atomicOr(err, (pi < 0 || pi >= pos_rows) ? 1 : 2);
    pi = pi < 0 ? 0 : (pi >= pos_rows ? pos_rows - 1 : pi);
    si = si < 0 ? 0 : (si >= size_rows ? size_rows - 1 : si);
  }
  float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * D);
  const float4* pr = reinterpret_cast<const float4*>(pos_emb + pi * D);
  const float4* sr = size_emb ? reinterpret_cast<const float4*>(size_emb + (int64_t)si * D) : nullptr;
  for (int q = lane; q < (D >> 2); q += 64) {
    float4 v = t == 0 ? reinterpret_cast<const float4*>(cls)[q] : xr[q];
    const float4 p = pr[q];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    if (sr) { const float4 s = sr[q]; v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; }
    xr[q] = v;
  }
}

// ---------------------------------------------------------------------------------------- attention: patch queries
// MODE 0 = time  : group = (b,h,patch p): F queries (tokens 1+f*n+p), keys = cls + the same F tokens, identity mask
// MODE 1 = space : group = (b,h,frame f): n queries (tokens 1+f*n+p), keys = cls + the same n tokens, no mask
// One lane per query. NKEYS = keys per query (F+1 or n+1).  PPW = patches per wavefront in time mode.
// Per wavefront LDS: a [ROWS][STRIDE] K tile (re-used for V) + a [NKEYS][64] score tile (lane-contiguous).
template <int MODE, int NKEYS, int PPW, int STRIDE, int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_patch_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 const uint8_t* __restrict__ mask,
                                                                 const uint8_t* __restrict__ ident,
                                                                 int B, int H, int F, int n, float scale, const PlaneRef op) {
  constexpr int ROWS = MODE == 0 ? 1 + PPW * (NKEYS - 1) : NKEYS;
  constexpr int WAVE_LDS = ROWS * STRIDE + NKEYS * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* kv = lds + wave * WAVE_LDS;
  float* sc = kv + ROWS * STRIDE + lane;

  const int N = 1 + F * n;
  const int inner = H * DH;
  const int ld = 3 * inner;
  const int chunks = MODE == 0 ? (n + PPW - 1) / PPW : F;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * chunks) return;       // wave-uniform
  const int c = (int)(wid % chunks);
  const int bh = (int)(wid / chunks);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;

  // my query
  int qtok = -1, pl = 0, fq = 0;
  if (MODE == 0) {
    pl = lane / (NKEYS - 1); fq = lane % (NKEYS - 1);
    const int p = c * PPW + pl;
    if (pl < PPW && p < n) qtok = 1 + fq * n + p;
    if (pl >= PPW) pl = 0;   // idle lanes must still address rows inside this wavefront's LDS slice
  } else {
    if (lane < n) qtok = 1 + c * n + lane;
  }
  const int row1 = MODE == 0 ? 1 + pl * (NKEYS - 1) : 1;   // LDS row of my key j = 1

  auto stage = [&](int which) {   // which: 1 = K, 2 = V
    for (int r0 = 0; r0 < ROWS; r0 += 4) {
      const int r = r0 + (lane >> 4), c4 = lane & 15;
      if (r < ROWS) {
        int tok = 0;
        if (r > 0) {
          if (MODE == 0) {
            const int pl2 = (r - 1) / (NKEYS - 1), f2 = (r - 1) % (NKEYS - 1);
            const int p = c * PPW + pl2;
            tok = p < n ? 1 + f2 * n + p : -1;
          } else {
            tok = 1 + c * n + (r - 1);
          }
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tok >= 0) v = *reinterpret_cast<const float4*>(base + (int64_t)tok * ld + which * inner + c4 * 4);
        *reinterpret_cast<float4*>(kv + r * STRIDE + c4 * 4) = v;
      }
    }
  };

  stage(1);
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

  float mx = -FLT_MAX;
#pragma unroll 2
  for (int j = 0; j < NKEYS; ++j) {
    const float* kr = kv + (j == 0 ? 0 : row1 + j - 1) * STRIDE;
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
      if (!ok) a = -FLT_MAX;                      // masked_fill_(~mask, -finfo.max)  (:82-85)
    }
    sc[j * 64] = a;
    mx = fmaxf(mx, a);
  }

  __builtin_amdgcn_wave_barrier();
  stage(2);
  __builtin_amdgcn_wave_barrier();

  float o[DH];
#pragma unroll
  for (int i = 0; i < DH; ++i) o[i] = 0.f;
  float sum = 0.f;
#pragma unroll 2
  for (int j = 0; j < NKEYS; ++j) {
    const float* vr = kv + (j == 0 ? 0 : row1 + j - 1) * STRIDE;
    const float pj = expf(sc[j * 64] - mx);
    sum += pj;
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 vv = *reinterpret_cast<const float4*>(vr + i * 4);
      o[4 * i] = fmaf(pj, vv.x, o[4 * i]); o[4 * i + 1] = fmaf(pj, vv.y, o[4 * i + 1]);
      o[4 * i + 2] = fmaf(pj, vv.z, o[4 * i + 2]); o[4 * i + 3] = fmaf(pj, vv.w, o[4 * i + 3]);
    }
  }
  if (qtok >= 0) {
    const float inv = 1.0f / sum;
    if (out) {
      float* orow = out + ((int64_t)b * N + qtok) * inner + h * DH;
#pragma unroll
      for (int i = 0; i < DH / 4; ++i)
        *reinterpret_cast<float4*>(orow + i * 4) =
            make_float4(o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv);
    }
    if (op.p) {                                       // the out-projection's operand planes: a lane owns 64 columns of one row
#pragma unroll
      for (int i = 0; i < DH / 4; ++i)
        planes_store4(op, b * N + qtok, h * DH + i * 4, o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv);
    }
  }
}

// ---------------------------------------------------------------------------------------- time attention, one patch per wavefront
// One wavefront per (b, h, patch): F queries x (cls + F) keys (the backward twin is attn_time_bwd_kernel, tsf_bwd.hip).  3F + 2
// coalesced 256-byte row loads with lane = d; q and k are mirrored in LDS for the scores (lane = (query f, key j): one 64-long dot
// product, softmax across the F lanes of a query by shuffles); o_f = sum_j P_fj v_j with lane = d from the v registers and broadcast
// reads of P; the rows go out eight columns per lane.  7 KB of LDS per wavefront against the 20+ KB of the 7-patch kernel.
template <int F, int WPB>
__global__ __launch_bounds__(WPB * 64, F == 8 ? 4 : 2) void attn_time_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                                const uint8_t* __restrict__ mask,
                                                                                const uint8_t* __restrict__ ident, int B, int H, int n,
                                                                                float scale, const PlaneRef op) {
  constexpr int NK = F + 1, ST = 68, SPP = (NK + 3) & ~3;
  constexpr int WAVE_LDS = (2 * F + 1) * ST + F * SPP;
  constexpr int LPF = 64 / F;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* qT = lds + wave * WAVE_LDS;                         // [F][ST] scaled q, later o
  float* kT = qT + F * ST;                                   // [NK][ST] k (row 0 = cls)
  float* Pm = kT + NK * ST;                                  // [F][SPP] scores, then P (column 0 = cls key)
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * n) return;                     // wave-uniform; no block-level barrier below
  const int p = (int)(wid % n);
  const int bh = (int)(wid / n);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;

  float v[NK];
  {
    float q[F], k[NK];
    k[0] = base[inner + lane];
    v[0] = base[2 * inner + lane];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const int64_t tok = 1 + f * n + p;
      q[f] = base[tok * ld + lane];
      k[f + 1] = base[tok * ld + inner + lane];
      v[f + 1] = base[tok * ld + 2 * inner + lane];
    }
#pragma unroll
    for (int f = 0; f < F; ++f) qT[f * ST + lane] = q[f] * scale;
#pragma unroll
    for (int j = 0; j < NK; ++j) kT[j * ST + lane] = k[j];
  }
  __builtin_amdgcn_wave_barrier();

  {   // the cls key's column, LPF lanes per query
    const int f = lane / LPF, c = lane % LPF;
    const float* qr = qT + f * ST + c * F;
    const float* kr = kT + c * F;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < F; i += 4) {
      const float4 qq = *reinterpret_cast<const float4*>(qr + i), kk = *reinterpret_cast<const float4*>(kr + i);
      a = fmaf(qq.x, kk.x, a); a = fmaf(qq.y, kk.y, a); a = fmaf(qq.z, kk.z, a); a = fmaf(qq.w, kk.w, a);
    }
#pragma unroll
    for (int o = 1; o < LPF; o <<= 1) a += __shfl_xor(a, o);
    if (c == 0) Pm[f * SPP] = a;
  }
  __builtin_amdgcn_wave_barrier();

#pragma unroll 1
  for (int pass = 0; pass < F * F / 64; ++pass) {
    const int pi = pass * 64 + lane;
    const int f = pi / F, jj = pi % F, j = jj + 1;
    const float* qr = qT + f * ST;
    const float* kr = kT + j * ST;
    float s_a = 0.f, s_b = 0.f;
#pragma unroll 4
    for (int i = 0; i < DH / 8; ++i) {
      const float4 q0 = *reinterpret_cast<const float4*>(qr + 8 * i), q1 = *reinterpret_cast<const float4*>(qr + 8 * i + 4);
      const float4 k0 = *reinterpret_cast<const float4*>(kr + 8 * i), k1 = *reinterpret_cast<const float4*>(kr + 8 * i + 4);
      s_a = fmaf(q0.x, k0.x, s_a); s_a = fmaf(q0.y, k0.y, s_a); s_a = fmaf(q0.z, k0.z, s_a); s_a = fmaf(q0.w, k0.w, s_a);
      s_b = fmaf(q1.x, k1.x, s_b); s_b = fmaf(q1.y, k1.y, s_b); s_b = fmaf(q1.z, k1.z, s_b); s_b = fmaf(q1.w, k1.w, s_b);
    }
    float s = s_a + s_b;
    if (!(mask[b * F + jj] && ident[(b * F + f) * F + jj])) s = -FLT_MAX;        // masked_fill_(~mask, -finfo.max)  (:82-85)
    const float s0 = Pm[f * SPP];
    float mx = fmaxf(s, s0);
#pragma unroll
    for (int o = 1; o < F; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = expf(s - mx), e0 = expf(s0 - mx);
    float sum = e;
#pragma unroll
    for (int o = 1; o < F; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / (sum + e0);
    Pm[f * SPP + j] = e * inv;
    if (jj == 0) Pm[f * SPP] = e0 * inv;
  }
  __builtin_amdgcn_wave_barrier();

#pragma unroll 2
  for (int f = 0; f < F; ++f) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) a = fmaf(Pm[f * SPP + j], v[j], a);
    qT[f * ST + lane] = a;
  }
  __builtin_amdgcn_wave_barrier();

  for (int it = lane; it < F * 8; it += 64) {
    const int f = it >> 3, seg = it & 7;
    const int tok = 1 + f * n + p;
    const float* src = qT + f * ST + seg * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
    const float o[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    if (out) {
      float* g = out + ((int64_t)b * N + tok) * inner + h * DH + seg * 8;
      *reinterpret_cast<float4*>(g) = a0;
      *reinterpret_cast<float4*>(g + 4) = a1;
    }
    if (op.p) planes_store8(op, b * N + tok, h * DH + seg * 8, o);
  }
}

// ---------------------------------------------------------------------------------------- space attention on the matrix cores
// One wavefront per (b, h, frame): n = 49 patch queries x (cls + 49) keys x dim_head 64, padded to 64 x 64 and computed TRANSPOSED
// so that nothing ever has to change layout between the two products (no LDS at all):
//   S^T[key][q] = sum_d K[key][d] * (scale Q)[q][d]     A = K rows, B = Q rows: lane (row, half) holds d = 32*half .. +31 of its row,
//                                                       one 128-byte contiguous read per lane; the k index of MFMA step ks is
//                                                       d = 32*half + ks on both operands (any bijection works if A and B agree)
//   softmax over keys = over the accumulator slots of ONE lane (+ one xor-32 exchange): the C layout of S^T keeps a query per lane
//   O^T[d][q]   = sum_key V[key][d] * P^T[key][q]       B = P^T straight out of the accumulators: step ks feeds slot (ks>>4, ks&15),
//                                                       i.e. key(ks, half) = 32*(ks>>4) + (ks&3) + 8*((ks&15)>>2) + 4*half;
//                                                       A = V read by columns for exactly that key order (coalesced across d)
// 256 v_mfma_f32_32x32x2_f32 per group against ~100 k VALU FMAs + 100 LDS reads per lane in the one-lane-per-query kernel.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mfma_slot_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_space_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                      int B, int H, int F, int n, float scale, const PlaneRef op) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane & 31, hf = lane >> 5;
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
  if (wid >= (int64_t)B * H * F) return;                 // wave-uniform
  const int f = (int)(wid % F);
  const int bh = (int)(wid / F);
  const int h = bh % H, b = bh / H;
  const float* base = qkv + (int64_t)b * N * ld + h * DH;
  const int t0 = 1 + f * n;                              // first patch token of the frame
  auto tok_k = [&](int key) { return key == 0 ? 0 : t0 + min(key, n) - 1; };   // keys >= n+1 are padding (masked below)
  auto tok_q = [&](int q) { return t0 + min(q, n - 1); };

  float ka[2][32], qb[2][32];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float* kr = base + (int64_t)tok_k(32 * i + c) * ld + inner + hf * 32;
    const float* qr = base + (int64_t)tok_q(32 * i + c) * ld + hf * 32;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 kk = *reinterpret_cast<const float4*>(kr + 4 * t);
      const float4 qq = *reinterpret_cast<const float4*>(qr + 4 * t);
      ka[i][4 * t] = kk.x; ka[i][4 * t + 1] = kk.y; ka[i][4 * t + 2] = kk.z; ka[i][4 * t + 3] = kk.w;
      qb[i][4 * t] = qq.x * scale; qb[i][4 * t + 1] = qq.y * scale; qb[i][4 * t + 2] = qq.z * scale; qb[i][4 * t + 3] = qq.w * scale;
    }
  }
  f32x16 st[2][2];                                        // S^T tiles [key tile][query tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[i][j][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 32; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) st[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[i][ks], qb[j][ks], st[i][j], 0, 0, 0);

  // V by columns in the key order the accumulators will be consumed in; issued now, consumed after the softmax
  float va[2][32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    const int key = 32 * (ks >> 4) + mfma_slot_row(ks & 15, hf);
    const float* vr = base + (int64_t)tok_k(key) * ld + 2 * inner;
    va[0][ks] = vr[c];
    va[1][ks] = vr[32 + c];
  }

  // softmax over the keys of each query (two queries per lane: tiles j = 0, 1)
  float inv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float mx = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * i + mfma_slot_row(r, hf) <= n) mx = fmaxf(mx, st[i][j][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (32 * i + mfma_slot_row(r, hf) <= n) ? __expf(st[i][j][r] - mx) : 0.f;
        st[i][j][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    inv[j] = 1.0f / sum;
  }

  f32x16 ot[2][2];                                        // O^T tiles [d tile][query tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[i][j][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 32; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ot[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][ks], st[ks >> 4][j][ks & 15], ot[i][j], 0, 0, 0);

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = 32 * j + c;
    if (q < n) {
      float* orow = out ? out + ((int64_t)b * N + t0 + q) * inner + h * DH : nullptr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {        // slots 4g..4g+3 are d = 32 i + 8 g + 4 half + (0..3)
          const float o0 = ot[i][j][4 * g] * inv[j], o1 = ot[i][j][4 * g + 1] * inv[j], o2 = ot[i][j][4 * g + 2] * inv[j],
                      o3 = ot[i][j][4 * g + 3] * inv[j];
          if (orow) *reinterpret_cast<float4*>(orow + 32 * i + 8 * g + 4 * hf) = make_float4(o0, o1, o2, o3);
          if (op.p) planes_store4(op, b * N + t0 + q, h * DH + 32 * i + 8 * g + 4 * hf, o0, o1, o2, o3);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------- attention: cls query
// one block of CLS_W wavefronts per (b,h): the (scaled) cls query attends to all N keys, padded frames masked (:120, :259-260).
// Keys are spread over all the block's lanes for the scores (one 256-byte K row per lane) and over its wavefronts for the
// weighted V sum (lane = d, coalesced rows); one wavefront per (b,h) left 3/4 of the chip idle at B*H = 256.
#ifndef MT_CLS_W
#define MT_CLS_W 16
#endif
constexpr int CLS_W = MT_CLS_W;          // (16 wavefronts: the N keys in two trips of 256 -- with 4 the kernel was a chain of 2 x 7 exposed round trips)

__device__ __forceinline__ float block_reduce(float v, float* red, int wave, int lane, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();                                   // red may still be read from the previous reduction
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < CLS_W; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// Lane mapping: 16 lanes per key (lane & 15 = which float4 of the 64-wide head), 16 keys per pass: every K / V row is one coalesced
// 256-byte access.  All reductions have a fixed order (eval stays bit-reproducible).
__global__ __launch_bounds__(CLS_W * 64) void attn_cls_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ att, const uint8_t* __restrict__ mask,
                                                                 int B, int H, int F, int n, float scale, const PlaneRef op) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // N probabilities, CLS_W reduction slots, CLS_W x 64 partial outputs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 15, grp = tid >> 4;
  constexpr int KPP = CLS_W * 4;
  const int bh = blockIdx.x, h = bh % H, b = bh / H;
  const int N = 1 + F * n, inner = H * DH, ld = 3 * inner;
  float* red = lds + N;
  float* part = lds + ((N + CLS_W + 3) & ~3);                   // 16-byte aligned (float4 slots)
  const float* base = qkv + (int64_t)b * N * ld + h * DH + sub * 4;
  float4 q = *reinterpret_cast<const float4*>(base);
  q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
  // CLS_U passes per trip, their K rows requested up front (one wave per SIMD: nothing else hides a pass's round trip)
  constexpr int CLS_U = 4;
  for (int j0 = grp; j0 < N; j0 += KPP * CLS_U) {
    float4 kk[CLS_U];
#pragma unroll
    for (int u = 0; u < CLS_U; ++u)
      kk[u] = *reinterpret_cast<const float4*>(base + (int64_t)min(j0 + u * KPP, N - 1) * ld + inner);
#pragma unroll
    for (int u = 0; u < CLS_U; ++u) {
      const int j = j0 + u * KPP;
      float a = fmaf(q.x, kk[u].x, fmaf(q.y, kk[u].y, fmaf(q.z, kk[u].z, q.w * kk[u].w)));
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o);
      if (sub == 0 && j < N) {
        if (j > 0 && mask && !mask[b * F + (j - 1) / n]) a = -FLT_MAX;
        lds[j] = a;
      }
    }
  }
  __syncthreads();
  float mx = -FLT_MAX;
  for (int j = tid; j < N; j += CLS_W * 64) mx = fmaxf(mx, lds[j]);
  mx = block_reduce(mx, red, wave, lane, true);
  float sum = 0.f;
  for (int j = tid; j < N; j += CLS_W * 64) { const float e = expf(lds[j] - mx); lds[j] = e; sum += e; }
  sum = block_reduce(sum, red, wave, lane, false);
  const float inv = 1.0f / sum;
  for (int j = tid; j < N; j += CLS_W * 64) {
    const float pj = lds[j] * inv;
    lds[j] = pj;
    if (att) att[(int64_t)bh * N + j] = pj;
  }
  __syncthreads();
  // out = sum_j p_j v_j
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = grp; j0 < N; j0 += KPP * CLS_U) {
    float4 vv[CLS_U];
#pragma unroll
    for (int u = 0; u < CLS_U; ++u)
      vv[u] = *reinterpret_cast<const float4*>(base + (int64_t)min(j0 + u * KPP, N - 1) * ld + 2 * inner);
#pragma unroll
    for (int u = 0; u < CLS_U; ++u) {
      const int j = j0 + u * KPP;
      if (j < N) {                                      // same order of additions as the one-pass loop: bit-identical outputs
        const float pj = lds[j];
        o4.x = fmaf(pj, vv[u].x, o4.x); o4.y = fmaf(pj, vv[u].y, o4.y); o4.z = fmaf(pj, vv[u].z, o4.z); o4.w = fmaf(pj, vv[u].w, o4.w);
      }
    }
  }
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    o4.x += __shfl_xor(o4.x, o); o4.y += __shfl_xor(o4.y, o); o4.z += __shfl_xor(o4.z, o); o4.w += __shfl_xor(o4.w, o);
  }
  if (lane < 16) *reinterpret_cast<float4*>(part + wave * 64 + sub * 4) = o4;
  __syncthreads();
  if (tid < 16) {
    float4 t = *reinterpret_cast<const float4*>(part + sub * 4);
#pragma unroll
    for (int w = 1; w < CLS_W; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(part + w * 64 + sub * 4);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    if (out) *reinterpret_cast<float4*>(out + (int64_t)b * N * inner + h * DH + sub * 4) = t;
    if (op.p) planes_store4(op, b * N, h * DH + sub * 4, t.x, t.y, t.z, t.w);
  }
  if (op.p && blockIdx.x == 0) planes_zero_pad(op, B * N, tid, CLS_W * 64);
}

// ---------------------------------------------------------------------------------------- classification head
// logits[b, c] = LayerNorm(x[b,0,:]) . w[c,:] + bias[c]       one wavefront per clip
__global__ __launch_bounds__(64) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ logits,
                                                     int N, int D, int C, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (int64_t)b * N * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; ss += d * d; }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  for (int c = 0; c < C; ++c) {
    float a = 0.f;
    for (int i = lane; i < D; i += 64) a += ((xr[i] - mean) * rstd * gamma[i] + beta[i]) * w[(int64_t)c * D + i];
    a = wave_sum(a);
    if (lane == 0) logits[b * C + c] = a + bias[c];
  }
}

}  // namespace

extern "C" int mt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                int rows, int dim, float eps, void* y_planes, void* stream) {
  if (!x || !gamma || !beta || (!y && !y_planes)) return fail(MT_ERR_ARG, "mt_layernorm_fwd: null pointer");
  if (dim <= 0 || (dim & 3) || dim > 1024) return fail(MT_ERR_ARG, "mt_layernorm_fwd: dim %d unsupported (need %%4==0, <=1024)", dim);
  if (y_planes && ((dim & 15) || ((uintptr_t)y_planes & 15))) return fail(MT_ERR_ARG, "mt_layernorm_fwd: plane output needs dim %% 16 == 0 and 16-byte alignment");
  if (rows <= 0) return 0;
  const int rp = (rows + 31) & ~31;
  const PlaneRef yp{reinterpret_cast<__bf16*>(y_planes), (int64_t)rp * dim, dim >> 4, rp};
  const int cover = y_planes ? rp : rows;            // the plane tensor's padding rows are written too (zeros)
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((cover + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats,
                     rows, dim, eps, yp);
  return check_launch("mt_layernorm_fwd");
}

extern "C" int mt_embed_fwd(float* x, const float* cls, const float* pos_emb, const float* size_emb,
                            const int64_t* positions, const int32_t* sizes, int B, int F, int n, int dim, int pos_rows,
                            int size_rows, int* err_flag, void* stream) {
  if (!x || !cls || !pos_emb) return fail(MT_ERR_ARG, "mt_embed_fwd: null pointer");
  if (dim & 3) return fail(MT_ERR_ARG, "mt_embed_fwd: dim %% 4 != 0");
  if (pos_rows <= 0 || (size_emb && size_rows <= 0)) return fail(MT_ERR_ARG, "mt_embed_fwd: empty embedding table");
  const int N = 1 + F * n;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((B * N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, cls, pos_emb, size_emb,
                     positions, sizes, B, N, n, F, dim, pos_rows, size_emb ? size_rows : 1, err_flag);
  return check_launch("mt_embed_fwd");
}

namespace {
template <int MODE, int NKEYS, int PPW, int STRIDE, int WPB>
int launch_patch(const float* qkv, float* out, const uint8_t* mask, const uint8_t* ident, int B, int H, int F, int n,
                 float scale, const PlaneRef& op, hipStream_t s) {
  constexpr int ROWS = MODE == 0 ? 1 + PPW * (NKEYS - 1) : NKEYS;
  const int chunks = MODE == 0 ? (n + PPW - 1) / PPW : F;
  const int64_t waves = (int64_t)B * H * chunks;
  const size_t lds = (size_t)WPB * (ROWS * STRIDE + NKEYS * 64) * sizeof(float);
  auto k = attn_patch_fwd_kernel<MODE, NKEYS, PPW, STRIDE, WPB>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_attn_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(WPB * 64), lds, s, qkv, out, mask, ident, B, H, F, n, scale, op);
  return check_launch("mt_attn_fwd(patch)");
}

template <int F, int WPB>
int launch_time_fwd(const float* qkv, float* out, const uint8_t* mask, const uint8_t* ident, int B, int H, int n, float scale,
                    const PlaneRef& op, hipStream_t s) {
  constexpr int NK = F + 1, SPP = (NK + 3) & ~3;
  const size_t lds = (size_t)WPB * ((2 * F + 1) * 68 + F * SPP) * sizeof(float);
  const int64_t waves = (int64_t)B * H * n;
  auto k = attn_time_fwd_kernel<F, WPB>;
  if (lds > 48 * 1024) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds);
    if (e != hipSuccess) return fail(MT_ERR_LAUNCH, "mt_attn_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((waves + WPB - 1) / WPB)), dim3(WPB * 64), lds, s, qkv, out, mask, ident, B, H, n, scale, op);
  return check_launch("mt_attn_fwd(time)");
}
}  // namespace

extern "C" int mt_attn_fwd(const float* qkv, float* out, float* cls_att, const uint8_t* mask, const uint8_t* ident,
                           int B, int H, int F, int n, int mode, float scale, void* out_planes, void* stream) {
  if (!qkv || (!out && !out_planes)) return fail(MT_ERR_ARG, "mt_attn_fwd: null pointer");
  if (out_planes && (mode == 2 || ((uintptr_t)out_planes & 15))) return fail(MT_ERR_ARG, "mt_attn_fwd: plane output needs mode 0 / 1 and 16-byte alignment");
  if (mode == 0 && (!mask || !ident)) return fail(MT_ERR_ARG, "mt_attn_fwd: time attention needs mask and identities_mask");
  if (n != 49) return fail(MT_ERR_UNSUPPORTED, "mt_attn_fwd: num-patches %d unsupported (49)", n);
  hipStream_t s = (hipStream_t)stream;
  const int N = 1 + F * n;
  const int rp = (B * N + 31) & ~31;
  const PlaneRef op{reinterpret_cast<__bf16*>(out_planes), (int64_t)rp * H * DH, H * DH / 16, rp};
  hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3(B * H), dim3(CLS_W * 64), (N + CLS_W + 4 + CLS_W * 64) * sizeof(float), s, qkv, out, cls_att, mask, B, H, F, n, scale, op);
  int rc = check_launch("mt_attn_fwd(cls)");
  if (rc || mode == 2) return rc;                 // mode 2: the cls query only (out row 0 of every clip; the patch rows are not written)
  if (mode == 1) {
    static const bool valu = getenv("MT_ATTN_VALU") != nullptr;     // A/B aid: the one-lane-per-query kernel
    if (valu) return launch_patch<1, 50, 1, 64, 2>(qkv, out, mask, ident, B, H, F, n, scale, op, s);
    const int64_t waves = (int64_t)B * H * F;
    hipLaunchKernelGGL(attn_space_fwd_mfma_kernel<4>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, qkv, out, B, H, F, n, scale, op);
    return check_launch("mt_attn_fwd(space, mfma)");
  }
  static const bool time_old = getenv("MT_ATTN_TIME_OLD") != nullptr;    // A/B aid: the 7-patches-per-wavefront kernel
  if (!time_old) {
    switch (F) {
      case 8: return launch_time_fwd<8, 4>(qkv, out, mask, ident, B, H, n, scale, op, s);
      case 16: return launch_time_fwd<16, 2>(qkv, out, mask, ident, B, H, n, scale, op, s);
    }
  }
  switch (F) {
    case 8: return launch_patch<0, 9, 7, 68, 4>(qkv, out, mask, ident, B, H, F, n, scale, op, s);
    case 16: return launch_patch<0, 17, 4, 68, 4>(qkv, out, mask, ident, B, H, F, n, scale, op, s);
    case 32: return launch_patch<0, 33, 2, 68, 4>(qkv, out, mask, ident, B, H, F, n, scale, op, s);
  }
  return fail(MT_ERR_UNSUPPORTED, "mt_attn_fwd: num-frames %d unsupported (8/16/32, reference train.py:101)", F);
}

extern "C" int mt_head_fwd(const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                           float* logits, int B, int N, int dim, int classes, float eps, void* stream) {
  if (!x || !gamma || !beta || !w || !bias || !logits) return fail(MT_ERR_ARG, "mt_head_fwd: null pointer");
  hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, gamma, beta, w, bias, logits, N, dim,
                     classes, eps);
  return check_launch("mt_head_fwd");
}

// ---------------------------------------------------------------------------------------- attention explainability (next-row f3)
// Reference utils.py:68-96 aggregate_attentions: per token, max over (batch*heads) of the cls attention; tokens split into
// num_frames chunks like numpy.array_split (the first N % F chunks get one extra token; the cls token sits in chunk 0);
// per chunk mean * scale_factor, softmax over the chunks.  Rows: 0 = space, 1 = time, 2 = combined (sum of the two maxima).
namespace {
__global__ __launch_bounds__(256) void attn_aggregate_kernel(const float* __restrict__ space, const float* __restrict__ time_,
                                                            float* __restrict__ out, int BH, int N, int F, float scale_factor) {
  extern __shared__ float sm[];        // tok[3][N] then chunk[3][F]
  float* tok = sm;
  float* chunk = sm + 3 * N;
  const int tid = threadIdx.x;
  for (int t = tid; t < N; t += 256) {
    float ms = -FLT_MAX, mt = -FLT_MAX;
    for (int r = 0; r < BH; ++r) {
      ms = fmaxf(ms, space[(int64_t)r * N + t]);
      mt = fmaxf(mt, time_[(int64_t)r * N + t]);
    }
    tok[t] = ms; tok[N + t] = mt; tok[2 * N + t] = ms + mt;
  }
  __syncthreads();
  const int base = N / F, extra = N % F;
  for (int i = tid; i < 3 * F; i += 256) {
    const int row = i / F, c = i % F;
    const int start = c * base + min(c, extra), len = base + (c < extra ? 1 : 0);
    float s = 0.f;
    for (int k = 0; k < len; ++k) s += tok[row * N + start + k];
    chunk[i] = s / (float)len * scale_factor;
  }
  __syncthreads();
  if (tid < 3) {
    float mx = -FLT_MAX, sum = 0.f;
    for (int c = 0; c < F; ++c) mx = fmaxf(mx, chunk[tid * F + c]);
    for (int c = 0; c < F; ++c) sum += expf(chunk[tid * F + c] - mx);
    for (int c = 0; c < F; ++c) out[tid * F + c] = expf(chunk[tid * F + c] - mx) / sum;
  }
}
}  // namespace

extern "C" int mt_attn_aggregate(const float* space_att, const float* time_att, float* out, int BH, int N, int F,
                                 float scale_factor, void* stream) {
  if (!space_att || !time_att || !out) return fail(MT_ERR_ARG, "mt_attn_aggregate: null pointer");
  if (F <= 0 || N < F) return fail(MT_ERR_ARG, "mt_attn_aggregate: bad sizes");
  hipLaunchKernelGGL(attn_aggregate_kernel, dim3(1), dim3(256), (size_t)(3 * N + 3 * F) * sizeof(float), (hipStream_t)stream,
                     space_att, time_att, out, BH, N, F, scale_factor);
  return check_launch("mt_attn_aggregate");
}
