/* libmintime_hip.so -- C ABI of the MI355X-native MINTIME hot path.
 *
 * The reference (davide-coccomini/MINTIME...) is pure Python on PyTorch and has NO FFI / plugin layer
 * (SURVEY.md §8b); its "operator API" for this path is the nn.Module surface
 *   models/efficientnet/efficientnet_pytorch/model.py:267  EfficientNet.forward
 *   models/size_invariant_timesformer.py:224               SizeInvariantTimeSformer.forward
 * and the tensor ops those call.  Each entry point below replaces the group of reference ops cited
 * next to it.  The Python host (package dir `mintime-..._amd/`) binds these with ctypes; a reference
 * maintainer would bind them the same way.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 unless stated; masks uint8; indices int32/int64)
 *   - every function enqueues on `stream` (a hipStream_t passed as void*), never synchronises,
 *     never allocates; scratch comes from the caller
 *   - return 0 on success, negative on error; mt_last_error() gives a thread-local message
 *   - re-entrant: launches go to the caller's stream on the current device; the only process-wide state are two switches
 *     (mt_gemm_set_split, mt_set_deterministic) and, with the latter on, the per-stream workspace described there
 */
#ifndef MINTIME_HIP_H
#define MINTIME_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MT_VERSION 119

int mt_version(void);
const char* mt_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Deterministic mode (reference train.py:110 `cudnn.deterministic = True`).  Off by default (MT_DETERMINISTIC=1 in the environment
 * turns it on at load time).  With the switch on no floating-point atomic is issued anywhere in the training step: partial sums go
 * through logs / split-K slabs in a per-stream workspace OWNED BY THE LIBRARY (the one exception to "never allocates": grown with
 * hipMalloc on first use, which may synchronise) and are added in a fixed order, so the same inputs and state give bit-identical
 * gradients run after run.  Slower (the bench line's `deterministic` leg has the cost).  Returns 0 / the current setting.
 * mt_det_bn_sums: BatchNorm sums in fixed order from a stored tensor x [rows][C] into stats[0 .. 2C) (fp64, +=): mode 0 = sum x,
 * sum x^2 (forward batch statistics); mode 1 = sum x, sum x * (z - mean) * invstd (backward; mean_invstd = [2][C]).  The engines call
 * it instead of the producers' fused statistics when the switch is on (round 4's engines; kept as an entry point, no longer on the path).
 * BatchNorm statistics without the second pass (EfficientNet since version 115): every entry point that takes a BatchNorm accumulator
 * (`double* stats, int slots`, or `stats` / `stats_slots` of a GEMM descriptor) accepts a NEGATIVE slot count: |slots| accumulators,
 * each held as two 64-bit INTEGER limbs |slots| * 2 * C doubles apart (stats must hold 2 * |slots| * 2 * C zeroed doubles): a block's
 * fp32 partial v adds trunc(v) to limb 0 and round((v - trunc(v)) * 2^44) to limb 1 with integer atomics.  Integer addition is
 * associative, so the sums are the same bits whatever order the blocks arrive in; mt_bn_finalize / mt_bn_bwd_finalize decode the limbs
 * when given the same negative count.
 */
int mt_set_deterministic(int on);
int mt_get_deterministic(void);
/* Frees the deterministic mode's workspaces of `stream` on the current device (synchronises that stream): call before destroying a
 * stream that ran deterministic launches.  The workspaces never grow under stream capture (the entry point fails instead). */
int mt_det_release(void* stream);
int mt_det_bn_sums(const float* x, const float* z, const float* mean_invstd, int64_t rows, int C, int mode, double* stats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction family (fp32 MFMA).  Replaces every nn.Linear / 1x1 Conv2d / their autograd:
 *   size_invariant_timesformer.py:111,144 (to_qkv, to_out), :69-74 (FeedForward), :228 (to_patch_embedding)
 *   efficientnet_pytorch/model.py:98,116,286 (_expand_conv, _project_conv, _conv_head)
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int gin, gout, off; } mt_rowmap;   /* row' = (r/gin)*gout + off + r%gin ; gin==0: identity */

enum { MT_OP_NT = 0,    /* C[M,N] = A[M,K] * B[N,K]^T      forward (torch weight layout)             */
       MT_OP_NN = 1,    /* C[M,N] = A[M,K] * B[K,N]        dgrad                                      */
       MT_OP_TN = 2 };  /* C[M,N] = A[K,M]^T * B[K,N]      wgrad (split-K + fp32 atomics, C pre-zeroed) */

enum { MT_PRO_NONE = 0, MT_PRO_BN_SWISH_GATE = 1, MT_PRO_BN_SWISH = 2, MT_PRO_AFFINE = 3,
       MT_PRO_BN_BWD = 4,
       MT_PRO_IM2COL = 5 };  /* A := act(affine(x[n, oh*s+kh-p, ow*s+kw-p, ci])) gathered from an NHWC image (conv_* fields):
                                dense k x k / strided 1x1 convolution as a GEMM with M = N*Ho*Wo, K = k*k*C (padded to %4) */  /* A := ka[c]*A + kb[c]*A2 + kc[c]  (BatchNorm backward folded into the load; ka,kb,kc = scale,shift,gate) */
enum { MT_BPRO_NONE = 0, MT_BPRO_BN_SWISH_GATE = 1, MT_BPRO_IM2COL = 2 };  /* TN only: B := swish(B*b_scale[n]+b_shift[n]) * b_gate[(k/b_hw)*N+n] */
enum { MT_EPI_STORE = 0, MT_EPI_BIAS_RES = 1, MT_EPI_GEGLU = 2, MT_EPI_STATS = 3, MT_EPI_ATOMIC = 4,
       MT_EPI_GEGLU_BWD = 5, MT_EPI_ACCUM = 6,
       /* MBConv backward around the squeeze-excite stage (autograd of efficientnet_pytorch/model.py:108-117), NN + BN_BWD only.
          The GEMM result da = dz_p . W_project is consumed in the accumulators and never stored; C2 = the depthwise conv's raw
          output z_d [M,N], u = z_d*e_scale[n] + e_shift[n], img = m / e_hw:
            SE_RED:  C[img,n] += sum_rows da * swish(u)                    (d gate, fp32 atomics, C [M/e_hw, N] zero-filled)
            ACT_BWD: C[m,n] = (da*e_gate[img,n] + e_dpool[img,n]/e_hw) * swish'(u)   and the BatchNorm-backward sums of it:
                     stats[..][0][n] += C, stats[..][1][n] += C * (z_d - e_mi[n]) * e_mi[N+n]                              */
       MT_EPI_SE_RED = 7, MT_EPI_ACT_BWD = 8 };

typedef struct {
  int op, prologue, epilogue;
  const float* A; const float* B; float* C;
  int M, N, K;                      /* GEGLU: N = full width of the weight (2*n_half)                   */
  int64_t lda, ldb, ldc;
  mt_rowmap a_map, b_map, c_map;
  const float* bias;
  const float* R; int64_t ldr;      /* residual (BIAS_RES)                                               */
  const float* scale; const float* shift; const float* gate; int hw;   /* prologue vectors               */
  float* C2; int64_t ldc2;          /* GEGLU: optional pre-activation store, row = (a_0,g_0,a_1,g_1,...) -- private
                                       to the GEGLU / GEGLU_BWD pair; GEGLU_BWD: those pre-activations (ldc2 even)        */
  double* stats; int stats_slots;   /* STATS: [slots][2][N] fp64 accumulators (sum, sum of squares)      */
  int n_half;
  int split_k;                      /* TN: number of K splits; <= 0 picks one that fills the chip        */
  const float* A2;                  /* BN_BWD prologue: second source, same layout as A                  */
  int b_prologue; const float* b_scale; const float* b_shift; const float* b_gate; int b_hw;
  int conv_H, conv_W, conv_C, conv_Ho, conv_Wo, conv_k, conv_stride, conv_pad, conv_act;   /* im2col prologues */
  int conv_src_u8;                  /* im2col source image is uint8 (raw crops, 3 channels) instead of fp32      */
  float* col_sum;                   /* GEGLU_BWD: optional [2*n_half] column sums of the stored gradient (= d bias of the
                                       Linear that produced the pre-activations), atomically accumulated; caller zero-fills */
  const void* b_planes;             /* NT only, optional: B pre-split by mt_split_planes (three bf16 planes, each [N][K] with B's
                                       ldb).  The split-operand loop then streams B by DMA instead of splitting it per tile;
                                       every other path ignores the field and reads B.  B must still be valid.             */
  int64_t b_plane_stride;           /* elements between planes */
  const float* e_scale; const float* e_shift;   /* SE_RED / ACT_BWD: BatchNorm affine of z_d [N]                            */
  const float* e_gate; const float* e_dpool;    /* ACT_BWD: squeeze-excite gate and pooled-gradient rows [M/e_hw, N]         */
  const float* e_mi; int e_hw;                  /* ACT_BWD: mean | invstd [2][N]; rows per image                             */
} mt_gemm_desc;

int mt_gemm(const mt_gemm_desc* d, void* stream);

/* Matrix pipe of the prologue-free contractions (the TimeSformer's Linear layers and their gradients).
 *   1 (default; MT_GEMM_SPLIT=0 in the environment starts at 0): split-operand fp32 -- every fp32 operand value is split exactly
 *     into three bf16 pieces and the six leading piece products are accumulated in fp32 on v_mfma_f32_32x32x16_bf16
 *     (csrc/gemm_split.hpp); error against fp64 equal to the fp32 MFMA pipe's (tests/test_gpu_gemm.py), 6/16 of its matrix time.
 *   0: v_mfma_f32_32x32x2_f32 everywhere.
 * Process-wide; returns the previous setting.  Inputs, outputs and accumulators are fp32 either way. */
int mt_gemm_set_split(int on);
int mt_gemm_get_split(void);

/* planes[p][i], p = 0..2: the exact three-piece bf16 split of src[i] (x = x0 + x1 + x2, round-to-nearest at each level) that the
 * split-operand loop computes on the fly -- done once for operands that many launches share (weights: once per optimizer step).
 * planes: 3 * n bf16 values, 16-byte aligned; plane stride n. */
int mt_split_planes(const float* src, void* planes, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Plane-operand contractions (csrc/gemm_planes.hpp): the TimeSformer's Linear layers and their gradients
 * (size_invariant_timesformer.py:60-76, 111, 144 and their autograd) with BOTH operands pre-split.
 *
 * A "plane tensor" of an fp32 matrix X [rows][cols] is its exact three-piece bf16 split X = P0 + P1 + P2 (round-to-nearest at each
 * level -- the pieces mt_gemm's split-operand loop computes on the fly) stored as
 *     planes[3][Rp/32][Cp/16][32][16] bf16,   Rp = rows rounded up to 32, Cp = cols rounded up to 16, padding = zeros,
 * i.e. 1 KB blocks of 32 rows x 16 columns; mt_planes_elems(rows, cols) = Rp * Cp is the plane stride in elements.  The producer
 * of a tensor writes its planes once (mt_layernorm_fwd / mt_layernorm_bwd_rows / mt_attn_fwd / mt_attn_bwd / the GEGLU epilogues
 * below / mt_split_planes_blk); forward, data-gradient and weight-gradient GEMMs all read the same planes by LDS-DMA, the first
 * along the columns, the other two along the rows (LDS transpose-reads).  fp32 in, fp32 out, fp32 accumulation:
 * NT / NN results are bit-identical to mt_gemm's split-operand loop on the same fp32 operands.
 * ------------------------------------------------------------------------------------------------ */
int64_t mt_planes_elems(int rows, int cols);

/* planes of src [rows][cols] (row-major, leading dimension ld). */
int mt_split_planes_blk(const float* src, int64_t ld, int rows, int cols, void* planes, void* stream);

/* Many contiguous matrices in one launch (the Linear weights, once per optimizer step).  items: device array of
 * { const float* src; void* planes; int64_t rows, cols, first; } with first = running sum of (Rp/32)*(Cp/16) over the items before;
 * total_blocks = that sum over all items. */
int mt_split_planes_blk_multi(const void* items, int count, int64_t total_blocks, void* stream);

/* planes of dz = ka * du + kb * z + kc  (du, z [rows][C] fp32, kabc = [3][C]: the BatchNorm-backward affine of mt_bn_bwd_finalize):
 * the operand a 1x1 convolution's data- AND weight-gradient GEMMs share, written once (Xception's pointwise convolutions,
 * reference models/xception.py:17-27 under autograd). */
int mt_bn_bwd_apply_planes(const float* du, const float* z, const float* kabc, void* planes, int rows, int C, void* stream);

typedef struct {
  int op;                           /* MT_OP_NT: C = A[M,K] B[N,K]^T ; MT_OP_NN: C = A[M,K] B[K,N] ; MT_OP_TN: C += A[K,M]^T B[K,N]        */
  int epilogue;                     /* NT: STORE, BIAS_RES, GEGLU, STATS ; NN: STORE, GEGLU_BWD ; TN: ATOMIC (C pre-zeroed, split-K)       */
  int M, N, K;                      /* GEGLU: N = 2 * n_half ; GEGLU_BWD: N = n_half                                                       */
  const void* a_planes;             /* plane tensor of A as stored ([M][K], TN: [K][M])                                                    */
  const void* b_planes;             /* plane tensor of B as stored (NT: [N][K], NN / TN: [K][N])                                           */
  float* C; int64_t ldc;            /* fp32 result (may be NULL for the GEGLU pair when c_planes is given)                                 */
  const float* bias;
  const float* R; int64_t ldr;      /* BIAS_RES residual                                                                                   */
  float* C2; int64_t ldc2;          /* GEGLU: optional pre-activation store (a_0,g_0,a_1,g_1,...); GEGLU_BWD: those pre-activations        */
  int n_half;
  float* col_sum;                   /* GEGLU_BWD: optional [2*n_half] column sums of the gradient (bias gradient), caller zero-fills       */
  void* c_planes;                   /* GEGLU: plane tensor of h [M][n_half]; GEGLU_BWD: of du [M][2*n_half]; NULL: fp32 output only        */
  int split_k;                      /* TN: number of K ranges; <= 0 picks one                                                               */
  void* sk_workspace;               /* NT / NN, optional: mt_gemm_planes_workspace_bytes() of device memory, 256-byte aligned, ZERO-FILLED   */
  int64_t sk_workspace_bytes;       /* once by the caller and lent to every launch of ONE stream (launches leave it zero-filled again).    */
                                    /* With it the launch runs stream-K: a persistent grid shares the (tile, k-step) list evenly instead of */
                                    /* one block per tile -- same result up to the association of a split tile's partial sums, which is     */
                                    /* fixed (bit-reproducible run to run).  NULL: one block per output tile.                               */
  double* stats; int stats_slots;   /* NT + MT_EPI_STATS (a 1x1 convolution with train-mode BatchNorm): column sums and sums of squares     */
                                    /* of the stored result into [slots][2][N] fp64 accumulators, like mt_gemm                              */
} mt_gemm_planes_desc;

/* Limit: an operand's three planes (padded rows x padded columns x 6 bytes) must stay under 4 GB -- the loop reaches them by 32-bit
 * byte offsets from one base -- or the call returns MT_ERR_UNSUPPORTED (before round 6's end it wrapped silently).                      */
int mt_gemm_planes(const mt_gemm_planes_desc* d, void* stream);

/* nn.Dropout inside the TimeSformer (size_invariant_timesformer.py:66-70 between GEGLU and net.3, :98-101 behind to_out.0), train mode
 * with attn-dropout / ff-dropout > 0 (the shipped YAML uses 0: these passes run only then).  m = keep / (1 - p), an fp32 tensor the
 * caller draws.  mt_mul_planes: planes of x * m ([rows][cols]; optionally the fp32 product too); mt_mul_add: out = r + y * m;
 * mt_geglu_bwd: du = [dh m gelu(g) | dh m a gelu'(g)] from the forward's interleaved pre-activations u = (a_0, g_0, a_1, g_1, ...),
 * as planes [rows][2 n_half] and optionally fp32 (m may be NULL). */
int mt_mul_planes(const float* x, const float* m, void* planes, float* out, int rows, int cols, void* stream);
int mt_mul_add(const float* y, const float* m, const float* r, float* out, int64_t n, void* stream);
int mt_geglu_bwd(const float* dh, const float* m, const float* u, void* du_planes, float* du, int rows, int n_half, void* stream);
int64_t mt_gemm_planes_workspace_bytes(void);
/* Persistent form of the fp32-output NT / NN plane GEMMs (MT_EPI_STORE, MT_EPI_BIAS_RES) with more tiles than resident block slots:
 * blocks_per_cu resident blocks per CU walk their XCD's share of the tile list and issue the next tile's first DMA stage before the
 * current tile's epilogue.  Same sums in the same order (bit-identical results).  0 = one block per tile (default; MT_PLANES_PERSIST
 * sets the initial value).  Process-wide like mt_gemm_set_split; returns the previous setting. */
int mt_gemm_planes_set_persist(int blocks_per_cu);

/* ------------------------------------------------------------------------------------------------
 * Size-Invariant TimeSformer forward, non-GEMM pieces
 * ------------------------------------------------------------------------------------------------ */

/* nn.LayerNorm over the last dim (size_invariant_timesformer.py:18-26).  stats (optional) [rows,2] = mean, rstd.
 * y (fp32) and / or y_planes (plane tensor of y [rows][dim], dim % 16 == 0; see mt_gemm_planes) -- at least one. */
int mt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                     int rows, int dim, float eps, void* y_planes, void* stream);

/* cls token + positional + size embeddings (:231-248), in place on x [B, 1+F*n, dim] whose rows 1.. hold the
 * patch-embedding output.  positions int64 [B,1+F*n]; sizes int32 [B,F] (NULL size_emb: enable-size-emb False). */
int mt_embed_fwd(float* x, const float* cls, const float* pos_emb, const float* size_emb,
                 const int64_t* positions, const int32_t* sizes, int B, int F, int n, int dim, int pos_rows, int size_rows,
                 int* err_flag, void* stream);
/* pos_rows / size_rows = rows of the two tables.  nn.Embedding raises on an out-of-range index; here such an index is clamped
 * into the table (no out-of-bounds access, forward or backward) and *err_flag (optional, device int32, sticky) gets bit 0
 * (positions) / bit 1 (size_embedding) set, which the host checks at its next synchronisation point. */

/* Divided attention core (:80-87 attn(), :112-141 of Attention.forward) on the QKV GEMM output
 * qkv [B, 1+F*n, 3*H*64] -> out [B, 1+F*n, H*64] (merged heads).  mode 0 = time (identity-masked), 1 = space,
 * 2 = the cls query only (out row 0 of each clip; what the LAST layer's space attention needs when only the cls token is read
 * afterwards -- the optional dead-row pruning of tsf_engine.py; mt_attn_bwd mode 2 is its adjoint: dk / dv of all keys, dq of row 0).
 * mask uint8 [B,F], ident uint8 [B,F,F]; cls_att (optional) [(B*H), 1+F*n] = the cls query's probabilities.
 * out_planes (optional, mode 0 / 1): plane tensor of out [B*(1+F*n)][H*64] (mt_gemm_planes), written by the same kernels -- the
 * operand of the out-projection and of its weight gradient; out may then be NULL. */
int mt_attn_fwd(const float* qkv, float* out, float* cls_att, const uint8_t* mask, const uint8_t* ident,
                int B, int H, int F, int n, int mode, float scale, void* out_planes, void* stream);

/* to_out: LayerNorm + Linear(dim, classes) on the cls row x[:,0] (:270-276). x [B,N,dim] -> logits [B,classes]. */
int mt_head_fwd(const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                float* logits, int B, int N, int dim, int classes, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EfficientNet-B0 forward, everything except the 1x1 convolutions (those are mt_gemm on NHWC rows).
 * Activations are NHWC fp32.  Tensors between kernels hold the RAW conv output z; consumers apply
 * swish(z*scale[c]+shift[c]) on load.  `stats` = [slots][2][C] fp64 accumulators (sum, sum of squares),
 * zeroed by the caller; pass NULL when batch statistics are not needed (eval).
 * ------------------------------------------------------------------------------------------------ */

/* _conv_stem: 3x3 stride-2 TF-SAME conv 3->32 (model.py:173,276; utils.py:248-276) + the BatchNorm batch statistics of its output
 * (stats [slots][2][32] fp64, accumulated; NULL = none).  x [N,H,W,3] fp32 or (x_is_u8) uint8 -> z [N,ceil(H/2),ceil(W/2),32];
 * w in torch layout [32,3,3,3]; W <= 512.  Streaming MFMA kernel (stem_fwd.hip): one output row per block and pass. */
int mt_stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* z, double* stats, int slots, int N, int H, int W,
                     void* stream);
/* The same kernel without padding (Conv2d(3, 32, 3, 2, 0): Xception's conv1, xception.py:135): z [N,(H-3)/2+1,(W-3)/2+1,32]. */
int mt_stem_conv_fwd_valid(const void* x, int x_is_u8, const float* w, float* z, double* stats, int slots, int N, int H, int W,
                           void* stream);

/* Depthwise conv (k 3|5, stride 1|2, TF-SAME padding; for k3 s1 that is pad 1) applied to act(zin*scale+shift);
 * act 1 = swish: EfficientNet _depthwise_conv on swish(bn(z)) (model.py:98-103);
 * act 2 = relu / 0 = none: Xception SeparableConv2d.conv1 (xception.py:21,25).  w torch layout [C,1,k,k]. */
int mt_dwconv_fwd(const float* zin, const float* scale, const float* shift, const float* w, float* zout,
                  double* stats, int slots, int N, int H, int W, int C, int k, int stride, int act, void* stream);
/* MBConv expand + depthwise in one kernel (efficientnet_pytorch/model.py:93-103: _expand_conv -> _bn0 -> swish -> _depthwise_conv):
 * y [N,H,W,cin] is the BLOCK input, we [C,cin] the expand weight, scale/shift the folded _bn0; the kernel rebuilds its 16-channel
 * chunk of the expanded tensor as y . we^T on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) while staging its tile, so the
 * expanded tensor -- the widest of the network, 6x the block input -- is neither written nor read (csrc/rc.hpp).  In train mode its
 * BatchNorm statistics come from mt_conv1x1_rows with out = NULL (statistics only).  Instances: mt_dwconv_rc_supported
 * (EfficientNet-B0 blocks 1-3: the 112^2 and 56^2 grids); act is swish.  zout / stats as mt_dwconv_fwd. */
int mt_dwconv_rc_supported(int cin, int C, int k, int stride, int H);
int mt_dwconv_fwd_rc(const float* y, const float* we, int cin, const float* scale, const float* shift, const float* w, float* zout,
                     double* stats, int slots, int N, int H, int W, int C, int k, int stride, void* stream);
/* The same convolution with the output written as a plane tensor ([N*Ho*Wo rows][C columns], mt_planes_elems; padding zeroed) and
 * nowhere else: Xception's SeparableConv2d (xception.py:17-27) feeds its depthwise output to the pointwise convolution only, which
 * runs on mt_gemm_planes -- the fp32 tensor and the mt_split_planes_blk pass over it are not needed. */
int mt_dwconv_fwd_planes(const float* zin, const float* scale, const float* shift, const float* w, void* planes, int N, int H,
                         int W, int C, int k, int stride, int act, void* stream);

/* nn.BatchNorm2d bookkeeping (model.py:51-52,62,72,86,174,202): training!=0 -> batch statistics from `stats`
 * (count = N*H*W), running-stat update with `momentum` (unbiased variance); else running statistics.
 * Emits scale = gamma*invstd, shift = beta - mean*scale, and mean_invstd [2][C] (optional, for backward). */
int mt_bn_finalize(const double* stats, int slots, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* scale, float* shift, float* mean_invstd,
                   int C, float eps, float momentum, int training, void* stream);

/* squeeze: mean_hw swish(bn(z)) (model.py:104-108), computed as `parts` pixel slices per image so that small batches still
 * fill the chip: partial[n, part, c] (already divided by HW); parts = mt_se_pool_parts(N, HW, C).  No atomics. */
int mt_se_pool_parts(int N, int HW, int C);
int mt_se_pool_fwd(const float* z, const float* scale, const float* shift, float* partial, int N, int HW, int C, int parts,
                   void* stream);

/* excite: pooled = sum of the slices (written to `pooled` [N,C] when not NULL: kept for backward);
 * gate = sigmoid(_se_expand(swish(_se_reduce(pooled)))) (model.py:109-112). w1 [CS,C], w2 [C,CS];
 * hidden [N,CS] (required: the hand-over between the kernel pair) receives the pre-activation of the squeeze layer, which backward keeps. */
int mt_se_gate_fwd(const float* partial, int parts, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* pooled, float* gate, float* hidden, int N, int C, int CS, void* stream);

/* y = act(z*scale+shift) * rowscale[row / rows_per_group] (+ res): _bn2 + drop_connect + identity skip
 * (model.py:117-127, utils.py:129-154; act=0) and head _bn1+swish (model.py:286; act=1). rowscale may be NULL. */
int mt_bn_act_fwd(const float* z, const float* scale, const float* shift, const float* res, float* y,
                  int64_t rows, int C, int act, const float* rowscale, int rows_per_group, void* stream);

/* Attention explainability post-process (utils.py:68-96 aggregate_attentions): out [3][F] = softmax over frame chunks of
 * scale_factor * mean over the chunk's tokens of max over (batch*heads) of the cls attention; rows space / time / combined.
 * space_att, time_att: [(B*H), N] as returned by mt_attn_fwd's cls_att. */
int mt_attn_aggregate(const float* space_att, const float* time_att, float* out, int BH, int N, int F,
                      float scale_factor, void* stream);

/* Input-sequence builder (next-row f1): the side inputs of SizeInvariantTimeSformer.forward for a batch of clips, built in HBM
 * from the compact description a loader produces.  Replaces the list-building code of deepfakes_dataset.py:259-287 (size
 * buckets via SIZE_EMB_DICT :30-31, padded slots), :315-321 (identities_mask), :324-329 (temporal positions) and
 * predict.py:285-309,335-347.  Inputs (device, int32): slots[B][max_identities] = slots per identity in slot order (0 = unused),
 * valid[B][max_identities] = faces actually present (<= slots; the rest of the identity's slots are padding), frames[B][F] =
 * video-frame number per slot (ignored at padded slots: they take the largest frame number seen so far, :271-275),
 * ratio[B][F] = int(face_area*100/video_area) per slot (0..100).  Outputs: mask [B][F] u8, identities_mask [B][F][F] u8,
 * size_embedding [B][F] int32 (bucket 1..20, 0 = padding), positions [B][1+F*num_patches] int64.
 * mask_mode 0 = the dataset's behaviour (mask all ones: its padding test at :281 runs after the list was already extended),
 * mask_mode 1 = predict.py:304 (padded slots masked out). */
int mt_build_clip_inputs(const int* slots, const int* valid, const int* frames, const int* ratio, unsigned char* mask,
                         unsigned char* identities_mask, int* size_embedding, int64_t* positions, int B, int F,
                         int num_patches, int max_identities, int mask_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Size-Invariant TimeSformer backward, non-GEMM pieces.  The reference derives these through torch autograd
 * from the same source lines as the forward entry points; here they are explicit adjoint kernels.
 * ------------------------------------------------------------------------------------------------ */

/* LayerNorm adjoint. dx (+)= LN'(dy) (accumulate!=0 adds into dx: the residual stream's gradient);
 * dgamma/dbeta are accumulated atomically (caller zero-fills). stats = [rows,2] mean,rstd saved by the forward.
 * dx_colsum (optional, [dim], atomically accumulated): column sums of the UPDATED dx rows -- the bias gradient of the Linear
 * whose output gradient dx is next (to_out.0.bias / net.3.bias / to_patch_embedding.bias), which the reference gets from
 * autograd's sum over rows; rows with row % skip_period == 0 are left out when skip_period > 0 (the cls rows, which the patch
 * embedding does not produce, :231-232).
 * dx_in (optional): with accumulate != 0 the sum is dx = LN'(dy) + dx_in instead of in place (NULL = dx itself), so that the
 * previous value of the residual-stream gradient stays readable (deferred weight gradients, tsf_backward.py). */
int mt_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, float* dx,
                     float* dgamma, float* dbeta, int rows, int dim, int accumulate, float* dx_colsum, int skip_period,
                     const float* dx_in, void* stream);
/* The same LayerNorm backward split by queue: mt_layernorm_bwd_rows writes only dx = LN'(dy) + dx_in (few registers, no LDS: it
 * co-resides with the weight-gradient GEMMs instead of waiting for their blocks to end) and mt_layernorm_bwd_cols accumulates the
 * parameter gradients (dgamma, dbeta, and dx_colsum = column sums of dx_new, rows with r % skip_period == 0 left out when
 * skip_period > 0) -- meant for the weight-gradient stream. */
int mt_layernorm_bwd_rows(const float* dy, const float* x, const float* stats, const float* gamma, float* dx, const float* dx_in,
                          int rows, int dim, void* dx_planes, void* stream);   /* dx_planes (optional): plane tensor of the new dx */
int mt_layernorm_bwd_cols(const float* dy, const float* x, const float* stats, const float* dx_new, float* dgamma, float* dbeta,
                          float* dx_colsum, int skip_period, int rows, int dim, void* stream);

/* (autograd of PreNorm's nn.LayerNorm, size_invariant_timesformer.py:18-26, as above.)  The rows kernel with the parameter-gradient
 * sums folded in: besides dx (and its planes) every block stores one row of
 * partials[blocks][3][dim] = its rows' column sums of dy * xhat, dy and dx_new (rows with r % skip_period == 0 left out of the
 * third when skip_period > 0); blocks = mt_layernorm_bwd_rows_blocks(rows).  mt_layernorm_bwd_cols_reduce adds the block rows,
 * in block order (a fixed summation order: deterministic as it is), into dgamma, dbeta and dx_colsum (NULL = skipped) -- 6 MB read at
 * B = 32 instead of the 75 MB mt_layernorm_bwd_cols re-reads.  dim <= 512. */
int mt_layernorm_bwd_rows_blocks(int rows);
int mt_layernorm_bwd_rows_sums(const float* dy, const float* x, const float* stats, const float* gamma, float* dx, const float* dx_in,
                               int rows, int dim, void* dx_planes, float* partials, int skip_period, void* stream);
int mt_layernorm_bwd_cols_reduce(const float* partials, int blocks, int dim, float* dgamma, float* dbeta, float* dx_colsum,
                                 void* stream);

/* out[n] += sum_m A[map(m)*lda + n]   (bias gradients). */
int mt_colsum(const float* A, int64_t lda, mt_rowmap map, int M, int N, float* out, void* stream);

/* adjoint of mt_head_fwd; writes dx[:,0,:] (dx must be zero elsewhere), accumulates the four parameter grads. */
int mt_head_bwd(const float* dlogits, const float* x, const float* gamma, const float* beta, const float* w,
                float* dx, float* dgamma, float* dbeta, float* dw, float* dbias, int B, int N, int dim, int classes,
                float eps, void* stream);

/* adjoint of mt_embed_fwd: scatter-adds into dcls [dim], dpos_emb, dsize_emb (zero-filled by the caller). */
int mt_embed_bwd(const float* dx, float* dcls, float* dpos_emb, float* dsize_emb, const int64_t* positions,
                 const int32_t* sizes, int B, int F, int n, int dim, int pos_rows, int size_rows, void* stream);

/* adjoint of mt_attn_fwd: dout [B,N,H*64] -> dqkv [B,N,3*H*64] (fully written). Probabilities are recomputed from qkv.
 * dqkv_planes (optional, mode 0 / 1): the result as a plane tensor [B*N][3*H*64] (operand of the QKV layer's data and weight
 * gradients).  dqkv is then working memory: its patch rows are left holding the cls query's contribution only. */
int mt_attn_bwd(const float* qkv, const float* dout, float* dqkv, const uint8_t* mask, const uint8_t* ident,
                int B, int H, int F, int n, int mode, float scale, void* dqkv_planes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EfficientNet-B0 backward, everything except the 1x1-convolution dgrad/wgrad (mt_gemm with the BN_BWD prologue).
 * BatchNorm backward is three steps: a producer accumulates sums[2][C] = (sum du, sum du*xhat) into fp64 slots;
 * mt_bn_bwd_finalize turns them into kabc[3][C] so that dz = ka*du + kb*z + kc; consumers apply that on load.
 * ------------------------------------------------------------------------------------------------ */

/* du = (din [*gate[n,c] + dpool[n,c]/hw] [*rowscale[n]]) * (act ? swish'(z*scale+shift) : 1), written to dout when
 * non-NULL (in place allowed), plus the BN sums.  Adjoint of swish/_bn* + SE scaling + drop_connect (model.py:100-127). */
int mt_bn_act_bwd(const float* din, const float* z, const float* scale, const float* shift,
                  const float* mean_invstd, const float* gate, const float* dpool, const float* rowscale,
                  float* dout, double* stats, int slots, int64_t rows, int C, int hw, int act, void* stream);

/* training != 0: batch-statistics BN adjoint; else running-statistics (kb = kc = 0). dgamma/dbeta accumulate. */
int mt_bn_bwd_finalize(const double* stats, int slots, double count, const float* gamma, const float* mean_invstd,
                       float* kabc, float* dgamma, float* dbeta, int C, int training, void* stream);

/* Squeeze-excite adjoint (model.py:104-113): from da = d(gated tensor) computes dgate, the two 1x1-conv weight/bias
 * grads (accumulated) and dpooled [N,C] (the pooling path's contribution to d(activated tensor)).
 * parts & 1: the reduction and the per-image adjoint (dgate, dpre2, dhid, dpooled -- what the data path waits for);
 * parts & 2: the weight / bias gradients from dpre2, dhid (independent of the data path: may run on another stream);
 * parts & 4: the per-image adjoint only (dgate given, e.g. reduced by mt_gemm's MT_EPI_SE_RED; da / z / scale / shift unused).
 * scratch (parts & 5): mt_se_scratch_floats(N, C, CS) floats of workspace (per-slab partial sums of the squeeze gradient). */
int mt_se_scratch_floats(int N, int C, int CS);
int mt_se_bwd(const float* da, const float* z, const float* scale, const float* shift, const float* gate,
              const float* hidden, const float* pooled, const float* w1, const float* w2, float* dgate,
              float* dpre2, float* dhid, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int N,
              int HW, int C, int CS, int parts, float* scratch, void* stream);

/* Depthwise-conv adjoint: dz = ka*du+kb*z+kc (virtual, output side).  parts&1: dw (accumulated, torch layout
 * [C,1,k,k]); parts&2: du_in = d(input pre-activation) = dgrad * swish'(bn_in(zin)), plus the input-side BN sums.
 * The two parts are independent kernels (the host runs the weight part on a second stream). */
int mt_dwconv_bwd(const float* du, const float* z, const float* kabc, const float* w, const float* zin,
                  const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in,
                  double* stats_in, int slots, float* dw, int N, int H, int W, int C, int k, int stride,
                  int parts, int act, const float* res_pre, const float* res_post, void* stream);
/* The adjoint of mt_dwconv_fwd_rc: `zin` of mt_dwconv_bwd is replaced by the block input y [N,H,W,cin] and the expand weight we
 * [C,cin]; the depthwise input's pre-activation is rebuilt in the kernel (csrc/rc.hpp).  parts = 1 (dw) or 2 (du_in + sums); act is
 * swish.  du_in is the gradient w.r.t. the expand convolution's BatchNorm output -- what mt_conv1x1_bwd_fused consumes. */
int mt_dwconv_bwd_rc(const float* du, const float* z, const float* kabc, const float* w, const float* y, const float* we, int cin,
                     const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in, double* stats_in,
                     int slots, float* dw, int N, int H, int W, int C, int k, int stride, int parts, void* stream);
/* act as in mt_dwconv_fwd; stats_in/mean_invstd_in may both be NULL.  du_in = (dgrad + res_pre) * act'(.) + res_post: gradients of
 * other consumers of the activated (res_pre) or raw (res_post) input tensor (Xception skip paths), either may be NULL. */
/* The same with res_pre / res_post given at half resolution, [N, ceil(H/2), ceil(W/2), C]: the gradient that a stride-2 1x1
 * convolution (Xception's skip path, xception.py:36-40) sends to the even (ih, iw) positions of its input; stride must be 1. */
int mt_dwconv_bwd_res2(const float* du, const float* z, const float* kabc, const float* w, const float* zin,
                       const float* scale_in, const float* shift_in, const float* mean_invstd_in, float* du_in,
                       double* stats_in, int slots, float* dw, int N, int H, int W, int C, int k, int stride,
                       int parts, int act, const float* res_pre, const float* res_post, void* stream);

/* _conv_stem weight gradient (accumulated, torch layout [32,3,3,3]); x [N,H,W,3]. */
int mt_stem_conv_wgrad(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N,
                       int H, int W, void* stream);   /* x fp32 or (x_is_u8) uint8 */
/* ... of the unpadded stride-2 3x3 convolution (Xception's conv1); du, z [N,(H-3)/2+1,(W-3)/2+1,32]. */
int mt_stem_conv_wgrad_valid(const float* du, const float* z, const float* kabc, const void* x, int x_is_u8, float* dw, int N,
                             int H, int W, void* stream);

/* Weight gradient of a 1x1 convolution with few channels and very many rows (MBConv expand / project convs of stages 1-4,
 * efficientnet_pytorch/model.py:93-104 under train.py:371):  dw[Cout,Cin] += sum_r (ka*du+kb*z+kc)[r,Cout] * a[r,Cin] with
 * a = x, or (gate != NULL) swish(sc*x+sh)*gate[r/hw].  The whole (32-padded) result stays in MFMA accumulators while the rows
 * stream through; mt_conv1x1_wgrad_supported says whether a shape has an instance (otherwise use mt_gemm, MT_OP_TN). */
int mt_conv1x1_wgrad_supported(int Cout, int Cin);
/* Data AND weight gradient of an expand conv (z = x . W^T, W [Cout, Cin], Cin <= 32) in ONE streaming pass over du (skinny_bwd.hip):
 * with dz = ka*du + kb*z + kc (BatchNorm backward of _bn0): dx[rows, Cin] = dz . W (+ res), dw[Cout, Cin] += dz^T . x.  z is not an
 * argument: it is linear in x, so its share is folded into two Cin x Cin matrices (W^T diag(kb) W and x^T x) inside the kernel.
 * Replaces mt_conv1x1_rows (mode 2) + mt_conv1x1_wgrad on the same tensors: 1 instead of 4 passes over the expanded activations. */
int mt_conv1x1_bwd_fused_supported(int Cout, int Cin);
/* Squeeze-excite stage of the MBConv reverse walk for the early blocks without the project conv's data gradient in memory
 * (skinny_se.hip): da = (ka*du_p + kb*z_p + kc) . W_p is rebuilt from the NARROW gradient inside two streaming passes over z_d:
 *   mode 0: dgate[rows / hw, C] += sum_rows da * swish(z_d*scale + shift)                     (zero-filled by the caller)
 *   mode 1: du_d = (da*gate + dpooled/hw) * swish'(z_d*scale + shift), plus its BatchNorm-backward sums (stats like MT_EPI_STATS,
 *           second sum = du_d * (z_d - mean) * invstd)
 * Replaces mt_conv1x1_rows (mode 2) + the reduction of mt_se_bwd + mt_bn_act_bwd: 3 instead of 6 passes over the expanded tensor.
 * w_p [Co, C] (Co <= 32; C = 32 / 96 / 144), hw % 64 == 0, rows % hw == 0. */
int mt_se_stage_fused_supported(int Co, int C, int hw);
int mt_se_stage_fused(const float* du_p, const float* z_p, const float* kabc_p, const float* w_p, const float* z_d,
                      const float* scale_d, const float* shift_d, int mode, float* dgate, const float* gate, const float* dpooled,
                      const float* mean_invstd_d, float* du_d, double* stats, int slots, int64_t rows, int Co, int C, int hw,
                      void* stream);
int mt_conv1x1_bwd_fused(const float* du, const float* kabc, const float* x, const float* w, const float* res, float* dx,
                         float* dw, int64_t rows, int Cout, int Cin, void* stream);
int mt_conv1x1_wgrad(const float* du, const float* z, const float* kabc, const float* x, const float* sc, const float* sh,
                     const float* gate, int hw, float* dw, int64_t rows, int Cout, int Cin, void* stream);
/* The same weight gradient for the WIDE project convolutions of the late stages (65-320 output x 224-2048 expanded channels, the gate
 * form only: sc, sh, gate required; hw >= 32): the result is cut into 128-column slabs, one block per (slab, row range), load / transform
 * wavefronts feeding MFMA wavefronts through double-buffered LDS (wide_wgrad.hip).  Replaces mt_gemm (MT_OP_TN with both operand
 * prologues), which re-applied the transforms per fragment read. */
int mt_conv1x1_wgrad_wide_supported(int Cout, int Cin);
int mt_conv1x1_wgrad_wide(const float* du, const float* z, const float* kabc, const float* x, const float* sc, const float* sh,
                          const float* gate, int hw, float* dw, int64_t rows, int Cout, int Cin, void* stream);

/* 1x1 convolution with few channels and very many rows as a streaming kernel (MBConv expand / project convs of stages 1-4 and
 * their data gradients, efficientnet_pytorch/model.py:93-118): out[rows,Cout] = a[rows,Cin] . W^T (+ res), with
 *   amode 0: a = x;   1: a = swish(c0*x + c1) * c2[row / hw]  (c2 = SE gate [rows/hw, Cin]);   2: a = c0*x + c1*x2 + c2 (BatchNorm backward).
 * w is [Cout, ldw] (w_transposed = 0) or the forward weight [Cin, ldw] used transposed (w_transposed = 1, data gradient).
 * stats (optional) receives the BatchNorm sums of `out` like MT_EPI_STATS; out may be NULL when stats is given (statistics only).  mt_conv1x1_rows_supported tells whether this kernel is
 * the measured better choice for a channel pair and mode (otherwise use mt_gemm). */
int mt_conv1x1_rows_supported(int Cin, int Cout, int amode);
int mt_conv1x1_rows_instance(int Cin, int Cout);      /* an instance exists (whether or not it is the better choice) */
int mt_conv1x1_rows(const float* x, const float* x2, const float* w, int ldw, int w_transposed, const float* c0, const float* c1,
                    const float* c2, int hw, int amode, const float* res, float* out, double* stats, int slots, int64_t rows,
                    int Cin, int Cout, void* stream);

/* dst_i [cols_i, rows_i] = src_i [rows_i, cols_i]^T for `count` matrices in one launch (the TimeSformer's transposed Linear weights,
 * tsf_engine.py: data gradients in the k-contiguous NT form; replaces 54 torch copy kernels per step).  items: device array of
 * { const float* src; float* dst; int64 rows; int64 cols; int64 tile0 } with tile0 = running count of 32 x 32 tiles. */
int mt_transpose_multi(const void* items, int count, int64_t total_tiles, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Training-step ends (next-row f4).
 * mt_bce_logits: torch.nn.BCEWithLogitsLoss(pos_weight)(logits, labels) with mean reduction (train.py:261,367-368) and its
 *   gradient in one launch: loss[1], dlogits[n] (may be NULL) = d loss / d logits.
 * mt_sgd_multi: torch.optim.SGD(lr, weight_decay) (train.py:186,378; no momentum) over many tensors in one launch:
 *   p -= lr * (g + weight_decay * p).  items = device array of {float* p; const float* g; int64 n; int64 block0} sorted by
 *   block0 = index of the tensor's first 4096-element block; total_blocks = sum over tensors of ceil(n / 4096).
 * ------------------------------------------------------------------------------------------------ */
int mt_bce_logits(const float* logits, const float* labels, float pos_weight, float* loss, float* dlogits, int n, void* stream);
int mt_sgd_multi(const void* items, int count, int64_t total_blocks, float lr, float weight_decay, void* stream);
/* mt_adam_multi: torch.optim.Adam (decoupled_weight_decay = 0: L2 term added to the gradient) / torch.optim.AdamW
 *   (decoupled_weight_decay = 1: p *= 1 - lr*wd first) over many tensors in one launch (train.py:187-190, :378; amsgrad off).
 *   items = device array of {float* p; const float* g; float* m; float* v; int64 n; int64 block0} sorted by block0 (as above).
 *   step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) for the step count t of this call. */
int mt_adam_multi(const void* items, int count, int64_t total_blocks, float lr, float weight_decay, double beta1, double beta2,
                  float eps, float step_size, float bias_correction2_sqrt, int decoupled_weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Xception (config 5 extractor, reference models/xception.py).  Dense convolutions are mt_gemm with the IM2COL
 * prologue; separable convolutions are mt_dwconv_* (act 0/2) + mt_gemm; the rest:
 * ------------------------------------------------------------------------------------------------ */

/* torch conv weight [Co,Ci,k,k] -> GEMM operand out[r][(kh,kw,c)] with row pitch ld (zero padded, ld %% 4 == 0).
 * transpose 0: rows = Co (forward / wgrad layout); 1: rows = Ci with the kernel flipped (data-gradient convolution). */
int mt_conv_weight_pack(const float* w, float* out, int Co, int Ci, int k, int ld, int transpose, void* stream);
/* dw[co][ci][kh][kw] += dwp[co][(kh,kw,ci)] */
int mt_conv_weight_unpack_grad(const float* dwp, float* dw, int Co, int Ci, int k, int ld, void* stream);

/* Block tail (xception.py:64-79): y = MaxPool2d(3,2,1)(z*scale+shift) + (zs*scale_s+shift_s); z [N,H,W,C], zs,y [N,Ho,Wo,C]. */
int mt_maxpool_add_fwd(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                       const float* shift_s, float* y, int N, int H, int W, int C, void* stream);
/* du (pre-zeroed, [N,H,W,C]) += dy routed to each window's arg-max of z*scale+shift. */
int mt_maxpool_bwd(const float* dy, const float* z, const float* scale, const float* shift, float* du, int N, int H,
                   int W, int C, void* stream);
/* The block tail with the adjoint's routing recorded (training): also writes arg [N,Ho,Wo,C] uint8 = the window position kh*3+kw of each
 * window's arg-max (first maximum in row-major order, like torch; 255 = none) and zmax = the raw z there.  The BatchNorm-backward sums of
 * the pooled unit are then sums over the POOLED tensors (S1 = sum dy, S2 = sum dy * xhat(zmax): mt_bn_act_bwd on dy / zmax), and the
 * routed gradient is a gather with one writer per element -- no zero fill, no atomics, bit-identical run to run:
 *   mt_maxpool_bwd_arg: du [N,H,W,C] = dy routed (every element written);
 *   mt_maxpool_bn_bwd_apply_planes: dz = ka*du + kb*z + kc with du routed on the fly, written as a plane tensor (mt_gemm_planes operand);
 *     du never exists in memory (replaces zero fill + mt_maxpool_bwd + the full-resolution sums pass + mt_bn_bwd_apply_planes). */
int mt_maxpool_add_fwd_arg(const float* z, const float* scale, const float* shift, const float* zs, const float* scale_s,
                           const float* shift_s, float* y, uint8_t* arg, float* zmax, int N, int H, int W, int C, void* stream);
int mt_maxpool_bwd_arg(const float* dy, const uint8_t* arg, float* du, int N, int H, int W, int C, void* stream);
int mt_maxpool_bn_bwd_apply_planes(const float* dy, const uint8_t* arg, const float* z, const float* kabc, void* planes, int N, int H,
                                   int W, int C, void* stream);
/* dz = ka*du + kb*z + kc materialised (dense conv2's data gradient is itself an im2col GEMM over dz). */
int mt_bn_bwd_apply(const float* du, const float* z, const float* kabc, float* dz, int64_t rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Operand-plane producers for EfficientNet-B0's late stages (model.py:89-128; csrc/effnet_planes.hip): the MBConv 1x1 convolutions of
 * the 14 x 14 / 7 x 7 stages and the head on mt_gemm_planes.
 * mt_bn_act_fwd_planes: mt_bn_act_fwd (block output y = act(z * scale + shift) [* rowscale[row / rows_per_group]] [+ res]) that also
 *   writes y as a plane tensor -- the next expand convolution's operand (forward and weight gradient).
 * mt_bn_swish_gate_planes: the project convolution's operand swish(z * scale[c] + shift[c]) * gate[(row / hw) * C + c] (model.py:104-116:
 *   _bn1, swish, squeeze-excite gate) as a plane tensor.
 * ------------------------------------------------------------------------------------------------ */
int mt_bn_act_fwd_planes(const float* z, const float* scale, const float* shift, const float* res, float* y, int rows, int C, int act,
                         const float* rowscale, int rows_per_group, void* y_planes, void* stream);
int mt_bn_swish_gate_planes(const float* z, const float* scale, const float* shift, const float* gate, int hw, void* planes, int rows,
                            int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Launch plans (the caller side of the step: reference train.py:332-378, the Python loop that issues every op).
 * A plan holds the SEQUENCE of entry-point calls of one phase (EfficientNet forward, TimeSformer forward, TimeSformer backward,
 * EfficientNet backward) so that the host issues a phase with one call instead of hundreds through its FFI.
 *   mt_plan_record_begin .. mt_plan_record_end   every call the CALLING THREAD makes in between to an entry point of this header
 *       that takes a `stream` (and to mt_plan_fork) is executed as usual AND appended to the plan with its argument values:
 *       pointers, shapes, scalars, descriptors (copied), the stream.
 *   mt_plan_run     re-issues the recorded calls in order, on the streams they were recorded on.  The caller guarantees what it
 *       guarantees for the calls themselves -- every buffer alive at the recorded address -- and that the recording thread's
 *       stream order is a valid order (cross-stream dependencies are part of the plan through mt_plan_fork).
 *       probe_mask: bit t set = bracket every call tagged t with timing events on its own stream (mt_plan_probe_read).
 *   mt_plan_fork    `to_stream` waits for everything enqueued so far on `from_stream` (event record + wait); recorded when recording.
 *   mt_plan_tag     tags the NEXT recorded call of this thread with tag 1..31 and a work figure (bytes or flops).
 *   mt_plan_probe_read   waits for the probe events of `tag`, returns launches / summed milliseconds / summed work, and clears them.
 *   mt_memset_async / mt_copy_async   hipMemsetAsync / device-to-device hipMemcpyAsync as entry points, so a plan can hold them.
 * A plan is bound to the device that was current at mt_plan_create.  Not thread-safe per plan; different plans are independent.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mt_plan mt_plan;
int mt_plan_create(mt_plan** out);
int mt_plan_destroy(mt_plan* plan);
int mt_plan_record_begin(mt_plan* plan);
int mt_plan_record_end(mt_plan* plan);
int mt_plan_size(const mt_plan* plan);
int mt_plan_tag(int tag, double work);
int mt_plan_fork(void* from_stream, void* to_stream);
int mt_plan_run(mt_plan* plan, unsigned probe_mask);
int mt_plan_probe_read(mt_plan* plan, int tag, int* launches, double* ms, double* work);
int mt_memset_async(void* p, int value, int64_t bytes, void* stream);
int mt_copy_async(void* dst, const void* src, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
