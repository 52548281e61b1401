/* acx.h -- C ABI of libacx, the MI355X (gfx950) hot-path library for AnomalyCLIP.
 *
 * The reference (lucazanella/AnomalyCLIP) is pure Python over stock PyTorch ops and has NO FFI of
 * its own; this header therefore defines the boundary SURVEY.md section 8(b) recommends: plain C
 * types, raw device pointers, caller-owned memory, one HIP stream per call.  Each entry point
 * names the reference code it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - every function returns 0 on success, a negative ACX_E_* code otherwise; the text of the
 *     last error is available from acx_last_error(ctx) (ctx may be NULL: thread-local slot).
 *   - never aborts, never throws across the ABI; asynchronous kernel faults surface at the
 *     caller's next synchronisation.
 *   - ALL buffers (inputs, outputs, weights, workspace) are owned by the caller (PyTorch);
 *     the library never allocates or frees device memory and retains no pointer after a call.
 *   - tensors are row-major; leading dimensions are in ELEMENTS.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 */
#ifndef ACX_H_
#define ACX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACX_VERSION 100 /* 0.1.0 */

enum {
  ACX_OK = 0,
  ACX_E_BADARG = -1,   /* null pointer, bad shape, misaligned leading dimension */
  ACX_E_UNSUPPORTED = -2, /* dtype / geometry this build has no kernel for */
  ACX_E_HIP = -3,      /* a HIP runtime call failed (text in acx_last_error) */
  ACX_E_WORKSPACE = -4, /* workspace too small */
  ACX_E_RCCL = -5      /* RCCL could not be opened, or an RCCL call failed (acx_comm_* / acx_allreduce / acx_allgather) */
};

enum { ACX_F32 = 0, ACX_BF16 = 1,                    /* storage dtypes */
       ACX_BF16X3 = 2,    /* OUTPUT only (acx_layernorm y_dtype, acx_gemm c_dtype of a pairs = 6 product): the value as three dense
                             bf16 planes hi | mid | lo [rows, ld] each, plane p at base + p * rows * ld elements -- the producer
                             writes the A operand of a pairs = 6 product directly */
       ACX_BF16X3P = 3 }; /* the same three planes in K-PANEL layout: a plane is [ld / 32][rows][32] -- the 32 columns of K-step kp
                             of all rows are contiguous (element (r, c) at ((c / 32) * rows + r) * 32 + c % 32; ld % 32 == 0,
                             dense).  The plane-reuse kernel then stages a 256-row x 32-column unit from ONE contiguous 16 KB block
                             instead of 256 half cache lines (acx_gemm_desc.panels) */
enum { ACX_F16 = 5,        /* acx_gemm_desc.a_dtype with pairs = 3: A and W are TWO fp16 planes each (acx_split_f16x2: x = hi + lo,
                              hi = fp16(x), lo = fp16(x - hi)) -- the ACX_PREC_F16X3 arithmetic: products (lo,hi) (hi,lo) (hi,hi) on
                              v_mfma_f32_32x32x16_f16, an error of ~2^-22 of sum |a||w| for operands inside fp16's range */
       ACX_F16X2P = 6 };   /* the two fp16 planes in K-panel layout (plane = [ld / 32][rows][32], plane p at base + p * rows * ld
                              elements): output of acx_layernorm / acx_vit_patches / an ACX_F16 product, input of the next one */
enum { ACX_BF16X2P = 4 };  /* acx_layernorm y_dtype / acx_vit_patches out_dtype: the ACX_BF16X3P image with the hi and mid planes written
                              only (the lo plane's bytes are left as they are): the A operand of a pairs = 3 product, which never reads
                              the lo plane.  A producer that does not special-case it writes all three planes. */
enum { ACX_PREC_F32 = 0, ACX_PREC_BF16 = 1,          /* MFMA arithmetic: exact f32 (v_mfma_f32_32x32x2_f32) or bf16 in / f32 acc */
       ACX_PREC_F32X3 = 3,    /* the same drivers and plane layouts with the THREE leading products only (acx_gemm_desc.pairs = 3:
                                 (mid, hi) (hi, mid) (hi, hi)): sixteen significant bits per operand -- an error of ~1e-5 of sum |a||w| per
                                 product, between TF32 and f32 -- at about twice the GEMM rate of ACX_PREC_F32X6.  NOT an f32-accurate
                                 path: opt-in (precision "bf16x3"); the attention keeps its six-product form */
       ACX_PREC_F16X3 = 4,    /* the same drivers with TWO fp16 planes per operand and the three products (lo,hi) (hi,lo) (hi,hi):
                                 the fp16 pair holds the f32 value to 2^-24 (lo normal), every product is exact in f32, the dropped
                                 (lo,lo) term is <= 2^-22: f32-MFMA-level results at the three-product rate, for operands inside fp16's
                                 range (|x| < 65504; weights are pre-scaled by a power of two per matrix).  Opt-in (precision "f16x3") */
       ACX_PREC_F32X6 = 2 };  /* acx_vit_encode / acx_transformer_forward only: f32 everywhere, the four large GEMMs of a layer as
                                 f32-accurate bf16 x 6 products (acx_gemm_desc.pairs); the *_w_bf16 weight fields then hold THREE
                                 planes each (acx_split_bf16x3 of the f32 weight) */
#define ACX_F16X3_WSCALE 1024.0f   /* ACX_PREC_F16X3: the weights' fp16 planes hold 2^10 * w (acx_split_f16x2: lifts CLIP-sized weights into
                                      fp16's normal range so that the lo plane keeps its 11 bits; |w| < 63.9); the drivers' products undo it */
enum { ACX_ACT_NONE = 0, ACX_ACT_QUICKGELU = 1, ACX_ACT_LEAKYRELU = 2 };
enum { ACX_AMAP_IDENTITY = 0, ACX_AMAP_CONV3X3 = 1, ACX_AMAP_TESTTILE = 2, ACX_AMAP_TILETABLE = 3 };
enum { ACX_NORM_LAYER = 0, ACX_NORM_CHAN = 1 };     /* nn.LayerNorm vs axial_attention ChanLayerNorm (eps added to std) */

typedef struct acx_ctx acx_ctx;

int acx_version(void);
int acx_create(acx_ctx** out, int device);
void acx_destroy(acx_ctx* ctx);
const char* acx_last_error(acx_ctx* ctx);

/* Per-context tuning options (dispatch thresholds; results never depend on them beyond summation order).
 *   ACX_OPT_RING_MIN_TILES  bf16 GEMMs with at least this many 256x256 output tiles take the persistent
 *                           256x256 LDS-DMA kernel instead of the 128x128 one (default 512). */
/*   ACX_OPT_SK_MAX_M        f32 acx_gemm problems with at most this many rows take the few-row kernel (32x32 tiles,
 *                           K split over the waves, no split-K reduction launch; default 320; 0 disables it except
 *                           for the a_act / gelu_grad_of fusions, which always need it). */
/*   ACX_OPT_TN_P256_MIN_ROWS acx_gemm_tn problems with at least this many rows (and >= 8 output tiles of 256 x 256, no
 *                           b_sub) take the 256 x 256 LDS-DMA kernel (default 4096). */
/*   ACX_OPT_X6_CUS          the pairs = 6 kernels (acx_gemm, acx_gemm_tn_x6) are persistent, ONE workgroup per CU with the whole
 *                           register file and 144 KB of LDS: nothing else runs on a CU they hold.  value > 0 caps their grid
 *                           (and the K split is chosen for that many workgroups) so that the remaining CUs stay free for
 *                           kernels of OTHER streams -- the data-parallel step runs the text tower's ~160 few-row launches
 *                           beside the head's convolutions; 0 (default): all CUs. */
/*   ACX_OPT_X6_TAIL_SPLIT   pairs = 6 problems with more 256 x 256 tiles than CUs whose last round of tiles fills only part of the chip
 *                           (and a caller-provided workspace): 1 = the whole tile rows of the full rounds as one launch, the
 *                           remaining rows as a K-split problem + reduce launch (ViT c_proj at 512 frames: +0.7 %; the four
 *                           products at 256 frames: +4-5 % frames/s).  The tail rows then sum K in another order than the
 *                           rows before them: identical frames are no longer bit-identical wherever they sit in a launch, which
 *                           is why the default is 0. */
/*   ACX_OPT_X6_STRIP_TAIL   pairs = 6 problems (identity rows, N % 256 == 0) whose last round of 256 x 256 tiles fills only part of the
 *                           chip: 1 (default) = the tiles of that round are cut into 128- or 64-column strips (one work item each)
 *                           when the cost model says the round gets shorter.  A strip runs the same K order and product order
 *                           per output element as a whole tile: results are BIT-IDENTICAL to 0 (= whole tiles only); 2 / 3 force
 *                           128- / 64-column strips (measurements). */
/*   ACX_OPT_X6_MIN_TILES    ACX_PREC_F32X6 drivers (acx_vit_encode, acx_transformer_forward): a product with at least this many
 *                           256 x 256 output tiles runs as a pairs = 6 product, smaller ones on the f32 MFMA kernels (default 18:
 *                           the ViT from 8 frames per launch; measured crossover, profiles/r06_x6_strip_tail.txt). */
enum { ACX_OPT_RING_MIN_TILES = 1, ACX_OPT_SK_MAX_M = 2, ACX_OPT_TN_P256_MIN_ROWS = 3, ACX_OPT_X6_CUS = 4, ACX_OPT_X6_TAIL_SPLIT = 5,
       ACX_OPT_X6_STRIP_TAIL = 6, ACX_OPT_X6_MIN_TILES = 7 };
int acx_set_option(acx_ctx* ctx, int32_t option, int64_t value);

/* ------------------------------------------------------------------------------------------
 * acx_gemm: C[M,N] = epilogue( amap(A)[M,K] . W[N,K]^T )            (MFMA-bound workhorse)
 * replaces every nn.Linear / F.linear / `@` / nn.Conv2d on the path:
 *   clip/model.py:192,197-199 (QKV, out-proj, MLP), :246-252 (patch conv as GEMM), :287-288
 *   (proj), text_encoder.py:23, temporal_model.py:31,43 (projection), axial_attention
 *   SelfAttention.to_q/to_kv/to_out and the two 3x3 convs of each feed-forward (implicit GEMM),
 *   selector_model.py:62 is a separate HBM-bound kernel (acx_selector_*).
 * epilogue order: +bias[n] -> activation -> +pos0[(m/gl)%gn][n] -> +pos1[m%gl][n] -> +residual[m][n]
 */
typedef struct acx_gemm_desc {
  const void* A;          /* [rows_of_A, lda]  f32 or bf16 */
  const void* W;          /* [N, ldw]          f32 (PREC_F32) or bf16 (PREC_BF16) */
  void* C;                /* [M, ldc]          f32 or bf16 */
  int32_t M, N, K;
  int32_t lda, ldw, ldc;
  int32_t a_dtype, c_dtype;
  int32_t prec;
  const float* bias;      /* [N] or NULL */
  int32_t act;
  const float* residual;  /* [M, ldr] f32 or NULL (may alias C) */
  int32_t ldr;
  const float* a_sub;     /* [K] subtracted from every A row before the product, or NULL
                             (selector_model.py:54 / anomaly_clip.py:143,201 re-centring) */
  int32_t amap;           /* ACX_AMAP_* */
  int32_t gn, gl;         /* grid (num_segments, seg_length) for CONV3X3 / TESTTILE / pos */
  int32_t cin;            /* CONV3X3: input channels; K == 9*cin, W laid out [N][tap=kh*3+kw][cin] */
  int32_t seg;            /* TESTTILE: segment_size S; row ((b s) n l) reads source row ((b n s l))
                             (temporal_model.py:46-53) */
  const float* pos0;      /* [gn, N] axial positional embedding param_0 (or NULL) */
  const float* pos1;      /* [gl, N] axial positional embedding param_1 (or NULL) */
  void* workspace;        /* optional: enables split-K for skinny problems (few output tiles, long K); */
  size_t workspace_bytes; /* needs up to 16*M*N*4 bytes, less is fine (fewer splits or none)           */
  /* Few-row fusions of the text tower (clip/model.py:183-185,188-230 and its dX chain).  Both need the few-row f32
   * kernel: f32 operands, identity row map, K % 256 == 0, N % 4 == 0, 16-byte aligned C / residual / gelu_grad_of;
   * ACX_E_UNSUPPORTED otherwise. */
  const void* zero_page;  /* optional: >= 256 bytes of zeros, 16-byte aligned, caller-owned.  With it, large CONV3X3 problems
                             (N >= 512, power-of-two grid, no residual) run on the LDS-DMA strip kernel: a tap outside the
                             token grid is a DMA from this page */
  int32_t a_act;          /* ACX_ACT_QUICKGELU: the activation is applied to A as it is read (x_next = gelu(pre) @ W^T) */
  const float* gelu_grad_of; /* [M, ldg] saved pre-activation p: C = (A W^T + bias) * d gelu(p)/dp  (no act / residual) */
  int32_t ldg;
  const float* a_norm_w;  /* LayerNorm over K applied to the A rows as they are read (y = LN(A) W^T + bias): weight / bias [K], or
                             NULL.  Few-row f32 kernel only, K == 512, no a_act / residual / gelu_grad_of / activation
                             (clip/model.py:214-216 ln_1 -> in_proj, ln_2 -> c_fc of the text tower); ACX_E_UNSUPPORTED otherwise */
  const float* a_norm_b;
  float a_norm_eps;
  uint32_t* counters;     /* optional: n_counters uint32, ZERO before the first call and left zero by every call (caller-owned,
                             one table per stream).  With it and `workspace`, few-row problems with K >= 1024 split K across
                             workgroups whose partial tiles meet through a last-arriver reduction (no second launch). */
  int32_t n_counters;
  const int32_t* tile_table; /* TILETABLE: [M / (gn gl)][2] device ints (base row, row stride between segments): output row
                                (tile, n, l) reads source row base + n * stride + l -- the test-mode tiling of
                                temporal_model.py:46-53 for a batch of videos with DIFFERENT segment sizes (video v, tile s:
                                base = row0_v + s gl, stride = S_v gl) */
  int32_t pairs;          /* 0 / 1: plain.  6 (with prec = ACX_PREC_BF16, a_dtype = ACX_BF16): f32-ACCURATE product on the bf16 matrix
                             cores.  A and W are each THREE bf16 planes (acx_split_bf16x3: x = hi + mid + lo to 24 bits; plane p of
                             A at A + p * a_plane_stride bytes, of W at W + p * w_plane_stride), and C accumulates, in f32, the six
                             cross products whose magnitude is >= 2^-16 of the leading one -- (hi,lo) (mid,mid) (lo,hi) (hi,mid)
                             (mid,hi) (hi,hi): per 32-wide k range, smallest first.  Every bf16 x bf16 product is exact in f32; the
                             three dropped terms and the split remainders are <= 2^-23 of the product: the error is that of an
                             f32 dot product, at 6/16 of the f32 MFMA's cost (2.5 PFLOP/s bf16 against 157 TFLOP/s f32 on gfx950).
                             Kernel: gemm_x6_p4_kernel (csrc/acx_gemm_x6.h).  K % 32 == 0, N % 4 == 0, 16-byte aligned planes; row
                             maps IDENTITY and CONV3X3 (power-of-two grid, cin % 32 == 0, M % 256 == 0, zero_page); epilogues bias,
                             QuickGELU / LeakyReLU, residual (f32 C), c_dtype ACX_F32 / ACX_BF16 / ACX_BF16X3[P] (plane output:
                             N % 8 == 0, no residual).  With `workspace`, launches of fewer output tiles than CUs split K across
                             workgroups (fixed split per shape; partial tiles + the generic reduce launch).
                             3: the same planes and kernel with the THREE leading products only -- (mid,hi) (hi,mid) (hi,hi); the lo
                             planes are never read.  The dropped terms are <= 2^-16 of the leading one: NOT f32-accurate (about 1e-5 of
                             sum |a||w|).  Identity rows, f32 or plane outputs, bias / QuickGELU / residual epilogues. */
  int32_t panels;         /* pairs = 6: bit 0 -- the A planes are in K-panel layout (ACX_BF16X3P, rows = a_plane_stride / (2 K));
                             bit 1 -- the W planes are (rows = N).  Identity row map only. */
  int64_t a_plane_stride, w_plane_stride;   /* bytes */
  int64_t c_plane_rows;   /* plane outputs (ACX_BF16X3 / ACX_BF16X3P): rows of the WHOLE plane image when C addresses a row sub-range of
                             it (plane p of the output at C + p * c_plane_rows * ldc elements; K-panel rows counted over c_plane_rows);
                             0: M.  acx_gemm uses it itself when it splits a launch into full rounds of tiles + a K-split tail. */
  float out_scale;        /* pairs = 3 with fp16 planes (a_dtype = ACX_F16): the accumulator is multiplied by this power of two before bias
                             / activation / residual (operand planes that were scaled by acx_split_f16x2); 0 = 1 */
} acx_gemm_desc;
int acx_gemm(acx_ctx* ctx, const acx_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * acx_layernorm: y[r,:] = norm(x[r,:]) * w + b     (HBM-bound; one wavefront per row)
 * replaces clip/model.py:174-180 (LayerNorm), classification_head.py:7,12, axial PreNorm and
 * ChanLayerNorm.  D must be a multiple of 64 and <= 1024... rows gathered with stride ldx. */
int acx_layernorm(acx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* b,
                  void* y, int64_t ldy, int32_t y_dtype, int64_t rows, int32_t D, float eps,
                  int32_t mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * acx_attention: multi-head softmax(q k^T / sqrt(64)) v for head dim 64, sequence <= 224.
 * replaces nn.MultiheadAttention's core as used by clip/model.py:206-212 (ViT: L=197, no mask;
 * text: L=77, causal mask clip/model.py:386-392).  qkv: [batch*L, 3*heads*64] packed as
 * in_proj output (q | k | v); out: [batch*L, heads*64]. */
int acx_attention(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                  int32_t batch, int32_t L, int32_t heads, int32_t causal, void* stream);
/* the same (non-causal ViT sequences, 128 < L <= 224) with the output as three bf16 planes hi | mid | lo (ACX_BF16X3:
 * plane p at (uint16_t*)out_planes + p * batch * L * ldo): the out-projection's A operand in ACX_PREC_F32X6 mode */
int acx_attention_x3(acx_ctx* ctx, const float* qkv, int64_t ldqkv, void* out_planes, int64_t ldo,
                     int32_t batch, int32_t L, int32_t heads, void* stream);
/* The ViT attention on the bf16 matrix cores at f32 accuracy (clip/model.py:206-212 in the ACX_PREC_F32X6 mode): q | k | v as
 * three bf16 planes in K-panel layout (ACX_BF16X3P of the in-projection output [batch * L, 3 heads * 64]: plane p at
 * (uint16_t*)qkv_planes + p * batch * L * 3 * heads * 64), S = Q K^T and O = P V as six-product bf16 x 6 products with f32
 * accumulation, softmax in f32; output = three bf16 planes of [batch * L, heads * 64] in K-panel layout (the out-projection's
 * A operand).  Non-causal, 192 < L <= 208 (seven 32-query tiles: the ViT-B/16 sequence of 197). */
int acx_attention_p3(acx_ctx* ctx, const void* qkv_planes, void* out_planes, int32_t batch, int32_t L, int32_t heads, void* stream);
/* ... with the number of cross products per contraction: 6 (= acx_attention_p3), 3 (the three leading ones: the operands' lo planes
 * are not read, the output's lo plane is not written -- the ACX_PREC_F32X3 mode, not f32-accurate), or 103 = three products on TWO
 * fp16 planes per operand (ACX_F16X2P images of q | k | v in, of the output out: the ACX_PREC_F16X3 mode) */
int acx_attention_p3n(acx_ctx* ctx, const void* qkv_planes, void* out_planes, int32_t batch, int32_t L, int32_t heads, int32_t products,
                      void* stream);
/* ... with the planes in K-panel layout (ACX_BF16X3P: [ldo / 32][batch * L][32] each; ldo == heads * 64) */
int acx_attention_x3_panel(acx_ctx* ctx, const float* qkv, int64_t ldqkv, void* out_planes, int64_t ldo,
                           int32_t batch, int32_t L, int32_t heads, void* stream);

/* bf16 variant of acx_attention for the bf16 mode of the ViT (NOT a parity path): qkv [batch*L, ldqkv] and out
 * [batch*L, ldo] are bf16, QK^T and PV run on the bf16 MFMA, softmax in f32.  Non-causal only. */
int acx_attention_bf16(acx_ctx* ctx, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int32_t batch,
                       int32_t L, int32_t heads, void* stream);
/* acx_attention_cls: same attention, but only for query row 0 of every sequence (out [batch, heads*64]).
 * Used for the LAST ViT layer, whose output is consumed at the CLS token only (clip/model.py:285). */
int acx_attention_cls(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                      int32_t batch, int32_t L, int32_t heads, void* stream);

/* ------------------------------------------------------------------------------------------
 * acx_vit_patches: im2col of 16x16/16 patches (clip/model.py:246-252,267-269).
 * frames [F,3,R,R] f32 -> patches [F*g*g, 3*P*P] (k = c*P*P + ky*P + kx, token = gy*g+gx); out_dtype ACX_F32, ACX_BF16, or
 * ACX_BF16X3P (three bf16 planes of the f32 pixels in K-panel layout: the patch embedding's A operand in ACX_PREC_F32X6). */
int acx_vit_patches(acx_ctx* ctx, const float* frames, void* patches, int32_t out_dtype,
                    int32_t F, int32_t R, int32_t P, void* stream);
/* acx_vit_embed: x[f,0,:] = cls + pos[0]; x[f,1+t,:] = patch_out[f,t,:] + pos[1+t]; then ln_pre
 * (clip/model.py:270-279).  patch_out [F*T, W] f32 -> x [F*(T+1), W] f32. */
int acx_vit_embed(acx_ctx* ctx, const float* patch_out, const float* cls, const float* pos,
                  const float* ln_w, const float* ln_b, float* x, int32_t F, int32_t T, int32_t W,
                  void* stream);

/* Transformer block weights (clip/model.py:188-217).  All f32 masters; the *_bf16 pointers are
 * the bf16 copies used when prec == ACX_PREC_BF16, or -- prec == ACX_PREC_F32X6 -- the THREE bf16 planes of the f32 weight in
 * K-panel layout (acx_split_bf16x3_panel: plane p at base + p * rows * cols * 2 bytes); may be NULL otherwise. */
typedef struct acx_block_weights {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  const float *in_proj_w, *in_proj_b;     /* [3W, W], [3W] */
  const float *out_proj_w, *out_proj_b;   /* [W, W], [W] */
  const float *fc_w, *fc_b;               /* [4W, W], [4W] */
  const float *proj_w, *proj_b;           /* [W, 4W], [W] */
  const void *in_proj_w_bf16, *out_proj_w_bf16, *fc_w_bf16, *proj_w_bf16;
} acx_block_weights;

typedef struct acx_vit_weights {
  const float* conv1_w;       /* [W, 3*P*P]  (conv1.weight viewed 2-D) */
  const void* conv1_w_bf16;
  const float* class_embedding;  /* [W] */
  const float* positional_embedding; /* [T+1, W] */
  const float *ln_pre_w, *ln_pre_b, *ln_post_w, *ln_post_b;
  const float* proj_t;        /* [E, W] = proj^T (proj is [W,E] in the reference, clip/model.py:264) */
  const void* proj_t_bf16;
  const acx_block_weights* blocks; /* [layers] (host array) */
} acx_vit_weights;

typedef struct acx_vit_desc {
  int32_t resolution, patch, width, layers, heads, embed_dim;
  int32_t prec;               /* ACX_PREC_* */
} acx_vit_desc;

/* acx_vit_encode: VisionTransformer.forward (clip/model.py:266-290): frames [F,3,R,R] f32 ->
 * features [F, embed_dim] f32.  Workspace from acx_vit_workspace_bytes(desc, F). */
size_t acx_vit_workspace_bytes(const acx_vit_desc* d, int32_t frames);
int acx_vit_encode(acx_ctx* ctx, const acx_vit_desc* d, const acx_vit_weights* w,
                   const float* frames, int32_t nframes, float* features, void* workspace,
                   size_t workspace_bytes, void* stream);

/* acx_transformer_forward: Transformer.forward (clip/model.py:220-230) in place on x
 * [batch*L, W] f32 -- used by the text encoder (text_encoder.py:16-18). */
size_t acx_transformer_workspace_bytes(int32_t width, int32_t rows);
size_t acx_transformer_workspace_bytes_prec(int32_t width, int32_t rows, int32_t prec);   /* ACX_PREC_F32X6 needs room for the planes */
int acx_transformer_forward(acx_ctx* ctx, float* x, int32_t batch, int32_t L, int32_t width,
                            int32_t heads, int32_t layers, int32_t causal, int32_t prec,
                            const acx_block_weights* blocks, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head, forward (HBM-bound rows of SURVEY.md 8a). */

/* acx_text_directions: selector_model.py:44-59: drop row normal_id, subtract ncentroid,
 * L2-normalise.  text [C, D] -> dirs [C-1, D]. */
int acx_text_directions(acx_ctx* ctx, const float* text, const float* ncentroid, float* dirs,
                        int32_t C, int32_t D, int32_t normal_id, void* stream);
/* acx_selector_project: raw[r, c] = (x[r,:] - ncentroid) . dirs[c,:]  (selector_model.py:54,62).
 * x [rows, D] (D in {64,128,256,512,768,1024}), raw [rows, C1], C1 <= 64. */
int acx_selector_project(acx_ctx* ctx, const float* x, const float* ncentroid, const float* dirs,
                         float* raw, int64_t rows, int32_t D, int32_t C1, void* stream);
/* acx_bn_stats: deterministic per-column batch statistics of raw[rows, C1] (training BatchNorm1d,
 * selector_model.py:30,65: biased variance normalises, unbiased variance feeds running_var).  Two
 * stages (row slabs -> f64 partials in `workspace` -> fixed-order sum); C1 <= 64;
 * workspace >= acx_bn_workspace_bytes(rows, C1), 8-byte aligned. */
size_t acx_bn_workspace_bytes(int64_t rows, int32_t C1);
/* SyncBN (configs/trainer/ddp.yaml sync_batchnorm): combine the ranks' statistics.  gathered[ranks][2 * C1 + 1] holds, per rank,
 * mean[C1], biased_var[C1] * rows, rows; outputs the statistics over all rows (Chan et al.) and the total row count (device
 * scalar), one launch; C1 <= 64. */
int acx_bn_combine(acx_ctx* ctx, const float* gathered, int32_t ranks, int32_t C1, float* mean, float* var_biased,
                   float* var_unbiased, float* total_rows, void* stream);
int acx_bn_stats(acx_ctx* ctx, const float* raw, int64_t rows, int32_t C1, float* mean,
                 float* var_biased, float* var_unbiased, void* workspace, size_t workspace_bytes,
                 void* stream);
/* acx_selector_project_stats: acx_selector_project with the batch statistics of its output
 * accumulated in the projection's epilogue (raw is not read again): the training-mode pair
 * selector_model.py:62 + :65's statistics in two launches.  Same workspace rule as acx_bn_stats. */
int acx_selector_project_stats(acx_ctx* ctx, const float* x, const float* ncentroid,
                               const float* dirs, float* raw, int64_t rows, int32_t D, int32_t C1,
                               float* mean, float* var_biased, float* var_unbiased,
                               void* workspace, size_t workspace_bytes, void* stream);
/* acx_selector_bn: logits = (raw - mean) / sqrt(var + eps)  (BatchNorm1d(C-1, affine=False),
 * selector_model.py:30,65).  mean/var [C1] (running stats in eval, batch stats in train). */
int acx_selector_bn(acx_ctx* ctx, const float* raw, const float* mean, const float* var,
                    float* logits, int64_t ldl, int64_t rows, int32_t C1, float eps, void* stream);

/* acx_axial_attention: the softmax(q k^T e^-1/2) v core of axial_attention.SelfAttention along one
 * axis of the (tiles, gn, gl) token grid.  qkv [rows, 3*heads*e] = (q | k | v) per token, out
 * [rows, heads*e].  axis 0: attend along gn (tokens with equal (tile, l)); axis 1: along gl. */
int acx_axial_attention(acx_ctx* ctx, const float* qkv, float* out, int32_t tiles, int32_t gn,
                        int32_t gl, int32_t heads, int32_t e, int32_t axis, void* stream);

/* acx_cls_head: score = sigmoid(LayerNorm((x1+x2)/2) . w + b) (ReversibleSequence mean +
 * classification_head.py:11-15); out_index maps row ((b s) n l) back to ((b n s l)) when seg > 0
 * (temporal_model.py:69-71). */
int acx_cls_head(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w,
                 const float* ln_b, const float* lin_w, const float* lin_b, float* scores,
                 int64_t rows, int32_t E, int32_t gn, int32_t gl, int32_t seg, void* stream);

/* acx_class_probs: softmax(similarity, dim=1) * score  (anomaly_clip_module.py:474-477). */
/* acx_cls_head with the scores scattered through a tile table (see acx_gemm_desc.tile_table): row (tile, n, l) is
 * written at base + n * stride + l -- the inverse tiling of temporal_model.py:67-73 for a batch of videos. */
int acx_cls_head_tiles(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w, const float* ln_b,
                       const float* lin_w, const float* lin_b, float* scores, int64_t rows, int32_t E, int32_t gn,
                       int32_t gl, const int32_t* tile_table, void* stream);
int acx_class_probs(acx_ctx* ctx, const float* sim, const float* scores, float* probs,
                    int64_t rows, int32_t C1, void* stream);

/* acx_prompt_embed: out[c,t,:] = cat(token_prefix, ctx, token_suffix)[c,t,:] + positional_embedding[t,:]
 * (coop.py:74-90 class_token_position "end" + text_encoder.py:15).  ctx is [C,n_ctx,W] or, when
 * shared_ctx != 0, [n_ctx,W] broadcast over classes (coop.py:76-77).  pos may be NULL (no add).  Lout (0 = Lc): only the
 * first Lout <= Lc positions of every class are written, out is [C, Lout, W] (the causal tower runs up to the last EOT). */
int acx_prompt_embed(acx_ctx* ctx, const float* prefix, const float* ctxv, const float* suffix,
                     const float* pos, float* out, int32_t C, int32_t n_ctx, int32_t Lc, int32_t W,
                     int32_t shared_ctx, int32_t Lout, void* stream);
/* acx_gather_rows: out[i,:] = x[idx[i],:]  (EOT-token gather, text_encoder.py:23). idx int64 on device. */
int acx_gather_rows(acx_ctx* ctx, const float* x, const int64_t* idx, float* out, int64_t n,
                    int32_t W, void* stream);

/* acx_add_bcast: out[r,:] = x[r,:] + p[:] over n rows of LW floats (text_encoder.py:15 when the
 * prompts were assembled without the positional embedding). */
int acx_add_bcast(acx_ctx* ctx, const float* x, const float* p, float* out, int64_t n, int64_t LW,
                  void* stream);
/* acx_concat_features: temporal-model input when concat_features is on: [logits | x - ncentroid | 0]
 * padded to Kp columns (anomaly_clip.py:143,201,223-233). */
int acx_concat_features(acx_ctx* ctx, const float* logits, const float* x, const float* ncentroid,
                        float* out, int64_t rows, int32_t C1, int32_t D, int32_t Kp, void* stream);

/* acx_preprocess_frames: decoded uint8 RGB frames [F,H,W,3] -> CLIP input [F,3,orows,ocols] f32:
 * Resize(shorter side, BICUBIC as PIL does it) + CenterCrop + /255 + Normalize (reference
 * src/utils/augmentations.py:21-34 and gtransforms.py GroupScale/GroupCenterCrop/GroupToTensor/GroupNormalize).
 * hbounds/hcoef: for each kept output column (xmin, n) and hksize 22-bit fixed-point coefficients (PIL's
 * precompute_coeffs + normalize_coeffs_8bpc, computed by anomalyclip_amd/preprocess.py); vbounds/vcoef likewise for
 * the kept output rows.  tmp: [F,H,ocols,3] uint8 scratch.  mean3/std3 are HOST pointers. */
int acx_preprocess_frames(acx_ctx* ctx, const unsigned char* frames, float* out, unsigned char* tmp,
                          const int32_t* hbounds, const int32_t* hcoef, int32_t hksize, const int32_t* vbounds,
                          const int32_t* vcoef, int32_t vksize, int32_t F, int32_t H, int32_t W, int32_t orows,
                          int32_t ocols, const float* mean3, const float* std3, void* stream);

/* utility: x[rows, cols] (f32, leading dimension ld) -> three dense bf16 planes dst + p * plane_stride_bytes, [rows, cols] each:
 * hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (each subtraction exact in f32): the operands of
 * acx_gemm_desc.pairs = 6.  cols % 4 == 0, 16-byte aligned src rows / 8-byte aligned planes. */
int acx_split_bf16x3(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows,
                     int64_t cols, void* stream);
/* the same into K-panel layout (ACX_BF16X3P: plane = [cols / 32][rows][32]; cols % 32 == 0) */
int acx_split_bf16x3_panel(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_bytes, int64_t rows, int64_t cols, void* stream);
/* TWO fp16 planes hi | lo of scale * src (hi = fp16(.), lo = fp16(. - hi); scale a power of two: exact), row-major or K-panel
 * layout: operands of a pairs = 3 product with a_dtype = ACX_F16 (the ACX_PREC_F16X3 arithmetic); the product's acx_gemm_desc.out_scale
 * takes the scales out again */
int acx_split_f16x2(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows, int64_t cols,
                    float scale, int32_t panel, void* stream);
/* n dense f32 tensors (numel[i] elements each, multiples of 8, 16-byte aligned) -> three row-major bf16 planes each
 * (dst[i]: 3 * numel[i] bf16, plane stride numel[i]) in ONE launch: a training step re-splits every convolution weight of the
 * temporal model after the optimizer has stepped. */
int acx_split_bf16x3_multi(acx_ctx* ctx, int32_t n, const float* const* src, void* const* dst, const int64_t* numel, void* stream);
/* utility: f32 -> bf16 (round-to-nearest-even) copy, used to prepare bf16 weight copies. */
int acx_cast_bf16(acx_ctx* ctx, const float* src, void* dst, int64_t n, void* stream);
/* utility: column sums of x[rows, D] accumulated into acc[D] (ncentroid, anomaly_clip_module.py:145-171). */
int acx_colsum(acx_ctx* ctx, const float* x, float* acc, int64_t rows, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training side (SURVEY.md 8a rows a4, a9, a12 and the backward of a2/a3/a6/a7/a8).  Row reductions
 * are two-stage and fixed-order: kernels emit per-block partials, acx_reduce_rows sums them. */

/* acx_gemm_tn: C[N1,N2] = sum_m A[m,n1] * bmap(B)[m,n2] (exact-f32 MFMA) -- weight gradients
 * dW = dY^T X of every nn.Linear / Conv2d of the temporal model and text_projection.  conv != 0: B column
 * k = tap*cin + ci reads token row shift_tap(m) on the (gn,gl) grid (zero outside) = dW of the 3x3 convs in
 * the [Cout][tap][Cin] layout.  b_sub [N2]: subtracted from every valid B row (selector direction gradient). */
size_t acx_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2);
int acx_gemm_tn(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                int32_t cin, void* workspace, size_t workspace_bytes, void* stream);
/* Several small weight gradients C_k[N1_k, N2_k] = A_k^T (B_k - b_sub_k) (identity row map, dense C_k) in ONE launch (+ one
 * reduce launch): each problem runs the blocks of its own acx_gemm_tn launch, so results are bit-identical to separate calls.
 * The temporal model's to_out / to_q|to_kv / projection gradients of a step (0.5-1.6 GFLOP each) are issued this way. */
typedef struct acx_tn_problem {
  const void* A; const void* B; void* C; const void* b_sub;
  int32_t M, N1, N2, lda, ldb, reserved;
} acx_tn_problem;
size_t acx_gemm_tn_group_workspace_bytes(int32_t nprob, const acx_tn_problem* probs);
int acx_gemm_tn_group(acx_ctx* ctx, int32_t nprob, const acx_tn_problem* probs, void* workspace, size_t workspace_bytes,
                      void* stream);
/* the same with a caller-owned zero page (>= 1024 bytes of zeros, 16-byte aligned, never written; may be NULL): the
 * 256 x 256 kernel then pads from it instead of clearing the tail of `workspace` with an extra launch per call */
int acx_gemm_tn_zp(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, void* stream);
/* acx_gemm_tn_zp WITHOUT its reduce launch: *splits_out (host) = number of K-split partial images left in `workspace`
 * ([splits][N1][N2] f32; their sum in image order is what acx_gemm_tn_zp writes to C); 1: C holds the result.  For consumers that
 * add the images themselves (acx_text_directions_bwd_parts). */
int acx_gemm_tn_parts(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                      int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                      int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, int32_t* splits_out, void* stream);
/* The same weight gradient as an f32-ACCURATE product on the bf16 matrix cores: A (dY, [M, lda]) and B (the layer input,
 * [M, ldb]) as three bf16 planes each (acx_split_bf16x3: plane p at base + p * plane_stride bytes), the six leading cross
 * products with f32 accumulation -- the TN instantiation of the plane-reuse kernel (acx_gemm_desc.pairs = 6).  N1, N2 multiples
 * of 256; conv: cin % 256 == 0, N2 == 9 cin, power-of-two grid, M % (gn gl) == 0.  zero_page: >= 256 bytes of zeros, 16-byte
 * aligned (rows behind M and taps outside the grid are DMA'd from it).  workspace: acx_gemm_tn_x6_workspace_bytes (the rows are
 * split across workgroups; fixed summation order for a given shape).  Replaces the weight-gradient products of
 * temporal_model.py:32-39's convolutions under autograd (loss.backward() in src/models/anomaly_clip_module.py:173-293). */
size_t acx_gemm_tn_x6_workspace_bytes(int32_t M, int32_t N1, int32_t N2);
int acx_gemm_tn_x6(acx_ctx* ctx, const void* A3, int64_t a_plane_stride, int32_t lda, const void* B3, int64_t b_plane_stride,
                   int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N1, int32_t N2, int32_t conv, int32_t gn, int32_t gl,
                   int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, void* stream);
int acx_reduce_rows(acx_ctx* ctx, const float* part, float* out, int32_t nparts, int32_t width, void* stream);
/* LayerNorm / ChanLayerNorm backward.  dx (may be NULL) = [add +] dx_scale * dL/dx (add [rows, D] may be NULL: the
 * residual branch of a pre-norm block, clip/model.py:214-216, folded into the same pass); part (may be NULL)
 * receives acx_row_parts(rows) rows of [dw(D) | db(D)] partials. */
int acx_layernorm_bwd(acx_ctx* ctx, const float* x, const float* w, const float* dy, float* dx, float* part,
                      int64_t rows, int32_t D, float eps, int32_t mode, float dx_scale, const float* add, void* stream);
/* backward of acx_cls_head (seg == 0): dx = dL/dx1 = dL/dx2; part: acx_row_parts(rows) rows of
 * [d ln_w (E) | d ln_b (E) | d lin_w (E) | d lin_b (1) | pad 3]. */
int acx_cls_head_bwd(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w, const float* ln_b,
                     const float* lin_w, const float* scores, const float* dscores, float* dx, float* part,
                     int64_t rows, int32_t E, void* stream);
/* elementwise: mode 0 LeakyReLU' from saved output, 1 QuickGELU' from saved pre-activation (out = d * f'),
 * 2 QuickGELU forward (d ignored). */
int acx_act(acx_ctx* ctx, const float* saved, const float* d, float* out, int64_t n, int32_t mode, void* stream);
/* LeakyReLU(0.01) backward for the bf16 x 6 path of the temporal model's feed-forward (temporal_model.py:32-39 under autograd):
 * u_hi = the hi bf16 plane of the saved activation (its sign is the activation's), d [n] the upstream gradient; out [n] f32 and
 * planes (three bf16 planes, plane p at (uint16_t*)planes + p * plane_elems) both receive d * (u > 0 ? 1 : 0.01). */
int acx_leaky_grad_planes(acx_ctx* ctx, const void* u_hi, const float* d, float* out, void* planes, int64_t plane_elems, int64_t n,
                          void* stream);
int acx_add(acx_ctx* ctx, const float* a, const float* b, float* out, int64_t n, void* stream);
int acx_transpose(acx_ctx* ctx, const float* in, float* out, int32_t R, int32_t Cn, void* stream);
/* conv weight [Cout,Cin,3,3] -> [Cin][tap'][Cout] with flipped taps: the W operand of the dX implicit GEMM */
int acx_conv_weight_dx(acx_ctx* ctx, const float* w, float* out, int32_t Cout, int32_t Cin, void* stream);
/* backward of acx_axial_attention (axis 0/1 on the (tiles,gn,gl) grid) and of acx_attention (tiles=batch,
 * gn=1, gl=L, axis=1, e=64, causal as in the forward): dqkv [rows, 3*heads*e]. */
int acx_seq_attention_bwd(acx_ctx* ctx, const float* qkv, const float* dout, float* dqkv, int32_t tiles,
                          int32_t gn, int32_t gl, int32_t heads, int32_t e, int32_t axis, int32_t causal,
                          float* stats_ws /* [rows*heads*3] floats or NULL */, void* stream);
/* gradients of the axial positional embeddings (pos0 [gn,E], pos1 [gl,E]) from dx [(tile,n,l), E]; `part` is a
 * scratch buffer of tiles*gn*E + tiles*ceil(gn/4)*gl*E floats (two-stage fixed-order reduction). */
int acx_pos_grad(acx_ctx* ctx, const float* dx, float* d0, float* d1, float* part, int32_t tiles, int32_t gn,
                 int32_t gl, int32_t E, void* stream);
/* BatchNorm1d(affine=False) training backward in two steps (so data-parallel SyncBN can all-reduce the
 * sums in between): stats -> sums[2*C1] = (sum dl, sum dl*xhat) per column over this rank's rows; apply ->
 * draw[r*ldo + c] with total_rows = rows of ALL ranks. */
int acx_bn_bwd_stats(acx_ctx* ctx, const float* logits, const float* dlogits, float* sums, int64_t rows,
                     int32_t C1, void* workspace /* acx_bn_workspace_bytes */, size_t workspace_bytes, void* stream);
int acx_bn_bwd_apply(acx_ctx* ctx, const float* logits, const float* dlogits, const float* var_biased,
                     const float* sums, float* draw, int32_t ldo, int64_t rows, int64_t total_rows, int32_t C1,
                     float eps, const float* total_rows_dev /* device scalar overriding total_rows, or NULL: SyncBN keeps
                     the all-gathered row count on the device instead of synchronising the host for it */, void* stream);
/* (draw is [rows, ldo], ldo >= C1: the kernel writes the pad columns C1 .. ldo - 1 as zeros itself) */
/* y = a*x + b*y (BatchNorm running-statistics update, selector_model.py:30 momentum 0.1) */
int acx_axpby(acx_ctx* ctx, const float* x, float* y, int32_t n, float a, float b, void* stream);
/* deterministic column sums: part[blk][D] partials over rows_per_block rows each (then acx_reduce_rows) */
int acx_colsum_partials(acx_ctx* ctx, const float* x, int32_t ld, float* part, int64_t rows, int32_t D,
                        int32_t rows_per_block, void* stream);
int acx_text_directions_bwd(acx_ctx* ctx, const float* text, const float* ncentroid, const float* ddirs,
                            float* dtext, int32_t C, int32_t D, int32_t normal_id, void* stream);
/* acx_select_idx: selector_model.py:119-158 / 227-266.  logits [B, N*Lg, C1]; masks [B,N] (1 keep, 0 drop);
 * idx_top [B,ktop] / idx_bot [B,kbot] int64 (first B/2 rows abnormal videos, rest normal).  Ties are broken
 * towards the LOWER segment index (torch.topk leaves tie order unspecified). */
int acx_select_idx(acx_ctx* ctx, const float* logits, const int64_t* labels, const float* mask_top,
                   const float* mask_bot, int64_t* idx_top, int64_t* idx_bot, int32_t B, int32_t N, int32_t Lg,
                   int32_t C1, int32_t normal_id, int32_t ktop, int32_t kbot, void* stream);
int acx_gather_segments(acx_ctx* ctx, const float* logits, const int64_t* idx, float* out, int32_t B, int32_t N,
                        int32_t Lg, int32_t C1, int32_t K, void* stream);
int acx_scatter_segments(acx_ctx* ctx, const float* dout, const int64_t* idx, float* dlogits, int32_t B, int32_t N,
                         int32_t Lg, int32_t C1, int32_t K, void* stream);
/* acx_mil_loss: ComputeLoss.__call__ (loss.py:51-195) forward AND backward in one pass.  losses[8] = (cost,
 * ldir_abn, ldir_nor, ltopk_abn, lbottomk_abn, ltopk_nor, lsmooth, lsparse); dsim / dsim_topk / dscores are the
 * gradients of gout*cost.  lambdas[7] = (dir_abn, dir_nor, topk_abn, bottomk_abn, topk_nor, smooth, sparse).
 * workspace: ceil(B*N*Lg/256)*8 + ceil(B*K*Lg/256) floats. */
int acx_mil_loss(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                 const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn,
                 float* dsim, float* dsim_topk, float* dscores, float* losses, float* workspace,
                 size_t workspace_floats, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t K, int32_t normal_id,
                 const float* lambdas, const float* gout /* device scalar or NULL (=1) */, void* stream);
/* acx_mil_loss as ONE launch: `counter` = one uint32, zero before the first call and left zero by every call (caller-owned;
 * NULL = the three-launch form above).  The block partials are added in a different (fixed) order than the three-launch
 * form's serial sum, in f64 either way. */
int acx_mil_loss_one(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                 const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn,
                 float* dsim, float* dsim_topk, float* dscores, float* losses, float* workspace,
                 size_t workspace_floats, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t K, int32_t normal_id,
                 const float* lambdas, const float* gout, uint32_t* counter, void* stream);
/* The loss, its gradients AND the selector BatchNorm's backward statistics as ONE launch (the whole-step graph's middle):
 * acx_mil_loss_one + acx_axpby(meter += losses) + acx_scatter_segments(dsim += dsim_topk at the top-k segments) + acx_bn_bwd_stats in
 * one kernel with the SAME arithmetic in the same order (bit-identical outputs): dlogits [B N Lg, C1] = the gradient wrt the logits
 * incl. the gathered top-k rows' share, bn_sums [2 C1] = (sum dl, sum dl * xhat), meter [8] (may be NULL) += losses.  Needs
 * B N Lg % 256 == 0 and <= 131072 rows (a block of 256 frame rows is a slab of acx_bn_bwd_stats); ACX_E_UNSUPPORTED otherwise.
 * workspace: ceil(B N Lg / 256) * 8 + ceil(B K Lg / 256) floats; bn_workspace: acx_bn_workspace_bytes; counter as acx_mil_loss_one. */
int acx_mil_loss_bn(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                    const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn, float* dlogits,
                    float* dscores, float* losses, float* meter, float* bn_sums, float* workspace, size_t workspace_floats,
                    void* bn_workspace, size_t bn_workspace_bytes, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t K,
                    int32_t normal_id, const float* lambdas, const float* gout, uint32_t* counter, void* stream);
/* The selector's forward tail as ONE launch, one workgroup per video (selector_model.py:60-99,119-225): [SyncBN combine of
 * `gathered` [ranks, 2 C1 + 1] -> stat_out [3 C1 + 1] = mean | biased var | unbiased var | total rows; gathered == NULL: the local
 * statistics mean / var_biased / var_unbiased] -> logits = BatchNorm(raw) [rows, ldl] -> [running statistics, NULL: none] ->
 * top-k / bottom-k segment picks (acx_select_idx) -> logits_topk [B ktop Lg, C1] (acx_gather_segments of idx_top).  The same
 * expressions in the same order as acx_bn_combine / acx_selector_bn / acx_bn_running_update / acx_select_idx /
 * acx_gather_segments: bit-identical results, four or five launches less. */
int acx_selector_tail(acx_ctx* ctx, const float* raw, const float* gathered, int32_t ranks, const float* mean,
                      const float* var_biased, const float* var_unbiased, float* stat_out, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float one_minus, float* logits, int64_t ldl, const int64_t* labels,
                      const float* mask_top, const float* mask_bot, int64_t* idx_top, int64_t* idx_bot, float* logits_topk, int32_t B,
                      int32_t N, int32_t Lg, int32_t C1, int32_t normal_id, int32_t ktop, int32_t kbot, float eps, void* stream);
/* acx_text_directions_bwd with d_dirs given as the K-split partial images of the TN product that forms it (acx_gemm_tn_parts:
 * image s at ddirs_parts + s * part_stride floats, rows = direction index, row stride D), summed here in image order. */
int acx_text_directions_bwd_parts(acx_ctx* ctx, const float* text, const float* ncentroid, const float* ddirs_parts, int32_t nparts,
                                  int64_t part_stride, float* dtext, int32_t C, int32_t D, int32_t normal_id, void* stream);
/* acx_adamw: one torch.optim.AdamW step (decoupled weight decay, bias correction; step counts from 1). */
int acx_adamw(acx_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
              float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* acx_multi_axpy: y_i += a * x_i for nseg tensors in ONE launch (HOST arrays of device pointers / sizes): the
 * gradient accumulation autograd performs with one elementwise launch per parameter (AccumulateGrad). */
int acx_multi_axpy(acx_ctx* ctx, int32_t nseg, void* const* y, const void* const* x, const int64_t* n, float a,
                   void* stream);
/* acx_adamw_multi: the same update for nseg parameter tensors in ONE launch (per-tensor lr / weight decay: the
 * reference's four param groups, anomaly_clip_module.py:693-746; common betas / eps / step).  The pointer and size
 * arrays are HOST arrays of length nseg; entries with n[i] <= 0 are skipped.  Hyper-parameters are doubles: the scalar
 * terms (1 - beta, bias corrections, lr / bc1, 1 - lr * wd) are formed in f64 and rounded once, like torch does. */
int acx_adamw_multi(acx_ctx* ctx, int32_t nseg, void* const* p, const void* const* g, void* const* m, void* const* v,
                    const int64_t* n, const double* lr, const double* weight_decay, double beta1, double beta2, double eps,
                    int32_t step, void* stream);
/* ---- whole-step training graph support (anomalyclip_amd/components/step_graph.py; reference: what Lightning's automatic
 * optimisation + DDP + torch.optim.AdamW do around training_step, anomaly_clip_module.py:203-293,693-746,
 * configs/trainer/ddp.yaml:1-9).  Nothing here allocates; every call is capturable in a HIP graph. */
/* number of per-block partial rows acx_layernorm_bwd (part != NULL) and acx_cls_head_bwd write for `rows` rows */
int64_t acx_row_parts(int64_t rows);
/* strided 2-D copies / transposes of f32 blocks, ALL in one launch: dst[r * dst_ld + c] = src[r * src_ld + c], or with
 * `transpose` dst[c * dst_ld + r] = src[r * src_ld + c]; rows / cols are the SOURCE block's extent, leading dimensions
 * in elements.  Builds every derived weight layout of the temporal model (temporal_model.py:18-40 parameters -> kernel
 * operands: q|kv concatenation, W^T for the dX GEMMs, flipped-tap conv weights, padded projection, positional tables). */
typedef struct acx_prep_seg {
  const void* src;
  void* dst;
  int32_t rows, cols, src_ld, dst_ld;
  int32_t transpose, reserved;
} acx_prep_seg;
int acx_prep_multi(acx_ctx* ctx, int32_t nseg, const acx_prep_seg* segs, void* stream);
/* y_i = x_i for nseg contiguous f32 tensors, one launch */
int acx_multi_copy(acx_ctx* ctx, int32_t nseg, void* const* y, const void* const* x, const int64_t* n, void* stream);
/* HOST helper: the f32 scalars of one AdamW step, out[0] = sqrt(1 - beta2^step), out[1 + 2 i] = 1 - lr_i wd_i,
 * out[2 + 2 i] = lr_i / (1 - beta1^step) -- formed in f64 and rounded once, exactly as acx_adamw_multi forms them. */
int acx_adamw_hyper(int32_t nseg, const double* lr, const double* weight_decay, double beta1, double beta2, int32_t step,
                    float* out);
/* acx_adamw_multi with those scalars read from DEVICE memory (`hyper_dev`: 1 + 2 nseg floats, refreshed by the caller
 * before each step), so a captured launch follows the LR schedule and the step count; every gradient is multiplied by
 * `grad_scale` as it is read (1 / world size after a summing all-reduce, 1.0f otherwise).  torch.optim.AdamW semantics
 * (anomaly_clip_module.py:693-746; wd 0.2 in the reference model configs). */
int acx_adamw_multi_dev(acx_ctx* ctx, int32_t nseg, void* const* p, const void* const* g, void* const* m, void* const* v,
                        const int64_t* n, const float* hyper_dev, float grad_scale, double beta1, double beta2, double eps,
                        void* stream);
/* SyncBatchNorm payload of a rank (configs/trainer/ddp.yaml:9): out[2 C1 + 1] = [mean | biased var * rows | rows] */
int acx_bn_pack(acx_ctx* ctx, const float* mean, const float* var_biased, int64_t rows, int32_t C1, float* out,
                void* stream);
/* nn.BatchNorm1d running statistics (selector_model.py:30,65): r = one_minus * r + momentum * batch; counter += 1 */
int acx_bn_running_update(acx_ctx* ctx, const float* mean, const float* var_unbiased, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, int32_t C1, float momentum, float one_minus,
                          void* stream);
int acx_fill_f32(acx_ctx* ctx, float* p, int64_t n, float value, void* stream);
/* out[D] = column sums of x[rows, ld] in ONE launch, fixed summation order (bias gradients: the reference's autograd sums
 * dY over rows for every nn.Linear / nn.Conv2d bias).  part: scratch of acx_colsum_fused_part_bytes(rows, D) bytes;
 * counters: >= ceil(D / 64) uint32, ZERO before the first call and left zero by every call (caller-owned, reusable).
 * out = sums + beta * out (beta = 0: overwrite; 1: the ncentroid accumulation of anomaly_clip_module.py:145-171). */
size_t acx_colsum_fused_part_bytes(int64_t rows, int32_t D);
int acx_colsum_fused(acx_ctx* ctx, const float* x, int32_t ld, int64_t rows, int32_t D, float* out, float* part,
                     size_t part_bytes, uint32_t* counters, float beta, void* stream);
/* nprob column sums in ONE launch (part[i]: acx_colsum_fused_part_bytes(rows[i], D[i]) bytes each; `counters`: ncounters
 * zero-at-rest uint32, at least sum_i ceil(D[i] / 64)); and nprob row-partial tables reduced in one launch
 * (out_i[width_i] = sum_p part_i[p][:], the summation tree of acx_reduce_rows). */
int acx_colsum_fused_group(acx_ctx* ctx, int32_t nprob, const void* const* x, const int32_t* ld, const int64_t* rows,
                           const int32_t* D, void* const* out, void* const* part, uint32_t* counters, int32_t ncounters,
                           void* stream);
int acx_reduce_rows_group(acx_ctx* ctx, int32_t nprob, const void* const* part, void* const* out, const int32_t* nparts,
                          const int32_t* width, void* stream);

int acx_ctx_grad(acx_ctx* ctx, const float* dx, float* dctx, int32_t C, int32_t n_ctx, int32_t Lc, int32_t W,
                 int32_t shared_ctx, void* stream);
int acx_scatter_rows(acx_ctx* ctx, const float* src, const int64_t* idx, float* out, int64_t n, int32_t W,
                     void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Metrics epilogue: anomaly_clip_module.py:501-626 and the torchmetrics==0.11.0
 * curve functions it calls (requirements.txt:6; functional/classification/{precision_recall_curve,roc,auroc,
 * average_precision}.py).  All results are exact-integer or fixed-order f64 -- run-to-run deterministic.
 * --------------------------------------------------------------------------------------------------------- */
/* Stable LSD radix sort (4 passes of 8 bits) of (f32 key, u32 payload) pairs; replaces
 * `torch.argsort(preds, descending=True)` + gathers of _binary_clf_curve.  Not in place.  -0.0 < +0.0. */
int64_t acx_sort_workspace_bytes(int64_t n);
int acx_sort_pairs(acx_ctx* ctx, const float* keys, const uint32_t* vals, float* keys_out, uint32_t* vals_out,
                   int64_t n, int32_t descending, void* workspace, int64_t workspace_bytes, void* stream);
/* `batch` independent sorts of n pairs in one launch sequence (12 launches for any batch).  Problem b reads keys + b * key_stride and
 * vals + b * val_stride (val_stride = 0: all problems carry the SAME payload array); outputs are dense [batch][n].
 * workspace_bytes >= batch * acx_sort_workspace_bytes(n).  Replaces the per-class loop of anomaly_clip_module.py:507-518. */
int acx_sort_pairs_batched(acx_ctx* ctx, const float* keys, int64_t key_stride, const uint32_t* vals, int64_t val_stride,
                           float* keys_out, uint32_t* vals_out, int64_t n, int32_t batch, int32_t descending, void* workspace,
                           int64_t workspace_bytes, void* stream);

typedef struct acx_curve_result {   /* written to DEVICE memory */
  double auroc;          /* trapz(tpr, fpr) over the distinct thresholds; 0 when a class is absent (torchmetrics) */
  double ap;             /* sum (R_k - R_{k-1}) P_k; NaN without positives */
  int64_t n_pos, n_neg, n_distinct;
  int64_t opt_index;     /* sorted index of argmax(tpr - fpr) (first maximum, exact integer compare), -1 = the
                            prepended (0,0) point */
  float opt_threshold;   /* thresholds[argmax(tpr - fpr)] (anomaly_clip_module.py:526-527); 1.0 for the (0,0) point */
  float pad;
} acx_curve_result;
/* From pairs sorted by score DESCENDING: target_i = (label_i == cls) (negate=0) or (label_i != cls) (negate=1,
 * the reference's labels_binary, :520).  Optional curve_* arrays (capacity n) receive the n_distinct points
 * (tps, fps, threshold) of _binary_clf_curve; pass NULL to skip. */
int64_t acx_clf_curve_workspace_bytes(int64_t n);
int acx_clf_curve(acx_ctx* ctx, const float* sorted_scores, const uint32_t* sorted_labels, int64_t n, int32_t cls,
                  int32_t negate, acx_curve_result* result, int32_t* curve_tps, int32_t* curve_fps,
                  float* curve_thresholds, void* workspace, int64_t workspace_bytes, void* stream);
/* `batch` (1..64) curves in one launch sequence (4 launches): problem b reads sorted_scores + b * score_stride and
 * sorted_labels + b * label_stride, target = (label == cls[b]), or (label != cls[b]) where negate[b] != 0; results[b] is its record;
 * cls / negate are HOST arrays.  The optional curve arrays (all three or none) receive the points of problem 0 only.
 * workspace_bytes >= batch * acx_clf_curve_workspace_bytes(n). */
int acx_clf_curve_batched(acx_ctx* ctx, const float* sorted_scores, int64_t score_stride, const uint32_t* sorted_labels,
                          int64_t label_stride, int64_t n, int32_t batch, const int32_t* cls, const int32_t* negate,
                          acx_curve_result* results, int32_t* curve_tps, int32_t* curve_fps, float* curve_thresholds, void* workspace,
                          int64_t workspace_bytes, void* stream);
/* anomaly_clip_module.py:538-581, 621-626, 673: y_pred and the integer counters behind top-1 / top-5 accuracy,
 * the confusion matrix and F1@{0.1..1.0}.  probs [n, C-1] (class_probs without the normal column), labels int64,
 * threshold = DEVICE pointer to the optimal threshold (e.g. &result->opt_threshold).  counts (int64, device,
 * 3C + C*C + 30 entries): top1_hit[C] | top5_hit[C] | class_n[C] | confusion[C][C] (row = true) | f1_tp[10] |
 * f1_fp[10] | f1_fn[10].  Probability ties rank the lower class index first (torch.topk leaves it unspecified). */
int acx_test_counts(acx_ctx* ctx, const float* scores, const float* probs, const int64_t* labels, int64_t n,
                    int32_t C, int32_t normal_idx, const float* threshold, int32_t* y_pred, int64_t* counts,
                    void* stream);

/* In-library launch timer used by bench.py's roofline leg: when enabled, every kernel launch is
 * bracketed by HIP events on the caller's stream.  kinds: 0 GEMM, 1 attention, 2 norm rows, 3 other.
 * acx_prof_collect synchronises the recorded events and returns per-kind launch counts and summed
 * durations (ms), then resets the recording. */
#define ACX_PROF_KINDS 4
/* on = 1: every launch kind; on = 2: the GEMM launches only (an event pair costs ~7 us of queue time: 260 launches of a headline
 * step with all kinds armed add 1.9 ms to its 131.5 ms, the ~110 GEMM launches alone 0.8 ms); on = 0: off */
int acx_prof_enable(acx_ctx* ctx, int on);
int acx_prof_collect(acx_ctx* ctx, int32_t* counts, double* total_ms);
/* executed GEMM flops (2*M*N*K of every acx_gemm / acx_gemm_tn launch) since acx_prof_enable(ctx, 1) */
int acx_prof_gemm_flops(acx_ctx* ctx, double* flops);
/* the acx_gemm_tn share of the above: flops since acx_prof_enable(ctx, 1); summed launch time and launch count of the events the
 * LAST acx_prof_collect consumed (call it after acx_prof_collect) */
int acx_prof_gemm_tn(acx_ctx* ctx, double* flops, double* total_ms, int32_t* launches);

/* ------------------------------------------------------------------------------------------
 * Measured roofline denominators (bench.py `peaks_measured`).  Not on the product path.
 * acx_probe_mfma: register-only MFMA loop on every SIMD (bf16 = 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16),
 *   `iters` x 4 MFMAs per wave, waves_per_simd waves per SIMD; *flops_out (host) = flops the launch executes.
 * acx_probe_copy: dst = src, 16 bytes per lane, grid-stride: the HBM stream (read + write) ceiling. */
int acx_probe_mfma(acx_ctx* ctx, int32_t bf16, int32_t iters, int32_t waves_per_simd, float* sink, double* flops_out,
                   void* stream);
int acx_probe_copy(acx_ctx* ctx, const void* src, void* dst, int64_t bytes, void* stream);
/* acx_probe_read: reads `bytes` once (16 bytes per lane, grid-stride), writes nothing: the floor of a one-shot launch over an
 * input of that size -- the denominator for the head's skinny reductions (selector projection, column sums). */
int acx_probe_read(acx_ctx* ctx, const void* src, int64_t bytes, float* sink, void* stream);

/* ------------------------------------------------------------------------------------------
 * Collectives (acx_comm_init / acx_allreduce).  The reference exchanges gradients and SyncBatchNorm statistics
 * through Lightning DDP over NCCL (configs/trainer/ddp.yaml:1-9).  The Python host of this repository issues the same exchange through
 * torch.distributed -- backend "nccl" is RCCL on ROCm -- and that remains its default ("collectives"); these entry
 * points are the torch-free form of it for a C / C++ host: thin RCCL calls on the CALLER's stream (so they order with the library's
 * kernels like any launch, and a capturing stream records them into its HIP graph).  RCCL is opened with dlopen on first use
 * (libacx.so has no link-time dependency on it; a process that already loaded an RCCL -- PyTorch-ROCm -- shares that copy);
 * ACX_E_RCCL when it cannot be opened or a call fails.  One communicator per context, bound to the context's device.
 *   acx_comm_unique_id  rank 0: fills ACX_COMM_ID_BYTES opaque bytes (ncclGetUniqueId) that the host transports to every rank
 *   acx_comm_init       every rank, collectively: ncclCommInitRank(world, id, rank)
 *   acx_allreduce       in place over `count` elements of `dtype` (ACX_F32 / ACX_BF16 / ACX_F64 / ACX_I64), op ACX_COMM_SUM / MAX / MIN
 *   acx_allgather       `count` elements from every rank, rank-major, into recv[world * count]
 *   acx_comm_destroy    releases the communicator (acx_destroy does NOT: collective teardown is the caller's to order) */
#define ACX_COMM_ID_BYTES 128
enum { ACX_F64 = 16, ACX_I64 = 17 };                  /* element types of the collectives only */
enum { ACX_COMM_SUM = 0, ACX_COMM_MAX = 1, ACX_COMM_MIN = 2 };
int acx_comm_unique_id(void* id_out, size_t id_bytes);
int acx_comm_init(acx_ctx* ctx, int32_t rank, int32_t world, const void* unique_id);
int acx_comm_destroy(acx_ctx* ctx);
int acx_comm_info(acx_ctx* ctx, int32_t* rank, int32_t* world);   /* world = 0: no communicator */
int acx_allreduce(acx_ctx* ctx, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream);
int acx_allgather(acx_ctx* ctx, const void* send, void* recv, int64_t count, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H_ */
