// libacx context management and the whole-encoder drivers (host code only: they sequence the
// kernels of acx_gemm / acx_norm / acx_attn / acx_head on the caller's stream; no allocation,
// no synchronisation, safe to capture in a hipGraph).
#include "acx_internal.h"

#include <new>
#include <stdlib.h>

thread_local char acx_tls_err[512] = {0};

extern "C" int acx_attention_cls(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo, int32_t batch,
                                 int32_t L, int32_t heads, void* stream);

extern "C" int acx_version(void) { return ACX_VERSION; }

extern "C" int acx_create(acx_ctx** out, int device) {
  if (!out) return acx_fail(nullptr, ACX_E_BADARG, "acx_create: null out%s");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || device < 0 || device >= n) {
    snprintf(acx_tls_err, 512, "acx_create: device %d not available (%s, %d devices)", device,
             e == hipSuccess ? "ok" : hipGetErrorString(e), n);
    return ACX_E_HIP;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    snprintf(acx_tls_err, 512, "acx_create: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return ACX_E_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(acx_tls_err, 512, "acx_create: libacx is built for gfx950 only, device %d is %s", device, prop.gcnArchName);
    return ACX_E_UNSUPPORTED;
  }
  acx_ctx* c = new (std::nothrow) acx_ctx;
  if (!c) return acx_fail(nullptr, ACX_E_HIP, "acx_create: out of host memory%s");
  c->device = device;
  c->multiprocessors = prop.multiProcessorCount;
  c->opt_ring_min_tiles = 512;
  c->opt_sk_max_m = 320;
  c->opt_tn_p256_min_rows = 4096;
  c->opt_x6_cus = 0;
  c->opt_x6_tail = 0;
  c->opt_x6_strip = 1;
  c->comm = nullptr; c->comm_rank = 0; c->comm_world = 0;
  c->opt_x6_min_tiles = 18;
  c->err[0] = 0;
  c->prof_on = false;
  c->prof_gemm_only = false;
  c->prof_n = c->prof_created = 0;
  c->prof_gemm_flops = 0.0;
  c->prof_tn_flops = c->prof_tn_ms = 0.0; c->prof_tn_count = 0;
  c->prof_ev = new (std::nothrow) hipEvent_t[2 * ACX_PROF_MAX];
  c->prof_kind = new (std::nothrow) unsigned char[ACX_PROF_MAX];
  *out = c;
  return ACX_OK;
}

extern "C" void acx_destroy(acx_ctx* ctx) {
  if (!ctx) return;
  for (int i = 0; i < 2 * ctx->prof_created; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
  delete[] ctx->prof_ev;
  delete[] ctx->prof_kind;
  delete ctx;
}

extern "C" int acx_prof_enable(acx_ctx* ctx, int on) {
  if (!ctx || !ctx->prof_ev || !ctx->prof_kind) return acx_fail(ctx, ACX_E_BADARG, "acx_prof_enable: no context%s");
  ctx->prof_on = on != 0;
  ctx->prof_gemm_only = on == 2;
  if (on) { ctx->prof_n = 0; ctx->prof_gemm_flops = 0.0; ctx->prof_tn_flops = 0.0; }
  return ACX_OK;
}

extern "C" int acx_prof_gemm_flops(acx_ctx* ctx, double* flops) {
  if (!ctx || !flops) return acx_fail(ctx, ACX_E_BADARG, "acx_prof_gemm_flops: null pointer%s");
  *flops = ctx->prof_gemm_flops;
  return ACX_OK;
}

extern "C" int acx_prof_gemm_tn(acx_ctx* ctx, double* flops, double* total_ms, int32_t* launches) {
  if (!ctx || !flops || !total_ms || !launches) return acx_fail(ctx, ACX_E_BADARG, "acx_prof_gemm_tn: null pointer%s");
  *flops = ctx->prof_tn_flops; *total_ms = ctx->prof_tn_ms; *launches = ctx->prof_tn_count;
  return ACX_OK;
}

extern "C" int acx_prof_collect(acx_ctx* ctx, int32_t* counts, double* total_ms) {
  if (!ctx || !counts || !total_ms) return acx_fail(ctx, ACX_E_BADARG, "acx_prof_collect: null pointer%s");
  for (int k = 0; k < ACX_K_COUNT; ++k) { counts[k] = 0; total_ms[k] = 0.0; }
  ctx->prof_tn_ms = 0.0; ctx->prof_tn_count = 0;
  for (int i = 0; i < ctx->prof_n; ++i) {
    hipError_t e = hipEventSynchronize(ctx->prof_ev[2 * i + 1]);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]);
    if (e != hipSuccess) {
      snprintf(ctx->err, 512, "acx_prof_collect: %s", hipGetErrorString(e));
      return ACX_E_HIP;
    }
    int kind = ctx->prof_kind[i];
    if (kind == ACX_K_GEMM_TN) { ctx->prof_tn_ms += ms; ctx->prof_tn_count++; kind = ACX_K_GEMM; }
    counts[kind]++;
    total_ms[kind] += ms;
  }
  ctx->prof_n = 0;
  return ACX_OK;
}

extern "C" int acx_set_option(acx_ctx* ctx, int32_t option, int64_t value) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: no context%s");
  switch (option) {
    case ACX_OPT_RING_MIN_TILES:
      if (value < 1) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: ring_min_tiles must be >= 1%s");
      ctx->opt_ring_min_tiles = (int)value;
      return ACX_OK;
    case ACX_OPT_TN_P256_MIN_ROWS:
      if (value < 1) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: tn_p256_min_rows must be >= 1%s");
      ctx->opt_tn_p256_min_rows = (int)(value > 0x7fffffff ? 0x7fffffff : value);
      return ACX_OK;
    case ACX_OPT_X6_TAIL_SPLIT:
      ctx->opt_x6_tail = value != 0;
      return ACX_OK;
    case ACX_OPT_X6_STRIP_TAIL:
      if (value < 0 || value > 3) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: x6_strip_tail is 0 (off), 1 (cost model), 2 or 3 (forced strip width)%s");
      ctx->opt_x6_strip = (int)value;
      return ACX_OK;
    case ACX_OPT_X6_MIN_TILES:
      if (value < 1) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: x6_min_tiles must be >= 1%s");
      ctx->opt_x6_min_tiles = (int)(value > 0x7fffffff ? 0x7fffffff : value);
      return ACX_OK;
    case ACX_OPT_X6_CUS:
      if (value < 0) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: x6_cus must be >= 0%s");
      ctx->opt_x6_cus = (int)(value > 4096 ? 4096 : value);
      return ACX_OK;
    case ACX_OPT_SK_MAX_M:
      if (value < 0) return acx_fail(ctx, ACX_E_BADARG, "acx_set_option: sk_max_m must be >= 0%s");
      ctx->opt_sk_max_m = (int)value;
      return ACX_OK;
    default:
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_set_option: unknown option %s%ld", "", (long)option);
  }
}

extern "C" const char* acx_last_error(acx_ctx* ctx) { return ctx ? ctx->err : acx_tls_err; }

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct TfWs {   // transformer scratch carved from the caller's workspace
  char *h, *qkv, *att, *mlp, *splitk;
  char *hp, *mp;  // ACX_PREC_F32X6: three bf16 planes of a [rows, W] / [rows, 4W] GEMM input (nullptr otherwise)
  char* qkv3;     // ACX_PREC_F32X6: three bf16 planes of q | k | v [rows, 3W] (the plane attention's operands)
  size_t splitk_bytes, total;
};

// the plane modes of the drivers: ACX_PREC_F32X6 (six products: f32-accurate) and ACX_PREC_F32X3 (the three leading products)
static inline bool is_xmode(int prec) { return prec == ACX_PREC_F32X6 || prec == ACX_PREC_F32X3 || prec == ACX_PREC_F16X3; }
static thread_local int tl_x_pairs = 6;          // acx_gemm_desc.pairs of the driver's plane products (set at the driver's entry)
static thread_local bool tl_x_f16 = false;       // ACX_PREC_F16X3: two fp16 planes per operand (weights scaled by ACX_F16X3_WSCALE)
static inline void set_xmode(int prec) { tl_x_pairs = (prec == ACX_PREC_F32X3 || prec == ACX_PREC_F16X3) ? 3 : 6; tl_x_f16 = prec == ACX_PREC_F16X3; }

TfWs carve_tf(char* base, int64_t rows, int W, int prec = ACX_PREC_F32) {
  TfWs w;
  size_t off = 0;
  w.h = base + off;   off += al((size_t)rows * W * 4);
  w.qkv = base + off; off += al((size_t)rows * 3 * W * 4);
  w.att = base + off; off += al((size_t)rows * W * 4);
  w.mlp = base + off; off += al((size_t)rows * 4 * W * 4);
  // split-K scratch for skinny problems (text tower: 14 x 77 rows): 8 partial copies of the widest output
  // ... and for the partial last wave of big problems: <= 256 tiles of 128x128 outputs, 4 splits
  w.splitk_bytes = rows <= 4096 ? (size_t)8 * rows * 4 * W * 4 : (size_t)4 * 256 * 128 * 128 * 4;
  w.splitk = base + off; off += al(w.splitk_bytes);
  w.hp = w.mp = w.qkv3 = nullptr;
  if (is_xmode(prec)) {
    w.hp = base + off; off += al((size_t)3 * rows * W * 2);
    w.mp = base + off; off += al((size_t)3 * rows * 4 * W * 2);
    w.qkv3 = base + off; off += al((size_t)3 * rows * 3 * W * 2);
  }
  w.total = off;
  return w;
}

// ACX_PREC_F32X6: the large GEMMs of the ViT as f32-accurate products on the bf16 matrix cores (acx_gemm_desc.pairs = 6): the
// f32 input is split into three bf16 planes (acx_split_bf16x3), the weights come as three planes from the caller.  A problem
// the persistent 256x256 kernel does not take (few rows) stays on the f32 MFMA kernels.
bool x6_takes(acx_ctx* ctx, const TfWs& ws, int64_t M, int N, int K, int lda) {
  if (!ws.hp || !ACX_DBG_SWITCH("X6", true)) return false;
  const int64_t rtiles = ((M + 255) / 256) * ((N + 255) / 256);
  const int x6_min = ctx ? ctx->opt_x6_min_tiles : 512;
  return rtiles >= x6_min && K % 256 == 0 && N % 8 == 0 && (size_t)M * lda * 2 < ((size_t)1 << 32) &&
         (size_t)N * K * 2 < ((size_t)1 << 32);
}

// Both operands' planes are in K-PANEL layout (ACX_BF16X3P: the driver's producers -- LayerNorm, the attention, c_fc's epilogue,
// acx_split_bf16x3_panel -- write it, the caller's weight planes come in it): a 256-row x 32-column unit of the plane-reuse
// kernel is then ONE contiguous 16 KB block (+5-6 % on the ViT products against row-major planes, profiles/r05_gemm_x6_notes.txt).
int linear_x6(acx_ctx* ctx, const void* A3, int lda, int64_t a_rows, const void* W3, int64_t w_plane_bytes, int ldw, void* C,
              int ldc, int M, int N, int K, const float* bias, int act, const float* residual, hipStream_t s, int ldr = 0,
              int c_dtype = ACX_F32, void* tail_ws = nullptr, size_t tail_bytes = 0) {
  if (!W3) return acx_fail(ctx, ACX_E_BADARG, "driver: missing bf16 x 3 weight planes for ACX_PREC_F32X6%s");
  acx_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A3; d.W = W3; d.C = C;
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldw = ldw; d.ldc = ldc;
  d.a_dtype = tl_x_f16 ? ACX_F16 : ACX_BF16; d.c_dtype = (tl_x_f16 && c_dtype == ACX_BF16X3P) ? ACX_F16X2P : c_dtype; d.prec = ACX_PREC_BF16;
  d.out_scale = tl_x_f16 ? 1.f / ACX_F16X3_WSCALE : 0.f;
  d.bias = bias; d.act = act; d.residual = residual; d.ldr = ldr ? ldr : ldc;
  d.pairs = tl_x_pairs; d.a_plane_stride = a_rows * (int64_t)lda * 2; d.w_plane_stride = w_plane_bytes;
  d.panels = 3;
  // scratch for the K split of a partly filled last round of tiles (acx_gemm: row-major outputs only)
  d.workspace = tail_ws; d.workspace_bytes = tail_ws ? tail_bytes : 0;
  return acx_gemm(ctx, &d, s);
}

int linear1(acx_ctx* ctx, int prec, const void* A, int a_dtype, int lda, const float* Wf, const void* Wb, int ldw,
            void* C, int c_dtype, int ldc, int M, int N, int K, const float* bias, int act, const float* residual,
            hipStream_t s, int ldr, void* splitk_ws, size_t splitk_bytes) {
  acx_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.C = C;
  d.W = prec == ACX_PREC_BF16 ? Wb : (const void*)Wf;
  if (!d.W) return acx_fail(ctx, ACX_E_BADARG, "driver: missing %s weight copy for the requested precision",
                            prec == ACX_PREC_BF16 ? "bf16" : "f32");
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldw = ldw; d.ldc = ldc;
  d.a_dtype = a_dtype; d.c_dtype = c_dtype; d.prec = prec;
  d.bias = bias; d.act = act; d.residual = residual; d.ldr = ldr ? ldr : ldc;
  d.workspace = splitk_bytes ? splitk_ws : nullptr; d.workspace_bytes = splitk_bytes;
  return acx_gemm(ctx, &d, s);
}

// Linear layer with tail handling: a grid of T tiles runs in ceil(T/512) waves of 128x128 tiles (2 per CU); when
// the last wave would be mostly empty (e.g. N=768: 9.23 waves -> 10), the rows of that partial wave are issued as a
// second launch that fills the chip (9 waves + ~1/3 wave): acx_gemm runs it on 64x64 tiles (f32, four blocks per CU)
// or splits K over the workspace handed in here (other precisions).
int linear(acx_ctx* ctx, int prec, const void* A, int a_dtype, int lda, const float* Wf, const void* Wb, int ldw,
           void* C, int c_dtype, int ldc, int M, int N, int K, const float* bias, int act, const float* residual,
           hipStream_t s, int ldr = 0, void* splitk_ws = nullptr, size_t splitk_bytes = 0) {
  bool tail_split = ACX_DBG_SWITCH("TAIL_SPLIT", true);
  if (tail_split && !ldr) {   // the persistent strip-stream kernel balances its own tail
    acx_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldw = ldw; d.ldc = ldc;
    d.a_dtype = a_dtype; d.c_dtype = c_dtype; d.prec = prec; d.act = act; d.residual = residual;
    if (acx_gemm_takes_strip_stream(&d)) tail_split = false;
  }
  const int tiles_n = (N + 127) / 128, tiles_m = (M + 127) / 128;
  const long tiles = (long)tiles_m * tiles_n;
  const long full = tiles / 512;
  const long rem = tiles - full * 512;
  if (tail_split && splitk_ws && full >= 2 && rem > 0 && rem <= 256 && !ldr) {
    const int head_mt = (int)((full * 512) / tiles_n);
    const int head_rows = head_mt * 128;
    const int tail_rows = M - head_rows;
    const size_t need = (size_t)4 * tail_rows * N * sizeof(float);
    if (head_rows > 0 && tail_rows > 0 && need <= splitk_bytes) {
      const size_t asz = a_dtype == ACX_BF16 ? 2 : 4, csz = c_dtype == ACX_BF16 ? 2 : 4;
      int rc = linear1(ctx, prec, A, a_dtype, lda, Wf, Wb, ldw, C, c_dtype, ldc, head_rows, N, K, bias, act, residual, s, 0,
                       nullptr, 0);
      if (rc) return rc;
      return linear1(ctx, prec, (const char*)A + (size_t)head_rows * lda * asz, a_dtype, lda, Wf, Wb, ldw,
                     (char*)C + (size_t)head_rows * ldc * csz, c_dtype, ldc, tail_rows, N, K, bias, act,
                     residual ? residual + (size_t)head_rows * ldc : nullptr, s, 0, splitk_ws, splitk_bytes);
    }
  }
  return linear1(ctx, prec, A, a_dtype, lda, Wf, Wb, ldw, C, c_dtype, ldc, M, N, K, bias, act, residual, s, ldr, splitk_ws,
                 splitk_bytes);
}

// cls_ws != nullptr: the LAST layer is evaluated only where its output is consumed (token 0 of every sequence,
// clip/model.py:285): K and V for all tokens, everything else for `batch` rows.  The final residual stream of
// the CLS tokens is left COMPACT in cls_ws[0 .. batch*W) instead of x.
int transformer_layers(acx_ctx* ctx, float* x, int batch, int L, int W, int heads, int layers, int causal, int prec,
                       const acx_block_weights* blk, const TfWs& ws, hipStream_t s, float* cls_ws = nullptr) {
  const int64_t rows = (int64_t)batch * L;
  const bool x6mode = is_xmode(prec);
  const int pdt3 = prec == ACX_PREC_F16X3 ? ACX_F16X2P : prec == ACX_PREC_F32X3 ? ACX_BF16X2P : ACX_BF16X3P;   // LayerNorm's plane output
  const bool f16mode = prec == ACX_PREC_F16X3;
  if (x6mode) prec = ACX_PREC_F32;               // everything that is not one of the four large GEMMs runs as in f32 mode
  const int hdt = prec == ACX_PREC_BF16 ? ACX_BF16 : ACX_F32;
  const size_t esz = prec == ACX_PREC_BF16 ? 2 : 4;
  int rc;
  for (int l = 0; l < layers; ++l) {
    const acx_block_weights& b = blk[l];
    if (cls_ws && l == layers - 1) {
      float* xc = cls_ws;                        // [batch, W]  final CLS residual stream
      float* attc = cls_ws + (size_t)batch * W;  // [batch, W]
      float* x1 = attc + (size_t)batch * W;      // [batch, W]
      char* hc = (char*)(x1 + (size_t)batch * W);          // [batch, W]   (f32 or bf16)
      char* mc = hc + (size_t)batch * W * 4;               // [batch, 4W]  (f32 or bf16)
      const bool kv_x6 = x6mode && x6_takes(ctx, ws, rows, 2 * W, W, W);
      // K | V for every token: rows [W, 3W) of in_proj
      if (kv_x6) {
        // LayerNorm writes the product's three planes itself (all rows) and the f32 rows of the CLS tokens (Q below reads only those)
        if (!b.in_proj_w_bf16) return acx_fail(ctx, ACX_E_BADARG, "driver: missing bf16 x 3 weight planes for ACX_PREC_F32X6%s");
        if ((rc = acx_layernorm(ctx, x, W, b.ln1_w, b.ln1_b, ws.hp, W, pdt3, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
        if ((rc = acx_layernorm(ctx, x, (int64_t)L * W, b.ln1_w, b.ln1_b, ws.h, (int64_t)L * W, hdt, batch, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
        if ((rc = linear_x6(ctx, ws.hp, W, rows, (const char*)b.in_proj_w_bf16 + (size_t)W * 64 /* row W of every K-panel */, (int64_t)3 * W * W * 2, W,
                            (float*)ws.qkv + W, 3 * W, (int)rows, 2 * W, W, b.in_proj_b + W, ACX_ACT_NONE, nullptr, s))) return rc;
      } else {
      if ((rc = acx_layernorm(ctx, x, W, b.ln1_w, b.ln1_b, ws.h, W, hdt, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
      if ((rc = linear(ctx, prec, ws.h, hdt, W, b.in_proj_w + (size_t)W * W,
                       b.in_proj_w_bf16 ? (const char*)b.in_proj_w_bf16 + (size_t)W * W * 2 : nullptr, W,
                       (float*)ws.qkv + W, ACX_F32, 3 * W, (int)rows, 2 * W, W, b.in_proj_b + W, ACX_ACT_NONE, nullptr, s))) return rc;
      }
      // Q for the CLS rows only (A rows strided by one sequence)
      if ((rc = linear(ctx, prec, ws.h, hdt, L * W, b.in_proj_w, b.in_proj_w_bf16, W, ws.qkv, ACX_F32, L * 3 * W, batch, W, W,
                       b.in_proj_b, ACX_ACT_NONE, nullptr, s))) return rc;
      if ((rc = acx_attention_cls(ctx, (const float*)ws.qkv, 3 * W, attc, W, batch, L, heads, s))) return rc;
      if ((rc = linear(ctx, prec, attc, ACX_F32, W, b.out_proj_w, b.out_proj_w_bf16, W, x1, ACX_F32, W, batch, W, W,
                       b.out_proj_b, ACX_ACT_NONE, x, s, L * W))) return rc;
      if ((rc = acx_layernorm(ctx, x1, W, b.ln2_w, b.ln2_b, hc, W, hdt, batch, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
      if ((rc = linear(ctx, prec, hc, hdt, W, b.fc_w, b.fc_w_bf16, W, mc, hdt, 4 * W, batch, 4 * W, W, b.fc_b,
                       ACX_ACT_QUICKGELU, nullptr, s))) return rc;
      if ((rc = linear(ctx, prec, mc, hdt, 4 * W, b.proj_w, b.proj_w_bf16, 4 * W, xc, ACX_F32, W, batch, W, 4 * W,
                       b.proj_b, ACX_ACT_NONE, x1, s))) return rc;
      (void)esz;
      continue;
    }
    // x = x + attn(ln_1(x))                                          clip/model.py:215
    // (ACX_PREC_F32X6: the producers of the four large GEMMs' inputs write the three bf16 planes themselves -- LayerNorm and
    // the QuickGELU epilogue of c_fc -- only the attention output goes through acx_split_bf16x3)
    const bool x6_qkv = x6mode && x6_takes(ctx, ws, rows, 3 * W, W, W), x6_out = x6mode && x6_takes(ctx, ws, rows, W, W, W);
    const bool x6_fc = x6mode && x6_takes(ctx, ws, rows, 4 * W, W, W), x6_proj = x6mode && x6_takes(ctx, ws, rows, W, 4 * W, 4 * W);
    if (x6_qkv) {
      if ((rc = acx_layernorm(ctx, x, W, b.ln1_w, b.ln1_b, ws.hp, W, pdt3, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
    } else
    if ((rc = acx_layernorm(ctx, x, W, b.ln1_w, b.ln1_b, ws.h, W, hdt, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
    // bf16 mode, non-causal (the ViT): q/k/v, the attention and its output stay bf16 end to end -- half the
    // QKV store traffic, bf16-MFMA attention, and a bf16 A operand (LDS-DMA kernels) for the out-projection
    const bool ab = prec == ACX_PREC_BF16 && !causal && L <= 224 && ACX_DBG_SWITCH("ATTN_BF16", true);
    const int qdt = ab ? ACX_BF16 : ACX_F32;
    // the attention on the bf16 matrix cores (acx_attention_p3): q | k | v leave the in-projection as three bf16 planes
    const bool att_p3 = x6_qkv && x6_out && ws.qkv3 && !causal && L > 192 && L <= 208 && ACX_DBG_SWITCH("ATTN_P3", true);
    if (x6_qkv) {
      if ((rc = linear_x6(ctx, ws.hp, W, rows, b.in_proj_w_bf16, (int64_t)3 * W * W * 2, W, att_p3 ? (void*)ws.qkv3 : (void*)ws.qkv, 3 * W,
                          (int)rows, 3 * W, W, b.in_proj_b, ACX_ACT_NONE, nullptr, s, 0, att_p3 ? ACX_BF16X3P : ACX_F32,
                          att_p3 ? ws.qkv : ws.h, att_p3 ? (size_t)rows * 3 * W * 4 : (size_t)rows * W * 4))) return rc;   // (free f32 buffers: tail scratch)
    } else
    if ((rc = linear(ctx, prec, ws.h, hdt, W, b.in_proj_w, b.in_proj_w_bf16, W, ws.qkv, qdt, 3 * W, (int)rows, 3 * W, W,
                     b.in_proj_b, ACX_ACT_NONE, nullptr, s, 0, ws.splitk, ws.splitk_bytes))) return rc;
    const bool att_x3 = x6_out && !causal && L > 128 && L <= 224 && ACX_DBG_SWITCH("ATTN16", true);
    if (f16mode && (x6_qkv || x6_out || x6_fc || x6_proj) && !(att_p3 && x6_fc && x6_proj))
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "driver: ACX_PREC_F16X3 needs the planes attention (192 < L <= 208) and all four products on the plane kernel%s");
    if (ab) {
      if ((rc = acx_attention_bf16(ctx, ws.qkv, 3 * W, ws.att, W, batch, L, heads, s))) return rc;
    } else if (att_p3) {   // planes in, planes out
      if ((rc = acx_attention_p3n(ctx, ws.qkv3, ws.hp, batch, L, heads, f16mode ? 103 : tl_x_pairs, s))) return rc;
    } else if (att_x3) {   // the attention writes the out-projection's three planes itself
      if ((rc = acx_attention_x3_panel(ctx, (const float*)ws.qkv, 3 * W, ws.hp, W, batch, L, heads, s))) return rc;
    } else {
      if ((rc = acx_attention(ctx, (const float*)ws.qkv, 3 * W, (float*)ws.att, W, batch, L, heads, causal, s))) return rc;
    }
    if (x6_out) {
      if (!att_x3 && !att_p3 && (rc = acx_split_bf16x3_panel(ctx, (const float*)ws.att, W, ws.hp, (int64_t)rows * W * 2, rows, W, s))) return rc;
      // (ws.h is free here and at c_proj: LayerNorm's output has been consumed, or went to the planes)
      if ((rc = linear_x6(ctx, ws.hp, W, rows, b.out_proj_w_bf16, (int64_t)W * W * 2, W, x, W, (int)rows, W, W, b.out_proj_b,
                          ACX_ACT_NONE, x, s, 0, ACX_F32, ws.h, (size_t)rows * W * 4))) return rc;
    } else
    if ((rc = linear(ctx, prec, ws.att, qdt, W, b.out_proj_w, b.out_proj_w_bf16, W, x, ACX_F32, W, (int)rows, W, W,
                     b.out_proj_b, ACX_ACT_NONE, x, s, 0, ws.splitk, ws.splitk_bytes))) return rc;
    // x = x + mlp(ln_2(x))                                           clip/model.py:216
    if (x6_fc) {
      if ((rc = acx_layernorm(ctx, x, W, b.ln2_w, b.ln2_b, ws.hp, W, pdt3, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
    } else
    if ((rc = acx_layernorm(ctx, x, W, b.ln2_w, b.ln2_b, ws.h, W, hdt, rows, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
    if (x6_fc) {
      // QuickGELU(c_fc) straight into the planes of c_proj's input when c_proj takes the x6 path too
      if ((rc = linear_x6(ctx, ws.hp, W, rows, b.fc_w_bf16, (int64_t)4 * W * W * 2, W, x6_proj ? (void*)ws.mp : (void*)ws.mlp, 4 * W,
                          (int)rows, 4 * W, W, b.fc_b, ACX_ACT_QUICKGELU, nullptr, s, 0, x6_proj ? ACX_BF16X3P : ACX_F32,
                          x6_proj ? ws.mlp : ws.h, x6_proj ? (size_t)rows * 4 * W * 4 : (size_t)rows * W * 4))) return rc;
    } else
    if ((rc = linear(ctx, prec, ws.h, hdt, W, b.fc_w, b.fc_w_bf16, W, ws.mlp, hdt, 4 * W, (int)rows, 4 * W, W, b.fc_b,
                     ACX_ACT_QUICKGELU, nullptr, s, 0, ws.splitk, ws.splitk_bytes))) return rc;
    if (x6_proj) {
      if (!x6_fc && (rc = acx_split_bf16x3_panel(ctx, (const float*)ws.mlp, 4 * W, ws.mp, (int64_t)rows * 4 * W * 2, rows, 4 * W, s))) return rc;
      if ((rc = linear_x6(ctx, ws.mp, 4 * W, rows, b.proj_w_bf16, (int64_t)W * 4 * W * 2, 4 * W, x, W, (int)rows, W, 4 * W, b.proj_b,
                          ACX_ACT_NONE, x, s, 0, ACX_F32, ws.h, (size_t)rows * W * 4))) return rc;
    } else
    if ((rc = linear(ctx, prec, ws.mlp, hdt, 4 * W, b.proj_w, b.proj_w_bf16, 4 * W, x, ACX_F32, W, (int)rows, W, 4 * W,
                     b.proj_b, ACX_ACT_NONE, x, s, 0, ws.splitk, ws.splitk_bytes))) return rc;
  }
  return ACX_OK;
}

}  // namespace

extern "C" size_t acx_transformer_workspace_bytes(int32_t width, int32_t rows) {
  return carve_tf(nullptr, rows, width).total;
}
extern "C" size_t acx_transformer_workspace_bytes_prec(int32_t width, int32_t rows, int32_t prec) {
  return carve_tf(nullptr, rows, width, prec).total;
}

extern "C" int acx_transformer_forward(acx_ctx* ctx, float* x, int32_t batch, int32_t L, int32_t width, int32_t heads,
                                       int32_t layers, int32_t causal, int32_t prec, const acx_block_weights* blocks,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !blocks || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_transformer_forward: null pointer%s");
  if (width != heads * 64) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_transformer_forward: head dim must be 64%s");
  // ACX_PREC_F32X6 with a workspace sized for it (acx_transformer_workspace_bytes_prec); a smaller (f32-sized) workspace runs
  // the f32 kernels
  TfWs ws = carve_tf((char*)workspace, (int64_t)batch * L, width, prec);
  if (ws.total > workspace_bytes && is_xmode(prec)) ws = carve_tf((char*)workspace, (int64_t)batch * L, width);
  if (ws.total > workspace_bytes) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_transformer_forward: workspace too small%s");
  set_xmode(prec);
  return transformer_layers(ctx, x, batch, L, width, heads, layers, causal, prec, blocks, ws, (hipStream_t)stream);
}

namespace {
struct VitWs {
  char *patches, *patch_out, *x, *cls, *cls_ws;
  size_t tf_off, total;
};
VitWs carve_vit(char* base, const acx_vit_desc* d, int F) {
  const int g = d->resolution / d->patch, T = g * g, W = d->width;
  const int K = 3 * d->patch * d->patch;
  VitWs w;
  size_t off = 0;
  w.patches = base + off;   off += al((size_t)F * T * K * (is_xmode(d->prec) ? 6 : 4));   // (plane modes: three bf16 planes)
  w.patch_out = base + off; off += al((size_t)F * T * W * 4);
  w.x = base + off;         off += al((size_t)F * (T + 1) * W * 4);
  w.cls = base + off;       off += al((size_t)F * W * 4);
  w.cls_ws = base + off;    off += al((size_t)F * W * 4 * 8);      // xc, attc, x1, hc, mc(4W)
  w.tf_off = off;
  off += carve_tf(nullptr, (int64_t)F * (T + 1), W, d->prec).total;
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t acx_vit_workspace_bytes(const acx_vit_desc* d, int32_t frames) {
  if (!d || frames <= 0) return 0;
  return carve_vit(nullptr, d, frames).total;
}

extern "C" int acx_vit_encode(acx_ctx* ctx, const acx_vit_desc* d, const acx_vit_weights* w, const float* frames,
                              int32_t nframes, float* features, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !w || !frames || !features || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_encode: null pointer%s");
  if (nframes <= 0) return ACX_OK;
  if (d->width != d->heads * 64) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_vit_encode: head dim must be 64%s");
  if (d->resolution % d->patch || d->patch % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_encode: bad patch geometry%s");
  const int g = d->resolution / d->patch, T = g * g, W = d->width, F = nframes;
  const int K = 3 * d->patch * d->patch;
  if (T + 1 > 224) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_vit_encode: more than 224 tokens%s");
  const VitWs ws = carve_vit((char*)workspace, d, F);
  if (ws.total > workspace_bytes) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_vit_encode: workspace too small%s");
  hipStream_t s = (hipStream_t)stream;
  const int prec = is_xmode(d->prec) ? ACX_PREC_F32 : d->prec;   // final projection (and small launches): f32 kernels
  set_xmode(d->prec);
  const int pdt = prec == ACX_PREC_BF16 ? ACX_BF16 : ACX_F32;
  int rc;
  // conv1 as GEMM over im2col'ed patches                               clip/model.py:267-269
  const TfWs tf = carve_tf((char*)workspace + ws.tf_off, (int64_t)F * (T + 1), W, d->prec);
  if (is_xmode(d->prec) && w->conv1_w_bf16 && K % 32 == 0 && x6_takes(ctx, tf, (int64_t)F * T, W, K, K)) {
    // ACX_PREC_F32X6: im2col writes the three bf16 planes of the pixels (K-panel layout), the embedding is a pairs = 6 product
    // like the layers' GEMMs (conv1_w_bf16: the weight's three K-panel planes); scratch for a K-split tail: the f32 q | k | v buffer
    if ((rc = acx_vit_patches(ctx, frames, ws.patches, d->prec == ACX_PREC_F16X3 ? ACX_F16X2P : d->prec == ACX_PREC_F32X3 ? ACX_BF16X2P : ACX_BF16X3P,
                              F, d->resolution, d->patch, s))) return rc;
    if ((rc = linear_x6(ctx, ws.patches, K, (int64_t)F * T, w->conv1_w_bf16, (int64_t)W * K * 2, K, ws.patch_out, W, F * T, W, K, nullptr,
                        ACX_ACT_NONE, nullptr, s, 0, ACX_F32, tf.qkv, (size_t)F * (T + 1) * 3 * W * 4))) return rc;
  } else {
  if ((rc = acx_vit_patches(ctx, frames, ws.patches, pdt, F, d->resolution, d->patch, s))) return rc;
  if ((rc = linear(ctx, prec, ws.patches, pdt, K, w->conv1_w, prec == ACX_PREC_BF16 ? w->conv1_w_bf16 : nullptr, K, ws.patch_out, ACX_F32, W, F * T, W, K,
                   nullptr, ACX_ACT_NONE, nullptr, s))) return rc;
  }
  // CLS + positional embedding + ln_pre                                :270-279
  if ((rc = acx_vit_embed(ctx, (const float*)ws.patch_out, w->class_embedding, w->positional_embedding, w->ln_pre_w,
                          w->ln_pre_b, (float*)ws.x, F, T, W, s))) return rc;
  const bool prune = ACX_DBG_SWITCH("VIT_PRUNE", true);
  if ((rc = transformer_layers(ctx, (float*)ws.x, F, T + 1, W, d->heads, d->layers, 0, d->prec, w->blocks, tf, s,
                               prune ? (float*)ws.cls_ws : nullptr))) return rc;
  // ln_post on the CLS rows, then @ proj                                :285-288
  if ((rc = acx_layernorm(ctx, prune ? (const float*)ws.cls_ws : (const float*)ws.x, prune ? (int64_t)W : (int64_t)(T + 1) * W,
                          w->ln_post_w, w->ln_post_b, ws.cls, W, ACX_F32, F, W, 1e-5f, ACX_NORM_LAYER, s))) return rc;
  return linear(ctx, prec, ws.cls, ACX_F32, W, w->proj_t, w->proj_t_bf16, W, features, ACX_F32, d->embed_dim, F,
                d->embed_dim, W, nullptr, ACX_ACT_NONE, nullptr, s);
}
