// Row-wise normalisation kernels (HBM-bound): one 64-lane wavefront per row, values kept in
// registers between the two reduction passes, 16-byte loads/stores when the row allows.
//   acx_layernorm  : nn.LayerNorm (clip/model.py:174-180, classification_head.py:7) and the
//                    axial_attention ChanLayerNorm variant (eps added to the std)
//   acx_vit_embed  : CLS concat + positional embedding + ln_pre (clip/model.py:270-279)
//   acx_cls_head   : mean of the two reversible streams + LayerNorm + Linear(E,1) + sigmoid
//                    (classification_head.py:11-15), with the inverse test-mode tiling on store
#include "acx_internal.h"

namespace {

template <int VPL>
__device__ __forceinline__ void normalize(float (&v)[VPL], float eps, int mode) {
  constexpr float invD = 1.f / (64 * VPL);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += v[i];
  const float mean = wave_sum(s) * invD;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] -= mean;
    q += v[i] * v[i];
  }
  const float var = wave_sum(q) * invD;
  const float scale = mode == ACX_NORM_LAYER ? 1.f / sqrtf(var + eps) : 1.f / (sqrtf(var) + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] *= scale;
}

// OUT_BF16 == 2: the row as THREE bf16 planes hi | mid | lo (x = hi + mid + lo to 24 bits: the A operand of acx_gemm_desc.pairs =
// 6), plane p at y + p * plane elements
// OUT_BF16 == 3: the same planes in K-panel layout (ACX_BF16X3P): y = plane base, element e of row `prow` of `prows` rows at
// ((e / 32) * prows + prow) * 32 + e % 32
template <int VPL, int OUT_BF16>
__device__ __forceinline__ void store_row_affine(void* y, int lane, const float (&v)[VPL],
                                                 const float* __restrict__ w, const float* __restrict__ b, int64_t plane = 0,
                                                 int64_t prow = 0, int64_t prows = 0) {
  if constexpr (VPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VPL / 4; ++i) {
      const int e = 4 * lane + 256 * i;
      const float4 ww = *reinterpret_cast<const float4*>(w + e);
      const float4 bb = *reinterpret_cast<const float4*>(b + e);
      float4 o;
      o.x = v[4 * i] * ww.x + bb.x; o.y = v[4 * i + 1] * ww.y + bb.y;
      o.z = v[4 * i + 2] * ww.z + bb.z; o.w = v[4 * i + 3] * ww.w + bb.w;
      if constexpr (OUT_BF16 == 2 || OUT_BF16 == 3) {
        const float ov[4] = {o.x, o.y, o.z, o.w};
        u16 hh[4], mm[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hh[k] = f2bf(ov[k]);
          const float r1 = ov[k] - bf2f(hh[k]);
          mm[k] = f2bf(r1);
          ll[k] = f2bf(r1 - bf2f(mm[k]));
        }
        const int64_t eo = OUT_BF16 == 3 ? ((int64_t)(e >> 5) * prows + prow) * 32 + (e & 31) : (int64_t)e;
        uint2 pk;
        pk.x = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16); pk.y = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
        *reinterpret_cast<uint2*>((u16*)y + eo) = pk;
        pk.x = (uint32_t)mm[0] | ((uint32_t)mm[1] << 16); pk.y = (uint32_t)mm[2] | ((uint32_t)mm[3] << 16);
        *reinterpret_cast<uint2*>((u16*)y + plane + eo) = pk;
        pk.x = (uint32_t)ll[0] | ((uint32_t)ll[1] << 16); pk.y = (uint32_t)ll[2] | ((uint32_t)ll[3] << 16);
        *reinterpret_cast<uint2*>((u16*)y + 2 * plane + eo) = pk;
      } else if constexpr (OUT_BF16 == 1) {
        uint2 pk;
        pk.x = (uint32_t)f2bf(o.x) | ((uint32_t)f2bf(o.y) << 16);
        pk.y = (uint32_t)f2bf(o.z) | ((uint32_t)f2bf(o.w) << 16);
        *reinterpret_cast<uint2*>((u16*)y + e) = pk;
      } else {
        *reinterpret_cast<float4*>((float*)y + e) = o;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int e = lane + 64 * i;
      const float o = v[i] * w[e] + b[e];
      if constexpr (OUT_BF16 == 2 || OUT_BF16 == 3) {
        const u16 h1 = f2bf(o);
        const float r1 = o - bf2f(h1);
        const u16 m1 = f2bf(r1);
        const int64_t eo = OUT_BF16 == 3 ? ((int64_t)(e >> 5) * prows + prow) * 32 + (e & 31) : (int64_t)e;
        ((u16*)y)[eo] = h1; ((u16*)y)[plane + eo] = m1; ((u16*)y)[2 * plane + eo] = f2bf(r1 - bf2f(m1));
      } else if constexpr (OUT_BF16 == 1) ((u16*)y)[e] = f2bf(o); else ((float*)y)[e] = o;
    }
  }
}

template <int VPL, int OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        void* __restrict__ y, int64_t ldy, int64_t rows,
                                                        float eps, int mode) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[VPL];
  load_row<VPL>(x + row * ldx, lane, v);
  normalize<VPL>(v, eps, mode);
  void* yr = OUT_BF16 == 3 ? y : OUT_BF16 ? (void*)((u16*)y + row * ldy) : (void*)((float*)y + row * ldy);
  store_row_affine<VPL, OUT_BF16>(yr, lane, v, w, b, rows * ldy, row, rows);
}

// K-panel planes (ACX_BF16X3P), TWO rows per wave: with one row per wave a store instruction writes eight separate 64-byte pieces
// (one per 32-column panel: half lines; 0.157 ms per ViT LayerNorm against 0.12 for row-major planes).  Here lane pairs trade
// halves -- the even lane of a pair ends up with eight consecutive elements of row r, the odd lane with the same eight of row r + 1
// -- so that eight lanes write the 128 contiguous bytes rows r, r + 1 occupy in a panel with 16-byte stores.  Same arithmetic.
// TWO (ACX_BF16X2P): the hi and mid planes only (the lo plane is left untouched: a pairs = 3 product does not read it)
// F16 (ACX_F16X2P): two fp16 planes hi | lo (hi = fp16(y), lo = fp16(y - hi)) instead of bf16 planes
template <int VPL, bool TWO = false, bool F16 = false>
__global__ __launch_bounds__(256) void layernorm_panel2_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                               const float* __restrict__ b, u16* __restrict__ y, int64_t rows,
                                                               float eps, int mode) {
  static_assert(VPL % 4 == 0, "16-byte row loads");
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;         // even: rows r0, r0 + 1 share a line in every panel
  if (r0 >= rows) return;
  const bool two = r0 + 1 < rows;
  float va[VPL], vb[VPL];
  load_row<VPL>(x + r0 * ldx, lane, va);
  load_row<VPL>(x + (two ? r0 + 1 : r0) * ldx, lane, vb);
  normalize<VPL>(va, eps, mode);
  normalize<VPL>(vb, eps, mode);
  const int64_t plane = rows * (int64_t)(64 * VPL);
  const int odd = lane & 1;
#pragma unroll
  for (int i = 0; i < VPL / 4; ++i) {
    const int e = 4 * lane + 256 * i;
    const float4 ww = *reinterpret_cast<const float4*>(w + e);
    const float4 bb = *reinterpret_cast<const float4*>(b + e);
    float oa[4] = {va[4 * i] * ww.x + bb.x, va[4 * i + 1] * ww.y + bb.y, va[4 * i + 2] * ww.z + bb.z, va[4 * i + 3] * ww.w + bb.w};
    float ob[4] = {vb[4 * i] * ww.x + bb.x, vb[4 * i + 1] * ww.y + bb.y, vb[4 * i + 2] * ww.z + bb.z, vb[4 * i + 3] * ww.w + bb.w};
    // the pair (2 j, 2 j + 1) holds elements 8 j .. 8 j + 7 of both rows: even keeps row a (own four + the odd lane's four), odd row b
    float o8[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float give = odd ? oa[k] : ob[k];                                    // what the partner needs from this lane
      const float got = __shfl_xor(give, 1, 64);
      o8[k] = odd ? got : oa[k];                                                 // elements 8 j + k
      o8[4 + k] = odd ? ob[k] : got;                                             // elements 8 j + 4 + k
    }
    u16 hh[8], mm[8], ll[8];
    if constexpr (F16) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const uint32_t ph_ = f2h2(o8[k], o8[k + 1]);
        const uint32_t pl_ = f2h2(o8[k] - h2f_lo(ph_), o8[k + 1] - h2f_hi(ph_));
        hh[k] = (u16)(ph_ & 0xffffu); hh[k + 1] = (u16)(ph_ >> 16);
        mm[k] = (u16)(pl_ & 0xffffu); mm[k + 1] = (u16)(pl_ >> 16);
        ll[k] = ll[k + 1] = 0;
      }
    } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      hh[k] = f2bf(o8[k]);
      const float r1 = o8[k] - bf2f(hh[k]);
      mm[k] = f2bf(r1);
      ll[k] = f2bf(r1 - bf2f(mm[k]));
    }
    }
    const int e8 = 8 * (lane >> 1) + 256 * i;
    const int64_t row = r0 + odd;
    if (odd && !two) continue;
    u16* dst = y + ((int64_t)(e8 >> 5) * rows + row) * 32 + (e8 & 31);
#define LNP_PACK(a) make_uint4((uint32_t)a[0] | ((uint32_t)a[1] << 16), (uint32_t)a[2] | ((uint32_t)a[3] << 16), \
                               (uint32_t)a[4] | ((uint32_t)a[5] << 16), (uint32_t)a[6] | ((uint32_t)a[7] << 16))
    *reinterpret_cast<uint4*>(dst) = LNP_PACK(hh);
    *reinterpret_cast<uint4*>(dst + plane) = LNP_PACK(mm);
    if constexpr (!TWO) *reinterpret_cast<uint4*>(dst + 2 * plane) = LNP_PACK(ll);
#undef LNP_PACK
  }
}

template <int VPL>
__global__ __launch_bounds__(256) void vit_embed_kernel(const float* __restrict__ patch_out,
                                                        const float* __restrict__ cls, const float* __restrict__ pos,
                                                        const float* __restrict__ lw, const float* __restrict__ lb,
                                                        float* __restrict__ x, int64_t rows, int T) {
  constexpr int W = 64 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t f = row / (T + 1);
  const int tok = (int)(row - f * (T + 1));
  float v[VPL], p[VPL];
  load_row<VPL>(tok == 0 ? cls : patch_out + (f * T + tok - 1) * W, lane, v);
  load_row<VPL>(pos + (int64_t)tok * W, lane, p);
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] += p[i];
  normalize<VPL>(v, 1e-5f, ACX_NORM_LAYER);
  store_row_affine<VPL, 0>(x + row * W, lane, v, lw, lb);
}

template <int VPL>
__global__ __launch_bounds__(256) void cls_head_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                       const float* __restrict__ lw, const float* __restrict__ lb,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ scores, int64_t rows, int gn, int gl, int seg,
                                                       const int* __restrict__ tile_table) {
  constexpr int E = 64 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float a[VPL], c[VPL];
  load_row<VPL>(x1 + row * E, lane, a);
  load_row<VPL>(x2 + row * E, lane, c);
#pragma unroll
  for (int i = 0; i < VPL; ++i) a[i] = (a[i] + c[i]) * 0.5f;   // torch.stack(chunks).mean(0)
  normalize<VPL>(a, 1e-5f, ACX_NORM_LAYER);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int e = elem_index<VPL>(lane, i);
    dot += (a[i] * lw[e] + lb[e]) * w[e];
  }
  dot = wave_sum(dot) + bias[0];
  if (lane == 0) {
    int64_t dst = row;
    if (tile_table) {   // batch of videos with their own segment sizes: row (tile, n, l) -> base + n * stride + l
      const int grid_sz = gn * gl;
      const int64_t tile = row / grid_sz;
      const int rem = (int)(row - tile * grid_sz);
      const int n = rem / gl, l = rem - n * gl;
      dst = (int64_t)tile_table[2 * tile] + (int64_t)n * tile_table[2 * tile + 1] + l;
    } else if (seg > 0) {  // row is ((b s) n l) -> write at ((b n s l))  (temporal_model.py:69-71)
      const int grid_sz = gn * gl;
      const int64_t per = (int64_t)grid_sz * seg;
      const int64_t bb = row / per, rem = row - bb * per;
      const int s = (int)(rem / grid_sz), rem2 = (int)(rem - (int64_t)s * grid_sz);
      const int n = rem2 / gl, l = rem2 - n * gl;
      dst = ((bb * gn + n) * seg + s) * gl + l;
    }
    scores[dst] = 1.f / (1.f + expf(-dot));
  }
}

}  // namespace

#define COMMA ,
#define DISPATCH_VPL(D, CALL)                          \
  switch ((D) / 64) {                                  \
    case 1: { constexpr int V = 1; CALL; } break;      \
    case 2: { constexpr int V = 2; CALL; } break;      \
    case 4: { constexpr int V = 4; CALL; } break;      \
    case 8: { constexpr int V = 8; CALL; } break;      \
    case 12: { constexpr int V = 12; CALL; } break;    \
    case 16: { constexpr int V = 16; CALL; } break;    \
    default: return acx_fail(ctx, ACX_E_UNSUPPORTED, "row width %s%ld not in {64,128,256,512,768,1024}", "", (long)(D)); \
  }

extern "C" int acx_layernorm(acx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* b,
                             void* y, int64_t ldy, int32_t y_dtype, int64_t rows, int32_t D, float eps,
                             int32_t mode, void* stream) {
  if (!x || !w || !b || !y) return acx_fail(ctx, ACX_E_BADARG, "acx_layernorm: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (D % 64 || ldx % 4 || ldy % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_layernorm: D%%64 / ld%%4%s");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  if (y_dtype == ACX_F16X2P) {         // two fp16 planes in K-panel layout (the ViT width only: the ACX_PREC_F16X3 driver)
    if (ldy != D || D != 768 || ((uintptr_t)y & 15) || ((rows * (int64_t)D * 2) & 15))
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_layernorm: ACX_F16X2P needs ldy == D == 768 and 16-byte aligned planes%s");
    layernorm_panel2_kernel<12, true, true><<<dim3((unsigned)((rows + 7) / 8)), block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode);
    ACX_CHECK_LAUNCH(ctx, "acx_layernorm");
    return ACX_OK;
  }
  if (y_dtype == ACX_BF16X3P || y_dtype == ACX_BF16X2P) {   // three planes in K-panel layout [D / 32][rows][32] each (ldy == D), y + p * rows * ldy
    if (ldy != D || D % 256) return acx_fail(ctx, ACX_E_BADARG, "acx_layernorm: ACX_BF16X3P needs ldy == D, D %% 256 == 0%s");
    if (!((uintptr_t)y & 15) && !((rows * (int64_t)D * 2) & 15)) {                // 16-byte stores: two rows per wave
      const dim3 grid2((unsigned)((rows + 7) / 8));
      if (y_dtype == ACX_BF16X2P && D == 768) {                                   // (hi and mid planes only: the ViT width)
        layernorm_panel2_kernel<12, true><<<grid2, block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode);
        ACX_CHECK_LAUNCH(ctx, "acx_layernorm");
        return ACX_OK;
      }
      switch (D / 64) {
        case 4: layernorm_panel2_kernel<4><<<grid2, block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode); break;
        case 8: layernorm_panel2_kernel<8><<<grid2, block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode); break;
        case 12: layernorm_panel2_kernel<12><<<grid2, block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode); break;
        case 16: layernorm_panel2_kernel<16><<<grid2, block, 0, s>>>(x, ldx, w, b, (u16*)y, rows, eps, mode); break;
        default: return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_layernorm: ACX_BF16X3P needs D in {256, 512, 768, 1024}%s");
      }
    } else
    DISPATCH_VPL(D, layernorm_kernel<V COMMA 3><<<grid COMMA block COMMA 0 COMMA s>>>(x, ldx, w, b, y, ldy, rows, eps, mode));
  } else if (y_dtype == ACX_BF16X3) {  // three dense planes [rows, ldy] each, y + p * rows * ldy
    DISPATCH_VPL(D, layernorm_kernel<V COMMA 2><<<grid COMMA block COMMA 0 COMMA s>>>(x, ldx, w, b, y, ldy, rows, eps, mode));
  } else if (y_dtype == ACX_BF16) {
    DISPATCH_VPL(D, layernorm_kernel<V COMMA 1><<<grid COMMA block COMMA 0 COMMA s>>>(x, ldx, w, b, y, ldy, rows, eps, mode));
  } else {
    DISPATCH_VPL(D, layernorm_kernel<V COMMA 0><<<grid COMMA block COMMA 0 COMMA s>>>(x, ldx, w, b, y, ldy, rows, eps, mode));
  }
  ACX_CHECK_LAUNCH(ctx, "acx_layernorm");
  return ACX_OK;
}

extern "C" int acx_vit_embed(acx_ctx* ctx, const float* patch_out, const float* cls, const float* pos,
                             const float* ln_w, const float* ln_b, float* x, int32_t F, int32_t T, int32_t W,
                             void* stream) {
  if (!patch_out || !cls || !pos || !ln_w || !ln_b || !x) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_embed: null pointer%s");
  const int64_t rows = (int64_t)F * (T + 1);
  if (rows <= 0) return ACX_OK;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  DISPATCH_VPL(W, vit_embed_kernel<V><<<grid COMMA block COMMA 0 COMMA s>>>(patch_out, cls, pos, ln_w, ln_b, x, rows, T));
  ACX_CHECK_LAUNCH(ctx, "acx_vit_embed");
  return ACX_OK;
}

extern "C" int acx_cls_head(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w, const float* ln_b,
                            const float* lin_w, const float* lin_b, float* scores, int64_t rows, int32_t E,
                            int32_t gn, int32_t gl, int32_t seg, void* stream) {
  if (!x1 || !x2 || !ln_w || !ln_b || !lin_w || !lin_b || !scores) return acx_fail(ctx, ACX_E_BADARG, "acx_cls_head: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (seg > 0 && (gn <= 0 || gl <= 0 || rows % ((int64_t)gn * gl * seg)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_cls_head: rows not a multiple of gn*gl*seg%s");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  DISPATCH_VPL(E, cls_head_kernel<V><<<grid COMMA block COMMA 0 COMMA s>>>(x1, x2, ln_w, ln_b, lin_w, lin_b, scores, rows, gn, gl, seg, nullptr));
  ACX_CHECK_LAUNCH(ctx, "acx_cls_head");
  return ACX_OK;
}

extern "C" int acx_cls_head_tiles(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w, const float* ln_b,
                                  const float* lin_w, const float* lin_b, float* scores, int64_t rows, int32_t E, int32_t gn,
                                  int32_t gl, const int32_t* tile_table, void* stream) {
  if (!x1 || !x2 || !ln_w || !ln_b || !lin_w || !lin_b || !scores || !tile_table)
    return acx_fail(ctx, ACX_E_BADARG, "acx_cls_head_tiles: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (gn <= 0 || gl <= 0 || rows % ((int64_t)gn * gl)) return acx_fail(ctx, ACX_E_BADARG, "acx_cls_head_tiles: rows %% (gn*gl) != 0%s");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  DISPATCH_VPL(E, cls_head_kernel<V><<<grid COMMA block COMMA 0 COMMA s>>>(x1, x2, ln_w, ln_b, lin_w, lin_b, scores, rows, gn, gl, 0, tile_table));
  ACX_CHECK_LAUNCH(ctx, "acx_cls_head_tiles");
  return ACX_OK;
}
