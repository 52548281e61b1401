// acx_attention -- exact-f32 multi-head attention for head dim 64 and short sequences (<= 224):
// CLIP ViT-B/16 (L = 197, 12 heads) and the CLIP text transformer (L = 77, causal, 8 heads).
// Replaces the core of nn.MultiheadAttention as called from clip/model.py:206-212; the reference
// materialises (batch*heads, L, L) score tensors in HBM, here scores never leave registers.
//
// CDNA4 design (one workgroup = one (sequence, head), 4 wavefronts, one per SIMD):
//   * K and V of the head are staged ONCE in LDS as f32 (L=197: 2 x 224 x 64 x 4 B = 112 KiB of
//     the CU's 160 KiB).  K rows are padded to 272 B so the 16-lane groups of ds_read_b128 hit
//     16 distinct 16-B slots; V is read with ds_read_b32 (32 consecutive floats per half-wave).
//   * each wave owns 32-query blocks.  Scores are computed TRANSPOSED, S^T = K Q^T, with
//     v_mfma_f32_32x32x2_f32 (exact f32): in the 32x32 C layout a lane then holds one QUERY
//     column (col = lane&31) and 16 keys per tile, so the softmax row reduction is in-register
//     plus ONE cross-half exchange, and P^T registers are directly the A operand of the P.V MFMA
//     (lane half h of register r holds key (r&3)+8(r>>2)+4h -- the V row each half fetches).
//   * K-permutation trick as in acx_gemm: one ds_read_b128 of K feeds four MFMAs.
#include "acx_internal.h"

#include <stdlib.h>

namespace {

constexpr int KROW = 68;   // floats per K row in LDS (64 + 4 pad = 272 B)
constexpr int VROW = 64;

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_kernel(const float* __restrict__ qkv, int64_t ldqkv,
                                                      float* __restrict__ out, int64_t ldo, int L, int heads,
                                                      int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + NT * 32 * KROW;
  float* sL = sV + NT * 32 * VROW;   // [NW waves][32] row sums

  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int W = heads * 64;
  const float* base = qkv + (int64_t)b * L * ldqkv + h * 64;
  const int t = threadIdx.x;

  // ---- stage K, V (rows >= L zero-filled): ALL global loads are issued before the first LDS write so the
  // staging costs one memory round trip, not one per iteration
  constexpr int NSTG = (NT * 32 * 16 + NW * 64 - 1) / (NW * 64);
  float4 stk[NSTG], stv[NSTG];
#pragma unroll
  for (int j = 0; j < NSTG; ++j) {
    const int i = t + j * NW * 64;
    const int row = i >> 4, c4 = i & 15;
    stk[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    stv[j] = stk[j];
    if (i < NT * 32 * 16 && row < L) {
      const float* p = base + (int64_t)row * ldqkv + 4 * c4;
      stk[j] = *reinterpret_cast<const float4*>(p + W);
      stv[j] = *reinterpret_cast<const float4*>(p + 2 * W);
    }
  }
#pragma unroll
  for (int j = 0; j < NSTG; ++j) {
    const int i = t + j * NW * 64;
    const int row = i >> 4, c4 = i & 15;
    if (i < NT * 32 * 16) {
      *reinterpret_cast<float4*>(sK + row * KROW + 4 * c4) = stk[j];
      *reinterpret_cast<float4*>(sV + row * VROW + 4 * c4) = stv[j];
    }
  }
  __syncthreads();

  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int nqb = (L + 31) / 32;

  for (int qb = wave; qb < nqb; qb += NW) {
    const int q0 = qb * 32;
    // ---- Q fragment (B operand of S^T = K.Q^T): lane (q=li, half hh) holds chunks (2c+hh)
    float4 qf[8];
    {
      const int q = q0 + li;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < L) v = *reinterpret_cast<const float4*>(base + (int64_t)q * ldqkv + (2 * c + hh) * 4);
        v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;   // 64^-0.5, exact
        qf[c] = v;
      }
    }
    const int t_hi = causal ? min(NT, (q0 + 31) / 32 + 1) : NT;   // key tiles that can be visible

    f32x16 st[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
      if (kt < t_hi) {
        const float* kp = sK + (kt * 32 + li) * KROW + hh * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float4 kf = *reinterpret_cast<const float4*>(kp + c * 8);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c].x, st[kt], 0, 0, 0);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c].y, st[kt], 0, 0, 0);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c].z, st[kt], 0, 0, 0);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c].w, st[kt], 0, 0, 0);
        }
      }
    }
    // ---- softmax over keys for query (q0 + li): keys live in registers (kt, r) and lane half hh
    const int qglob = q0 + li;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const bool ok = key < L && (!causal || key <= qglob) && kt < t_hi;
        st[kt][r] = ok ? st[kt][r] : -INFINITY;
        mx = fmaxf(mx, st[kt][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __expf(st[kt][r] - mx);   // exp(-inf) = 0 for masked keys
        st[kt][r] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    if (hh == 0) sL[wave * 32 + li] = sum;

    // ---- O = P.V : A operand = P^T registers as they are, B operand = V rows from LDS
    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt < t_hi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float v0 = sV[key * VROW + li];
          const float v1 = sV[key * VROW + 32 + li];
          o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kt][r], v0, o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kt][r], v1, o[1], 0, 0, 0);
        }
      }
    }
    // ---- normalise and store: o[dt] C layout: col = d = 32*dt + li, row = query (r&3)+8(r>>2)+4hh
    // (sL written by this wave above; same-wave LDS ops are ordered, no barrier needed)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const int q = q0 + ql;
      if (q < L) {
        const float inv = 1.f / sL[wave * 32 + ql];
        float* op = out + ((int64_t)b * L + q) * ldo + h * 64 + li;
        op[0] = o[0][r] * inv;
        op[32] = o[1][r] * inv;
      }
    }
  }
}


// CLS-only attention for the LAST ViT layer: only token 0 of every sequence is consumed downstream
// (clip/model.py:285 takes x[:, 0, :]), so only query row 0 needs softmax(q k^T) v.  One wavefront per
// (sequence, head): lanes own keys for the scores, then own output dims for the weighted sum of V.
__global__ __launch_bounds__(256) void attn_cls_kernel(const float* __restrict__ qkv, int64_t ldqkv, float* __restrict__ out,
                                                       int64_t ldo, int batch, int L, int heads) {
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= (int64_t)batch * heads) return;
  const int b = (int)(idx / heads), h = (int)(idx % heads);
  const int W = heads * 64;
  const float* base = qkv + (int64_t)b * L * ldqkv + h * 64;
  float q[64];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(base + 4 * c);
    q[4 * c] = v.x * 0.125f; q[4 * c + 1] = v.y * 0.125f; q[4 * c + 2] = v.z * 0.125f; q[4 * c + 3] = v.w * 0.125f;
  }
  float sc[4];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = lane + 64 * u;
    float acc = -INFINITY;
    if (j < L) {
      const float* kp = base + (int64_t)j * ldqkv + W;
      acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(kp + 4 * c);
        acc += q[4 * c] * v.x + q[4 * c + 1] * v.y + q[4 * c + 2] * v.z + q[4 * c + 3] * v.w;
      }
    }
    sc[u] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) { sc[u] = __expf(sc[u] - mx); sum += sc[u]; }
  sum = wave_sum(sum);
  float o = 0.f;
  for (int j = 0; j < L; ++j) {
    const int u_ = j >> 6;
    const float mine = u_ == 0 ? sc[0] : (u_ == 1 ? sc[1] : (u_ == 2 ? sc[2] : sc[3]));
    const float p = __shfl(mine, j & 63, 64);
    o += p * base[(int64_t)j * ldqkv + 2 * W + lane];
  }
  out[(int64_t)b * ldo + h * 64 + lane] = o / sum;
}

}  // namespace

extern "C" int acx_attention(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                             int32_t batch, int32_t L, int32_t heads, int32_t causal, void* stream) {
  if (!qkv || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_attention: null pointer%s");
  if (batch <= 0) return ACX_OK;
  if (L <= 0 || L > 224 || heads <= 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention: need 0 < L <= 224%s");
  if (ldqkv % 4 || ((uintptr_t)qkv & 15)) return acx_fail(ctx, ACX_E_BADARG, "acx_attention: qkv must be 16-byte aligned, ld%%4==0%s");
  const int nt = (L + 31) / 32;
  static const bool nw4 = getenv("ACX_ATTN_NW4") != nullptr;
  const dim3 grid((unsigned)(batch * heads));
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, (hipStream_t)stream);
#define ACX_ATTN(NT, NW)                                                                           \
  do {                                                                                             \
    const size_t lds = (size_t)NT * 32 * (KROW + VROW) * 4 + NW * 32 * 4;                          \
    static bool done = false;                                                                      \
    if (!done) {                                                                                   \
      (void)hipFuncSetAttribute((const void*)attn_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      done = true;                                                                                 \
    }                                                                                              \
    hipLaunchKernelGGL((attn_kernel<NT, NW>), grid, dim3(NW * 64), lds, s, qkv, ldqkv, out, ldo, L, heads, causal); \
  } while (0)
  switch (nt) {
    // one wave per 32-query block where it fits; >4 blocks -> 8 waves (2 per SIMD: one wave's softmax and
    // LDS phases overlap the other's MFMAs)
    case 1: ACX_ATTN(1, 4); break;
    case 2: ACX_ATTN(2, 4); break;
    case 3: ACX_ATTN(3, 4); break;
    case 4: ACX_ATTN(4, 4); break;
    case 5: ACX_ATTN(5, 8); break;
    case 6: ACX_ATTN(6, 8); break;
    default:
      if (nw4) ACX_ATTN(7, 4); else ACX_ATTN(7, 8);
      break;
  }
#undef ACX_ATTN
  ACX_CHECK_LAUNCH(ctx, "acx_attention");
  return ACX_OK;
}

extern "C" int acx_attention_cls(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo, int32_t batch,
                                 int32_t L, int32_t heads, void* stream) {
  if (!qkv || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_cls: null pointer%s");
  if (batch <= 0) return ACX_OK;
  if (L <= 0 || L > 256 || heads <= 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention_cls: need 0 < L <= 256%s");
  if (ldqkv % 4 || ((uintptr_t)qkv & 15)) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_cls: alignment%s");
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, s);
  const int64_t n = (int64_t)batch * heads;
  hipLaunchKernelGGL(attn_cls_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, qkv, ldqkv, out, ldo, batch, L, heads);
  ACX_CHECK_LAUNCH(ctx, "acx_attention_cls");
  return ACX_OK;
}
