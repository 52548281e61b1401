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
  int* vflag = reinterpret_cast<int*>(sL + NW * 32);   // STAGGER: number of waves that have finished staging V

  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int W = heads * 64;
  const float* base = qkv + (int64_t)b * L * ldqkv + h * 64;
  const int t = threadIdx.x;

  // ---- stage K, V (rows >= L zero-filled): ALL global loads are issued before the first LDS write so the
  // staging costs one memory round trip, not one per iteration.
  // STAGGER (8 waves, one query block each): only K is staged up front; the second wave of every SIMD (waves 4-7)
  // then stages V while the first (waves 0-3) is already in QK^T, and one barrier in front of P.V publishes V.
  // The two waves of a SIMD thereby run out of phase -- one's softmax (VALU) under the other's MFMAs -- instead of
  // both doing QK^T, then both softmax with the matrix pipe idle, then both P.V.  V is published through an LDS
  // counter, not a barrier, so the phase shift survives until the end of the block.
  constexpr bool STAGGER = NW == 8;
  constexpr int NSTG = (NT * 32 * 16 + NW * 64 - 1) / (NW * 64);
  {
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
        if constexpr (!STAGGER) stv[j] = *reinterpret_cast<const float4*>(p + 2 * W);
      }
    }
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      const int i = t + j * NW * 64;
      const int row = i >> 4, c4 = i & 15;
      if (i < NT * 32 * 16) {
        *reinterpret_cast<float4*>(sK + row * KROW + 4 * c4) = stk[j];
        if constexpr (!STAGGER) *reinterpret_cast<float4*>(sV + row * VROW + 4 * c4) = stv[j];
      }
    }
  }
  if constexpr (STAGGER) { if (t == 0) *vflag = 0; }
  __syncthreads();
  if constexpr (STAGGER) {
    if (t >= 256) {
      constexpr int NSV = (NT * 32 * 16 + 255) / 256;
      float4 stv[NSV];
#pragma unroll
      for (int j = 0; j < NSV; ++j) {
        const int i = (t - 256) + j * 256;
        const int row = i >> 4, c4 = i & 15;
        stv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < NT * 32 * 16 && row < L) stv[j] = *reinterpret_cast<const float4*>(base + (int64_t)row * ldqkv + 4 * c4 + 2 * W);
      }
#pragma unroll
      for (int j = 0; j < NSV; ++j) {
        const int i = (t - 256) + j * 256;
        if (i < NT * 32 * 16) *reinterpret_cast<float4*>(sV + (i >> 4) * VROW + 4 * (i & 15)) = stv[j];
      }
      // publish (LDS ops of a wave complete in order; the counter is bumped after this wave's V writes) and
      // fall behind the first wave of this SIMD by about one QK^T phase
      __threadfence_block();
      if ((t & 63) == 0) atomicAdd(vflag, 1);
      __builtin_amdgcn_s_sleep(100);     // ~6.4k cycles; measured optimum (0 / 6.4k / 12.8k / 19.2k: -2.5 / -3.9 / -2.4 / 0 %)
    }
  }

  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int nqb = (L + 31) / 32;

  for (int qb = wave; qb < (STAGGER ? NW : nqb); qb += NW) {      // STAGGER: exactly one trip per wave (nqb <= 8)
    const bool has = qb < nqb;
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
    const int t_hi = !has ? 0 : (causal ? min(NT, (q0 + 31) / 32 + 1) : NT);   // key tiles that can be visible (none: idle wave)

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

    if constexpr (STAGGER) {                     // V (staged by waves 4-7 meanwhile) is complete: no barrier, the waves stay out of phase
      while (*reinterpret_cast<volatile int*>(vflag) < 4) __builtin_amdgcn_s_sleep(2);
    }
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


// =====================================================================================================
// attn16_kernel -- exact-f32 attention for the ViT sequence (non-causal, 129 <= L <= 256; L = 197 for ViT-B/16).
//
// What limited attn_kernel<7,8> (0.43 of the f32 MFMA roof): 197 tokens padded to 7 x 32 (29 % padded MFMA work),
// 7 query blocks on 4 SIMDs, and ONE 112 KiB workgroup per CU, so nothing overlaps a workgroup's K/V staging,
// softmax or store phases.  This kernel
//   * tiles with v_mfma_f32_16x16x4_f32: 197 -> 13 x 16 = 208 (11 % padding); a wave owns TWO 16-query tiles (one
//     K / V fragment read feeds both), 7 waves per (frame, head) workgroup;
//   * streams K and V through LDS in 32-key chunks (16 KB per stage, two stages) by LDS-DMA (global_load_lds_dwordx4,
//     bank swizzle of K on the SOURCE address), flash-style online softmax in between -- 32 KB of LDS and <= 128 VGPRs,
//     so TWO workgroups share a CU: one workgroup's DMA waits, barriers, softmax VALU work and output stores sit
//     under the other's MFMAs;
//   * keeps the transposed-score trick: S^T = K Q^T puts a lane's QUERY in the MFMA column (lane & 15), so row
//     max / sum are per-lane scalars (+ two cross-row exchanges, v_permlane16/32_swap: no LDS), P^T registers are
//     directly the B operand of O^T = V^T P^T, and with the output-column permutation d = 4 i + e (MFMA e of a group
//     computes output tile e) one ds_read_b128 of V feeds four MFMAs per query tile and a lane ends up with 16
//     CONTIGUOUS floats of its query's output row.
// Online softmax changes the summation order and the scaling sequence only (exp(s - m) rescaled by exp(m - m')):
// differences to the two-pass form are f32 round-off (tests: 3e-6 against fp64).
#ifndef ACX_A16_KT
#define ACX_A16_KT 2
#endif
#ifndef ACX_A16_ABL
#define ACX_A16_ABL 0                               // timing ablations (wrong results), bit mask: 1 no softmax, 2 no chunk
#endif                                              // barrier, 4 no LDS fragment reads, 8 no DMA
constexpr int A16_ABL = ACX_A16_ABL;
constexpr int A16_KT = ACX_A16_KT;                  // 16-key tiles per staged chunk
constexpr int A16_CHUNK = 16 * A16_KT;              // keys per staged chunk
constexpr int A16_OP_B = A16_CHUNK * 256;           // one operand chunk: rows of 64 f32
constexpr int A16_STAGE_B = 2 * A16_OP_B;           // K | V

typedef __attribute__((address_space(3))) void a16_lds_t;

// max / sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48): v_permlane16_swap exchanges odd rows of
// its first operand with even rows of the second, v_permlane32_swap the upper half of the first with the lower half
// of the second; with both operands = x the two results hold x of this row pair's even and odd member.
typedef unsigned a16_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float a16_rowmax(float x) {
  a16_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float a16_rowsum(float x) {
  a16_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// One wave's share of a workgroup's items.  HB: the wave owns two query tiles (ta, tb) or one (ta).
// x3plane > 0: the output is written as three bf16 planes hi | mid | lo (ACX_BF16X3; plane p at (u16*)out + p * x3plane): the A
// operand of the out-projection in ACX_PREC_F32X6 mode
// prows > 0: the planes are in K-panel layout (ACX_BF16X3P: [ldo / 32][prows][32])
__device__ __forceinline__ void a16_store4(float* out, int64_t row, int col, int64_t ldo, int64_t x3plane, int64_t prows, float a, float b,
                                           float c, float d) {
  const int64_t off = prows > 0 ? ((int64_t)(col >> 5) * prows + row) * 32 + (col & 31) : row * ldo + col;
  if (x3plane > 0) {
    const float ov[4] = {a, b, c, d};
    u16 hh[4], mm[4], ll[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hh[k] = f2bf(ov[k]);
      const float r1 = ov[k] - bf2f(hh[k]);
      mm[k] = f2bf(r1);
      ll[k] = f2bf(r1 - bf2f(mm[k]));
    }
    u16* o3 = reinterpret_cast<u16*>(out) + off;
    uint2 pk;
    pk.x = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16); pk.y = (uint32_t)hh[2] | ((uint32_t)hh[3] << 16);
    *reinterpret_cast<uint2*>(o3) = pk;
    pk.x = (uint32_t)mm[0] | ((uint32_t)mm[1] << 16); pk.y = (uint32_t)mm[2] | ((uint32_t)mm[3] << 16);
    *reinterpret_cast<uint2*>(o3 + x3plane) = pk;
    pk.x = (uint32_t)ll[0] | ((uint32_t)ll[1] << 16); pk.y = (uint32_t)ll[2] | ((uint32_t)ll[3] << 16);
    *reinterpret_cast<uint2*>(o3 + 2 * x3plane) = pk;
  } else {
    *reinterpret_cast<float4*>(out + off) = make_float4(a, b, c, d);
  }
}

template <bool HB, int NW>
__device__ __forceinline__ void a16_run(const float* __restrict__ qkv, int64_t ldqkv, float* __restrict__ out, int64_t ldo,
                                        int L, int heads, int nitems, char* smem, int ta, int tb, int64_t x3plane, int64_t prows) {
  const int W = heads * 64;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int nw = NW;                            // waves per workgroup
  const int qi = lane & 15, g = lane >> 4;
  const int nkt = (L + 15) >> 4;                    // 16-key tiles ( = 16-query tiles)
  const int nch = (nkt + A16_KT - 1) / A16_KT;      // chunks

  // ---- DMA plan: a chunk is 8 KT wave-instructions of 1 KB (4 rows x 256 B): first the K rows, then the V rows;
  // wave w issues instructions w, w + nw, ... (8 KT instructions per chunk, nw = 8 waves).  K: LDS slot p of row r holds source slot p ^ (r & 15).
  // The DMA goes through inline asm: hipcc (ROCm 7.2) treats a global_load_lds it can see as a pending LDS write that may
  // alias EVERY later ds_read and drains it (s_waitcnt vmcnt(0)) in front of the first fragment read of the same
  // iteration -- the next chunk's DMA would never overlap this chunk's MFMAs.  An asm statement is opaque to that
  // bookkeeping; its completion is counted by hand (vmcnt(0) + barrier at the end of each chunk; the loop has no
  // other VMEM operation).  M0 (LDS destination base, wave-uniform) is written and restored inside the statement.
  const unsigned lds0 = (unsigned)(uintptr_t)(a16_lds_t*)smem;
#define A16_DMA(gptr, ldsaddr)                                                                                   \
  do {                                                                                                           \
    unsigned keep_;                                                                                              \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                                           \
  } while (0)
#define A16_ISSUE1(stage, ch, i_)                                                                                \
  do {                                                                                                           \
    const int ii_ = (i_);                                           /* wave-uniform */                           \
    if (ii_ < 8 * A16_KT && !(A16_ABL & 8)) {                                                                      \
      const int r_ = 4 * (ii_ % (4 * A16_KT)) + g;                                                               \
      const int e_ = ii_ < 4 * A16_KT ? W + 4 * (qi ^ (r_ & 15)) : 2 * W + 4 * qi;                                \
      A16_DMA(bp_ + (int64_t)min((ch) * A16_CHUNK + r_, L - 1) * ldqkv + e_,                                     \
              lds0 + (stage) * A16_STAGE_B + ii_ * 1024);                                                        \
    }                                                                                                            \
  } while (0)
#define A16_ISSUE(bp, stage, ch)                                                                                 \
  do { const float* bp_ = (bp);                                                                                  \
       _Pragma("unroll") for (int j_ = 0; j_ < (8 * A16_KT + nw - 1) / nw; ++j_) A16_ISSUE1(stage, ch, wave + j_ * nw); } while (0)
  // K fragment address: row (tile-local key = lane & 15), slot (4 c4 + g) ^ (key & 15) = ((g ^ qi) ^ 4 c4): the four
  // slots differ by an XOR of the byte offset with 64 c4.  V: row (4 g + r), slot qi (d = 4 qi ..), linear image.
  const int ka = qi * 256 + ((g ^ qi) * 16);
  const int va = A16_OP_B + (4 * g) * 256 + qi * 16;

  // ---- persistent over (frame, head) items; the first chunk of the NEXT item is in flight while this item's output
  // is normalised and stored
  int item = blockIdx.x;
  if (item < nitems) A16_ISSUE(qkv + (int64_t)(item / heads) * L * ldqkv + (item % heads) * 64, 0, 0);
  for (; item < nitems; item += gridDim.x) {
    const int b = item / heads, h = item % heads;
    const float* base = qkv + (int64_t)b * L * ldqkv + h * 64;
    // ---- Q fragments (B operand of S^T = K Q^T): lane (query qi, k-group g) holds d = 16 c4 + 4 g + (0..3)
    float4 qa[4], qb[4];
    {
      const float* pa = base + (int64_t)min(16 * ta + qi, L - 1) * ldqkv + 4 * g;
      const float* pb = base + (int64_t)min(16 * tb + qi, L - 1) * ldqkv + 4 * g;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        float4 v = *reinterpret_cast<const float4*>(pa + 16 * c4);
        v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;     // 64^-0.5, exact
        qa[c4] = v;
        if constexpr (HB) {
          v = *reinterpret_cast<const float4*>(pb + 16 * c4);
          v.x *= 0.125f; v.y *= 0.125f; v.z *= 0.125f; v.w *= 0.125f;
          qb[c4] = v;
        }
      }
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 oa0 = z4, oa1 = z4, oa2 = z4, oa3 = z4;                         // O^T tiles e = 0..3 (d = 16 g + 4 reg + e)
    f32x4 ob0 = z4, ob1 = z4, ob2 = z4, ob3 = z4;
    float ma = -INFINITY, la = 0.f, mb = -INFINITY, lb = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int ch = 0; ch < nch; ++ch) {
      if (ch + 1 < nch) A16_ISSUE(base, (ch + 1) & 1, ch + 1);
      const char* sS = smem + (ch & 1) * A16_STAGE_B;
      // ---- two 16-key tiles per chunk, each: S^T tiles of the wave's query tiles -> online softmax -> O^T += V^T P^T
#define A16_QK(c4, off, toff)                                                                                   \
  do {                                                                                                           \
    const float4 k_ = (A16_ABL & 4) ? qa[c4] : *reinterpret_cast<const float4*>(sS + (ka ^ (off)) + (toff));      \
    sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.x, qa[c4].x, sa, 0, 0, 0);                                       \
    if constexpr (HB) sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.x, qb[c4].x, sb, 0, 0, 0);                     \
    sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.y, qa[c4].y, sa, 0, 0, 0);                                       \
    if constexpr (HB) sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.y, qb[c4].y, sb, 0, 0, 0);                     \
    sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.z, qa[c4].z, sa, 0, 0, 0);                                       \
    if constexpr (HB) sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.z, qb[c4].z, sb, 0, 0, 0);                     \
    sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.w, qa[c4].w, sa, 0, 0, 0);                                       \
    if constexpr (HB) sb = __builtin_amdgcn_mfma_f32_16x16x4f32(k_.w, qb[c4].w, sb, 0, 0, 0);                     \
  } while (0)
#define A16_SOFTMAX(S, M, LS, O0, O1, O2, O3)                                                                    \
  do {                                                                                                           \
    const float mx_ = a16_rowmax(fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3])));                                   \
    const float mn_ = fmaxf(M, mx_);                                 /* finite: every tile holds >= 1 valid key */ \
    const float al_ = __expf(M - mn_);                               /* first tile: exp(-inf) = 0 */             \
    M = mn_;                                                                                                     \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) S[r] = __expf(S[r] - mn_);   /* exp(-inf) = 0 for masked keys */ \
    LS = LS * al_ + ((S[0] + S[1]) + (S[2] + S[3]));                                                             \
    if (!__all(al_ == 1.f)) {                                     /* x * 1 == x: skipping is bit-identical */     \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) { O0[r] *= al_; O1[r] *= al_; O2[r] *= al_; O3[r] *= al_; }  \
    }                                                                                                            \
  } while (0)
#define A16_PV(r, toff)                                                                                         \
  do {                                                                                                           \
    const float4 v_ = (A16_ABL & 4) ? qa[r] : *reinterpret_cast<const float4*>(sS + va + (toff) + (r) * 256);      \
    oa0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.x, sa[r], oa0, 0, 0, 0);                                        \
    if constexpr (HB) ob0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.x, sb[r], ob0, 0, 0, 0);                      \
    oa1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.y, sa[r], oa1, 0, 0, 0);                                        \
    if constexpr (HB) ob1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.y, sb[r], ob1, 0, 0, 0);                      \
    oa2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.z, sa[r], oa2, 0, 0, 0);                                        \
    if constexpr (HB) ob2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.z, sb[r], ob2, 0, 0, 0);                      \
    oa3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.w, sa[r], oa3, 0, 0, 0);                                        \
    if constexpr (HB) ob3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v_.w, sb[r], ob3, 0, 0, 0);                      \
  } while (0)
#define A16_TILE(toff, kbase)                                                                                    \
  do {                                                                                                           \
    f32x4 sa = z4, sb = z4;                                                                                      \
    A16_QK(0, 0, toff); A16_QK(1, 64, toff); A16_QK(2, 128, toff); A16_QK(3, 192, toff);                         \
    if ((kbase) + 16 > L) {                                           /* only the last tile holds keys >= L */   \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                            \
        const bool ok_ = (kbase) + 4 * g + r < L;                                                                \
        sa[r] = ok_ ? sa[r] : -INFINITY; sb[r] = ok_ ? sb[r] : -INFINITY;                                        \
      }                                                                                                          \
    }                                                                                                            \
    if constexpr (!(A16_ABL & 1)) {                                                                                \
      A16_SOFTMAX(sa, ma, la, oa0, oa1, oa2, oa3);                                                               \
      if constexpr (HB) A16_SOFTMAX(sb, mb, lb, ob0, ob1, ob2, ob3);                                             \
    }                                                                                                            \
    A16_PV(0, toff); A16_PV(1, toff); A16_PV(2, toff); A16_PV(3, toff);                                          \
  } while (0)
      A16_TILE(0, A16_CHUNK * ch);
#pragma unroll
      for (int j = 1; j < A16_KT; ++j)
        if (A16_KT * ch + j < nkt) A16_TILE(j * 16 * 256, A16_CHUNK * ch + 16 * j);
#undef A16_TILE
#undef A16_PV
#undef A16_SOFTMAX
#undef A16_QK
      // next chunk landed (this wave's DMAs) and everybody is done reading this stage
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if constexpr (!(A16_ABL & 2)) __builtin_amdgcn_s_barrier();
    }
    // every wave is past its last fragment read (barrier above): stage 0 is free for the next item's first chunk
    {
      const int nxt = item + (int)gridDim.x;
      if (nxt < nitems) A16_ISSUE(qkv + (int64_t)(nxt / heads) * L * ldqkv + (nxt % heads) * 64, 0, 0);
    }
    // ---- normalise and store: lane (query qi, g) holds O[query][16 g + 4 reg + e]
    {
      const float inv = 1.f / a16_rowsum(la);
      const int q = 16 * ta + qi;
      if (q < L) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a16_store4(out, (int64_t)b * L + q, h * 64 + 16 * g + 4 * r, ldo, x3plane, prows, oa0[r] * inv, oa1[r] * inv, oa2[r] * inv, oa3[r] * inv);
      }
    }
    if constexpr (HB) {
      const float inv = 1.f / a16_rowsum(lb);
      const int q = 16 * tb + qi;
      if (q < L) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a16_store4(out, (int64_t)b * L + q, h * 64 + 16 * g + 4 * r, ldo, x3plane, prows, ob0[r] * inv, ob1[r] * inv, ob2[r] * inv, ob3[r] * inv);
      }
    }
  }   // item loop
#undef A16_ISSUE
#undef A16_ISSUE1
#undef A16_DMA
}

// 8 waves per workgroup; nkt (9..16) query tiles: nd = nkt - 8 waves own two tiles, the others one (L = 197: waves 0..4
// two tiles, 5..7 one).  Measured and rejected on top of this: 13 one-tile waves at <= 64 VGPRs (spills, 0.86 ms), chunks
// of 16 / 64 keys (no change: the chunk barrier is not the limit), s_setprio around the MFMA groups, rotating the
// two-tile waves between co-resident workgroups, dealing the two-tile roles by the SIMD id the waves report (s_getreg
// HW_ID; waves w and w + 4 share a SIMD, tools/probes/wave_simd.hip) so that a CU's SIMDs carry 7/6/7/6 tiles (no
// change), and software-pipelining the softmax inside each wave (scores of key tile j + 1 issued next to the softmax
// of tile j, one 16-key tile per stage, ring of four: 0.785 ms, not faster; hipcc's scheduler clusters the MFMAs
// and sched_group_barrier patterns made it 0.81).  Timing ablations (ACX_A16_ABL, profiles/r02_attn_ablation.txt):
// no single phase is the limit -- softmax -11 %, DMA -8 %, chunk barrier -6 %, LDS fragment reads -4 %, and with ALL of
// them removed (MFMAs, Q loads and output stores only) the kernel still takes 0.60 ms = 0.73 of the MFMA roof on its
// padded work: 13 tiles on 8 waves, single-accumulator 16x16x4 chains, item prologue / epilogue.
__global__ __launch_bounds__(512, 4) void attn16_kernel(const float* __restrict__ qkv, int64_t ldqkv, float* __restrict__ out,
                                                        int64_t ldo, int L, int heads, int nitems, int64_t x3plane, int64_t prows) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nkt = (L + 15) >> 4;
  const int nd = nkt - 8;                                                 // waves with two tiles
  const bool dbl = wave < nd;
  const int ta = dbl ? 2 * wave : nd + wave;
  if (dbl) a16_run<true, 8>(qkv, ldqkv, out, ldo, L, heads, nitems, smem, ta, ta + 1, x3plane, prows);
  else a16_run<false, 8>(qkv, ldqkv, out, ldo, L, heads, nitems, smem, ta, ta, x3plane, prows);
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

static int attention_impl(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                          int32_t batch, int32_t L, int32_t heads, int32_t causal, void* stream, int x3) {
  if (!qkv || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_attention: null pointer%s");
  if (batch <= 0) return ACX_OK;
  if (L <= 0 || L > 224 || heads <= 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention: need 0 < L <= 224%s");
  if (ldqkv % 4 || ((uintptr_t)qkv & 15)) return acx_fail(ctx, ACX_E_BADARG, "acx_attention: qkv must be 16-byte aligned, ld%%4==0%s");
  const int nt = (L + 31) / 32;
  const bool nw4 = ACX_DBG_SWITCH("ATTN_NW4", false);
  const dim3 grid((unsigned)(batch * heads));
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, (hipStream_t)stream);
  if (!causal && L > 128 && L <= 256 && ldo % 4 == 0 && !((uintptr_t)out & 15) && ACX_DBG_SWITCH("ATTN16", true)) {
    // ViT sequence: 16-wide tiles (one or two query tiles per wave), chunked LDS-DMA staging, two workgroups per CU
    const int nitems = batch * heads;
    const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
    const int slots = 2 * ncu;                                            // two workgroups per CU
    hipLaunchKernelGGL(attn16_kernel, dim3((unsigned)(nitems < slots ? nitems : slots)), dim3(512), 2 * A16_STAGE_B, s, qkv,
                       ldqkv, out, ldo, L, heads, nitems, x3 ? (int64_t)batch * L * ldo : (int64_t)0,
                       x3 == 2 ? (int64_t)batch * L : (int64_t)0);
    ACX_CHECK_LAUNCH(ctx, "acx_attention");
    return ACX_OK;
  }
  if (x3) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention_x3: the ViT kernel only (non-causal, 128 < L <= 224)%s");
#define ACX_ATTN(NT, NW)                                                                           \
  do {                                                                                             \
    const size_t lds = (size_t)NT * 32 * (KROW + VROW) * 4 + NW * 32 * 4 + 16;                          \
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

extern "C" int acx_attention(acx_ctx* ctx, const float* qkv, int64_t ldqkv, float* out, int64_t ldo,
                             int32_t batch, int32_t L, int32_t heads, int32_t causal, void* stream) {
  return attention_impl(ctx, qkv, ldqkv, out, ldo, batch, L, heads, causal, stream, 0);
}

namespace {
// =====================================================================================================
// attn_p3_kernel -- the ViT's attention on the bf16 matrix cores at f32 accuracy (the ACX_PREC_F32X6 mode of the layer loop).
// q, k, v arrive as THREE bf16 planes each (hi | mid | lo: exact 24-bit split of the f32 values, written by the in-projection's
// epilogue in K-panel layout ACX_BF16X3P: plane = [3 W / 32][rows][32]); S = Q K^T and O = P V are six-product bf16 x 6
// products with f32 accumulation (acx_gemm_desc.pairs), the softmax runs in f32 in registers, P is split into three planes as
// it is consumed.  The f32 MFMA kernel above is bounded by the f32 pipe (1/16 of the bf16 rate): 61 GFLOP per layer at 157
// TFLOP/s = 0.39 ms at best (0.73-0.88 measured); here the same layer is 2 x 6 x 30.5 GFLOP of bf16 work = 0.15 ms of pipe time.
//   * one workgroup (8 waves, one per CU: 156 KB of LDS) per (frame, head), persistent; waves 0..6 own the seven 32-query
//     tiles of L = 197, wave 7 stages the operands: K and V of the head are copied by LDS-DMA from the panel-layout planes -- a
//     (plane, panel) block is 197 contiguous rows of 64 B, so the LDS image IS the global image (13 instructions of 1 KB per
//     block; K with the source-side bank swizzle chunk ^ ((row >> 2) & 3) of the GEMM units);
//   * phase A: S^T = K Q^T per 32-key tile (A operand = K rows from LDS by ds_read_b128, B operand = this wave's Q fragments,
//     loaded once per item from global): a lane then holds ONE query's scores (col = lane & 31), half of the keys each for the
//     two lane halves -- row max / sum are in-register plus one cross-half exchange;
//   * phase B: O^T = V^T P^T per 16-key step: the B operand is the lane's own probabilities, converted to bf16 planes in the
//     order the accumulator holds them (k-slot j of lane half hh <-> key 4 hh + (j & 3) + 8 (j >> 2) of the step: no lane
//     exchange), the A operand V^T by LDS transpose reads (ds_read_b64_tr_b16) in the SAME key order;
//   * K of item i + 1 is staged while item i is in phase B, V of item i + 1 while it is in phase A (two barriers per item).
// Output: three bf16 planes of O in K-panel layout (the out-projection's A operand), 8 bytes per lane and plane.
#ifndef AP3_NOFENCE
#define AP3_NOFENCE 0
#endif
#ifndef AP3_TRACE
#define AP3_TRACE 0
#endif
#ifndef AP3_ABL
#define AP3_ABL 0   // timing ablations (wrong results): 1 no exp, 2 no output stores, 4 no P V MFMAs, 8 no Q K^T MFMAs, 16 no staging, 32 no plane split of P, 64 no V fragment reads, 128 no K fragment reads
#endif
constexpr int AP3_ROWS = 208;                         // key rows staged per (plane, panel) block: 13 DMA instructions of 16 rows
constexpr int AP3_BLK_B = AP3_ROWS * 64;              // 13312
constexpr int AP3_LDS_B = 12 * AP3_BLK_B;             // K: 6 blocks (plane, panel), V: 6 blocks = 159744

typedef short ap3_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ap3_s16x4 ap3_lds_s16x4;
typedef __attribute__((address_space(3))) void ap3_lds_void;

template <int F16 = 0>
__device__ __forceinline__ void ap3_split8(const float (&p)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = acx_pk2<F16>(p[2 * e], p[2 * e + 1]);
    const float r0 = p[2 * e] - acx_unpk_lo<F16>(h[e]), r1 = p[2 * e + 1] - acx_unpk_hi<F16>(h[e]);
    m[e] = acx_pk2<F16>(r0, r1);
    l[e] = f2bf2(r0 - acx_unpk_lo<F16>(m[e]), r1 - acx_unpk_hi<F16>(m[e]));
  }
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ hv = {h[0], h[1], h[2], h[3]}, mv = {m[0], m[1], m[2], m[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(bf16x8, hv); mid = __builtin_bit_cast(bf16x8, mv); lo = __builtin_bit_cast(bf16x8, lv);
}

// NPROD = 3 (ACX_PREC_F32X3): the three leading products of both contractions only -- (hi, mid) (mid, hi) (hi, hi); the lo planes of
// K and V are neither staged nor read, P is split into two planes, the output's lo plane is not written (its consumer does not read it)
// F16 (with NPROD = 3: ACX_PREC_F16X3): the planes are TWO fp16 planes (hi | lo of the two-plane fp16 split), the products run on
// v_mfma_f32_32x32x16_f16, P and the output are split into fp16 pairs
template <int F16>
__device__ __forceinline__ f32x16 ap3_mfma(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (F16 != 0) {
    typedef _Float16 ap3_h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ap3_h8, a), __builtin_bit_cast(ap3_h8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
}
template <int NPROD, int F16 = 0>
__global__ __launch_bounds__(512, 2) void attn_p3_kernel(const u16* __restrict__ qkv3, int64_t plane_elems, int64_t rows_total,
                                                         u16* __restrict__ out3, int64_t out_plane_elems, int L, int heads, int nitems,
                                                         long long* __restrict__ trace) {
  constexpr int NPL = NPROD == 3 ? 2 : 3;               // planes of K / V / Q / the output in use
  // development timeline (-DAP3_TRACE=1 builds + ACX_TRACE_PTR): workgroup 0's waves stamp s_memrealtime (100 MHz) at the phase
  // boundaries of their first AP3_TRACE_ITEMS items: trace[(item_ordinal * 8 + wave) * 8 + point]
#if AP3_TRACE
#ifndef AP3_TRACE_FIRST
#define AP3_TRACE_FIRST 0
#endif
#define AP3_STAMP(ord, pt) do { if (trace && blockIdx.x == 0 && (ord) >= AP3_TRACE_FIRST && (ord) < AP3_TRACE_FIRST + 16 && lane == 0) trace[(((ord) - AP3_TRACE_FIRST) * 8 + wave) * 8 + (pt)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AP3_STAMP(ord, pt) do { } while (0)
#endif
  int ord = 0;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int W = heads * 64;
  const int np = W / 32;                                // panels of q (k: + np, v: + 2 np)
  const unsigned lds0 = (unsigned)(uintptr_t)(ap3_lds_void*)smem;
  // one LDS-DMA instruction: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to LDS [m0 .. + 1 KB)
#define AP3_GLDS(base, voff, ldsaddr)                                                              \
  do {                                                                                             \
    if ((AP3_ABL & 16) && nitems != 12345) break;                                                  \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(ldsaddr), "s"(base) : "memory"); \
  } while (0)
  // the stager (wave 7: the seven query tiles belong to waves 0..6): the 6 blocks (plane p, panel c) of K (which = 1) or V
  // (which = 2) of item `it` -> LDS blocks `blk0 + 2 p + c`.  Rolled loops: this code shares the kernel's register budget.
#define AP3_STAGE(it, which, blk0, SWZ)                                                            \
  do {                                                                                             \
    const int b_ = (it) / heads, h_ = (it) - b_ * heads;                                           \
    const unsigned vc_ = (unsigned)(((SWZ) ? ((lane & 3) ^ ((lane >> 4) & 3)) : (lane & 3)) * 16); \
    _Pragma("unroll 1") for (int pc = 0; pc < 2 * NPL; ++pc) {                                     \
      const u16* src_ = qkv3 + (int64_t)(pc >> 1) * plane_elems + ((int64_t)((which) * np + 2 * h_ + (pc & 1)) * rows_total + (int64_t)b_ * L) * 32; \
      _Pragma("unroll 1") for (int j = 0; j < 13; ++j) {                                           \
        const unsigned r_ = min(16u * (unsigned)j + (unsigned)(lane >> 2), (unsigned)(L - 1));   /* behind the sequence: a finite duplicate */ \
        AP3_GLDS(src_, r_ * 64u + vc_, lds0 + (unsigned)(((blk0) + pc) * AP3_BLK_B + j * 1024));   \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  // the same 78 instructions dealt over all eight waves (V: phase A is short -- 168 MFMAs per wave -- and one wave needs longer
  // than that just to ISSUE 78 LDS-DMA instructions; every wave waits for its own share before the barrier that ends the phase)
#define AP3_STAGE_SHARED(it, which, blk0, SWZ)                                                     \
  do {                                                                                             \
    const int b_ = (it) / heads, h_ = (it) - b_ * heads;                                           \
    const unsigned vc_ = (unsigned)(((SWZ) ? ((lane & 3) ^ ((lane >> 4) & 3)) : (lane & 3)) * 16); \
    _Pragma("unroll 1") for (int idx = wave; idx < 26 * NPL; idx += 8) {                           \
      const int pc = idx / 13, j = idx - 13 * pc;                                                  \
      const u16* src_ = qkv3 + (int64_t)(pc >> 1) * plane_elems + ((int64_t)((which) * np + 2 * h_ + (pc & 1)) * rows_total + (int64_t)b_ * L) * 32; \
      const unsigned r_ = min(16u * (unsigned)j + (unsigned)(lane >> 2), (unsigned)(L - 1));       \
      AP3_GLDS(src_, r_ * 64u + vc_, lds0 + (unsigned)(((blk0) + pc) * AP3_BLK_B + j * 1024));     \
    }                                                                                              \
  } while (0)

  int item = blockIdx.x;
  if (item >= nitems) return;
  const float sc = 0.125f * 1.44269504088896340736f;    // 1 / sqrt(64) and log2(e): p = exp2((s - max) sc)
  if (wave == 7) {
    // ================================================================ the stager
    AP3_STAGE(item, 1, 0, 1);                           // K of the first item
    for (; item < nitems; item += (int)gridDim.x) {
      AP3_STAMP(ord, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K(item) landed
      AP3_STAMP(ord, 1);
      __builtin_amdgcn_s_barrier();                     // B1: ... and everybody is done with V(previous item)
      AP3_STAMP(ord, 2);
      AP3_STAGE_SHARED(item, 2, 6, 0);                  // this wave's share of V(item): needed after phase A
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      AP3_STAMP(ord, 3);
      __builtin_amdgcn_s_barrier();                     // B2: V(item) landed, everybody is done with K(item)
      AP3_STAMP(ord, 4);
      const int nxt = item + (int)gridDim.x;
      if (nxt < nitems) AP3_STAGE(nxt, 1, 0, 1);        // K(next item) during this item's phase B
      AP3_STAMP(ord, 5);
      ++ord;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no DMA may still be writing this workgroup's LDS at exit
    return;
  }
  // ================================================================== the seven query tiles
  // V^T fragment addresses (ds_read_b64_tr_b16: every 16 lanes read a [4 keys][16 d] block): lane i of the group supplies
  // key (4 hh + 8 r + (i >> 2)) of the 16-key step, d = 16 ((lane >> 4) & 1) + 4 (i & 3) of the 32-d panel
  const int i16 = lane & 15;
  // every LDS address below = one of six per-lane base registers + an immediate < 64 KB (LDS offsets beyond the 16-bit
  // immediate would each cost a hoisted address register: the blocks of the lo planes get bases of their own)
  int va = 6 * AP3_BLK_B + (4 * hh + (i16 >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;     // V hi / mid planes
  int va2 = va + 4 * AP3_BLK_B;                                                                         // V lo plane
  const int swk = (li >> 2) & 3;
  int ka0 = li * 64 + ((0 + hh) ^ swk) * 16, ka1 = li * 64 + ((2 + hh) ^ swk) * 16;                     // K hi / mid planes, d step parity
  int ka0l = ka0 + 4 * AP3_BLK_B, ka1l = ka1 + 4 * AP3_BLK_B;                                           // K lo plane
  asm volatile("" : "+v"(va), "+v"(va2), "+v"(ka0), "+v"(ka1), "+v"(ka0l), "+v"(ka1l));
  const int qrow = min(32 * wave + li, L - 1);
  for (; item < nitems; item += (int)gridDim.x) {
    const int b = item / heads, h = item - b * heads;
    // ---------------------------------------------------------------- phase A: S^T = K Q^T
    bf16x8 qf[3][4];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[p][ks] = *reinterpret_cast<const bf16x8*>(qkv3 + (int64_t)p * plane_elems +
                                                     ((int64_t)(2 * h + (ks >> 1)) * rows_total + (int64_t)b * L + qrow) * 32 + ((ks & 1) * 2 + hh) * 8);
    AP3_STAMP(ord, 0);
    __builtin_amdgcn_s_barrier();                       // B1
    AP3_STAMP(ord, 1);
    AP3_STAGE_SHARED(item, 2, 6, 0);                    // this wave's share of V(item)
    AP3_STAMP(ord, 2);
    f32x16 sacc[7];
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cst = (ks >> 1) * AP3_BLK_B + 32 * kt * 64;        // panel of the d step, key tile (swizzle: row bits 2..3 = li's)
        const bool nok_ = (AP3_ABL & 128) && nitems != 12345;                        // (timing ablation: no K fragment reads)
        const bf16x8 kh = nok_ ? qf[0][ks] : *reinterpret_cast<const bf16x8*>(smem + ((ks & 1) ? ka1 : ka0) + cst);
        const bf16x8 km = nok_ ? qf[1][ks] : *reinterpret_cast<const bf16x8*>(smem + ((ks & 1) ? ka1 : ka0) + 2 * AP3_BLK_B + cst);
        // smallest cross terms first: (hi,lo) (mid,mid) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
        if constexpr (NPROD == 6) {
        const bf16x8 kl = nok_ ? qf[2][ks] : *reinterpret_cast<const bf16x8*>(smem + ((ks & 1) ? ka1l : ka0l) + cst);
        if ((AP3_ABL & 8) && nitems != 12345) { sacc[kt][0] += (float)kh[0] + (float)km[0] + (float)kl[0]; continue; }
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[2][ks], sacc[kt], 0, 0, 0);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[1][ks], sacc[kt], 0, 0, 0);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][ks], sacc[kt], 0, 0, 0);
        }
        sacc[kt] = ap3_mfma<F16>(kh, qf[1][ks], sacc[kt]);
        sacc[kt] = ap3_mfma<F16>(km, qf[0][ks], sacc[kt]);
        sacc[kt] = ap3_mfma<F16>(kh, qf[0][ks], sacc[kt]);
      }
#if !AP3_NOFENCE
      __builtin_amdgcn_sched_barrier(0);                // keep the tiles' fragment loads from piling up (register budget)
#endif
    }
    AP3_STAMP(ord, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's share of V(item) landed
    __builtin_amdgcn_s_barrier();                       // B2
    AP3_STAMP(ord, 4);
    // ---------------------------------------------------------------- phase B: softmax, O^T = V^T P^T
    // (measured in round 6: the softmax moved IN FRONT of this barrier -- it needs neither K nor V, and the two waves of a SIMD leave
    // phase A 2-3 us apart, the older one winning the matrix pipe -- shortens the traced critical path by ~2 us per item and changes
    // the kernel's time by nothing: profiles/r06_attention_notes.txt)
    // lane (query li, half hh) holds the scores of keys 32 kt + (e & 3) + 8 (e >> 2) + 4 hh
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (kt == 6) {                                  // 192 < L <= 208: only the last tile is cut
          const int key = 192 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          sacc[kt][e] = key < L ? sacc[kt][e] : -INFINITY;
        }
        mx = fmaxf(mx, sacc[kt][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * sc;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (!(AP3_ABL & 1) || nitems == 12345) sacc[kt][e] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], sc, -mxs));      // v_exp_f32: 2^x, exp2(-inf) = 0
        sum += sacc[kt][e];
      }
    sum += __shfl_xor(sum, 32, 64);
    AP3_STAMP(ord, 5);                                  // softmax done
    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
#pragma unroll
    for (int st = 0; st < 13; ++st) {                   // 16-key steps; keys >= 208 are all masked
      const int kt = st >> 1, u = st & 1;
      const float pv[8] = {sacc[kt][8 * u], sacc[kt][8 * u + 1], sacc[kt][8 * u + 2], sacc[kt][8 * u + 3],
                           sacc[kt][8 * u + 4], sacc[kt][8 * u + 5], sacc[kt][8 * u + 6], sacc[kt][8 * u + 7]};
      bf16x8 ph, pm, pl;
      if ((AP3_ABL & 32) && nitems != 12345) { ph = pm = pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&pv[0])); } else
      ap3_split8<F16>(pv, ph, pm, pl);                  // (NPROD == 3: pl is dead code)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x8 vf[3];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          if ((AP3_ABL & 64) && nitems != 12345) { vf[p] = qf[p][dt]; continue; }     // (timing ablation: no V fragment reads)
          const char* vb = smem + (p == 2 ? va2 : va) + ((p == 2 ? 0 : 2 * p) + dt) * AP3_BLK_B + (16 * st) * 64;
          const ap3_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ap3_lds_s16x4*)(vb));
          const ap3_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ap3_lds_s16x4*)(vb + 8 * 64));
          typedef short s16x8_ __attribute__((ext_vector_type(8)));
          const s16x8_ v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          vf[p] = __builtin_bit_cast(bf16x8, v8);
        }
        if constexpr (NPROD == 6) {
        if ((AP3_ABL & 4) && nitems != 12345) { oacc[dt][0] += (float)vf[0][0] + (float)vf[1][0] + (float)vf[2][0] + (float)ph[0] + (float)pm[0] + (float)pl[0]; continue; }
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], ph, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pm, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pl, oacc[dt], 0, 0, 0);
        }
        oacc[dt] = ap3_mfma<F16>(vf[1], ph, oacc[dt]);
        oacc[dt] = ap3_mfma<F16>(vf[0], pm, oacc[dt]);
        oacc[dt] = ap3_mfma<F16>(vf[0], ph, oacc[dt]);
      }
#if !AP3_NOFENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    AP3_STAMP(ord, 6);                                  // P V done
    // normalise, split into planes, store: lane (query li, hh) holds d = 32 dt + (e & 3) + 8 (e >> 2) + 4 hh, i.e. for every
    // group g4 of four registers an 8-byte piece of its row; v_permlane32_swap pairs the pieces of the two lane halves -- half
    // 0 ends up with d = 16 gp .. + 7, half 1 with d = 16 gp + 8 .. + 15 of the row: 16-byte stores, 12 per wave and item
    const float inv = 1.f / sum;
    const int q = 32 * wave + li;
    const bool st_ok = q < L && (!(AP3_ABL & 2) || nitems == 12345);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        uint2 pl_[2][3];                                  // [g4 = 2 gp + {0, 1}][plane]
#pragma unroll
        for (int z = 0; z < 2; ++z) {
          const int g4 = 2 * gp + z;
          const float ov[4] = {oacc[dt][4 * g4] * inv, oacc[dt][4 * g4 + 1] * inv, oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv};
          uint2 ph2, pm2, pl2;
          ph2.x = acx_pk2<F16>(ov[0], ov[1]); ph2.y = acx_pk2<F16>(ov[2], ov[3]);
          const float r0 = ov[0] - acx_unpk_lo<F16>(ph2.x), r1 = ov[1] - acx_unpk_hi<F16>(ph2.x);
          const float r2 = ov[2] - acx_unpk_lo<F16>(ph2.y), r3 = ov[3] - acx_unpk_hi<F16>(ph2.y);
          pm2.x = acx_pk2<F16>(r0, r1); pm2.y = acx_pk2<F16>(r2, r3);
          pl2.x = f2bf2(r0 - __uint_as_float(pm2.x << 16), r1 - __uint_as_float(pm2.x & 0xffff0000u));
          pl2.y = f2bf2(r2 - __uint_as_float(pm2.y << 16), r3 - __uint_as_float(pm2.y & 0xffff0000u));
          pl_[z][0] = ph2; pl_[z][1] = pm2; pl_[z][2] = pl2;
        }
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          // swap(X = piece of g4 = 2 gp, Y = piece of g4 = 2 gp + 1): half 0 gets (own X, partner's X), half 1 (partner's Y, own Y)
          typedef unsigned ap3_u2 __attribute__((ext_vector_type(2)));
          const ap3_u2 w0 = __builtin_amdgcn_permlane32_swap(pl_[0][p].x, pl_[1][p].x, false, false);
          const ap3_u2 w1 = __builtin_amdgcn_permlane32_swap(pl_[0][p].y, pl_[1][p].y, false, false);
          // half 0: w0[0], w1[0] = own X (d 0..3), w0[1], w1[1] = partner's X (d 4..7); half 1: [0] = partner's Y (d 0..3), [1] = own Y
          const uint4 v16 = make_uint4(w0[0], w1[0], w0[1], w1[1]);
          if (st_ok) {
            u16* dst = out3 + (int64_t)p * out_plane_elems + ((int64_t)(2 * h + dt) * rows_total + (int64_t)b * L + q) * 32 + 8 * (2 * gp + hh);
            *reinterpret_cast<uint4*>(dst) = v16;
          }
        }
      }
    AP3_STAMP(ord, 7);
    ++ord;
  }
#undef AP3_STAMP
#undef AP3_STAGE_SHARED
#undef AP3_STAGE
#undef AP3_GLDS
}
}  // namespace

// q | k | v as three bf16 planes in K-panel layout (ACX_BF16X3P of the [batch * L, 3 heads * 64] in-projection output) ->
// attention output as three bf16 planes in K-panel layout ([heads * 64 / 32][batch * L][32]); 128 < L <= 208, non-causal
extern "C" int acx_attention_p3(acx_ctx* ctx, const void* qkv_planes, void* out_planes, int32_t batch, int32_t L, int32_t heads,
                                void* stream) {
  return acx_attention_p3n(ctx, qkv_planes, out_planes, batch, L, heads, 6, stream);
}
// products = 6: the f32-accurate form; 3: the three leading products of both contractions (ACX_PREC_F32X3: the lo planes of the
// operands are not read, the output's lo plane is not written)
extern "C" int acx_attention_p3n(acx_ctx* ctx, const void* qkv_planes, void* out_planes, int32_t batch, int32_t L, int32_t heads,
                                 int32_t products, void* stream) {
  if (!qkv_planes || !out_planes) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_p3: null pointer%s");
  const bool f16 = products == 103;                     // 103: three products on TWO fp16 planes (ACX_PREC_F16X3)
  if (f16) products = 3;
  if (products != 6 && products != 3) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_p3n: products must be 6, 3 or 103 (3 on fp16 planes)%s");
  if (batch <= 0) return ACX_OK;
  if (L <= 192 || L > AP3_ROWS || heads <= 0 || (((uintptr_t)qkv_planes | (uintptr_t)out_planes) & 15))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention_p3: 192 < L <= 208, 16-byte aligned planes%s");
  const int64_t rows = (int64_t)batch * L;
  const int nitems = batch * heads;
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, s);
  const int dev_slot = (ctx ? ctx->device : 0) & 63;
  static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_p3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP3_LDS_B);
    (void)hipFuncSetAttribute((const void*)attn_p3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP3_LDS_B);
    (void)hipFuncSetAttribute((const void*)attn_p3_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AP3_LDS_B);
    attr_done = true;
  }
#if AP3_TRACE
  long long* trace_ = getenv("ACX_TRACE_PTR") ? (long long*)strtoull(getenv("ACX_TRACE_PTR"), nullptr, 0) : nullptr;
#else
  long long* trace_ = nullptr;
#endif
  if (f16)
    hipLaunchKernelGGL((attn_p3_kernel<3, 1>), dim3((unsigned)(nitems < ncu ? nitems : ncu)), dim3(512), (size_t)AP3_LDS_B, s, (const u16*)qkv_planes,
                       rows * 3 * heads * 64, rows, (u16*)out_planes, rows * heads * 64, L, heads, nitems, trace_);
  else if (products == 3)
    hipLaunchKernelGGL(attn_p3_kernel<3>, dim3((unsigned)(nitems < ncu ? nitems : ncu)), dim3(512), (size_t)AP3_LDS_B, s, (const u16*)qkv_planes,
                       rows * 3 * heads * 64, rows, (u16*)out_planes, rows * heads * 64, L, heads, nitems, trace_);
  else
  hipLaunchKernelGGL(attn_p3_kernel<6>, dim3((unsigned)(nitems < ncu ? nitems : ncu)), dim3(512), (size_t)AP3_LDS_B, s, (const u16*)qkv_planes,
                     rows * 3 * heads * 64, rows, (u16*)out_planes, rows * heads * 64, L, heads, nitems, trace_);
  ACX_CHECK_LAUNCH(ctx, "acx_attention_p3");
  return ACX_OK;
}

extern "C" int acx_attention_x3(acx_ctx* ctx, const float* qkv, int64_t ldqkv, void* out_planes, int64_t ldo,
                                int32_t batch, int32_t L, int32_t heads, void* stream) {
  return attention_impl(ctx, qkv, ldqkv, (float*)out_planes, ldo, batch, L, heads, 0, stream, 1);
}
extern "C" int acx_attention_x3_panel(acx_ctx* ctx, const float* qkv, int64_t ldqkv, void* out_planes, int64_t ldo,
                                      int32_t batch, int32_t L, int32_t heads, void* stream) {
  if (ldo != (int64_t)heads * 64) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_x3_panel: dense planes (ldo == heads * 64)%s");
  return attention_impl(ctx, qkv, ldqkv, (float*)out_planes, ldo, batch, L, heads, 0, stream, 2);
}

namespace {
// =====================================================================================================
// acx_attention_bf16 -- the same attention with bf16 operands (bf16 mode of the ViT, not a parity path):
// qkv and the output are bf16 in global memory, QK^T and PV run on v_mfma_f32_32x32x16_bf16, softmax in f32.
//   * K of the head in LDS as bf16 rows of 128 B padded to 144 B (conflict-free ds_read_b128 A fragments);
//     V as two 32-column panels of 64-byte rows (16-byte LDS writes, like K); the bf16 MFMA packs 8 contraction indices
//     (keys) per lane, so the V^T fragments come out of LDS by transpose reads (ds_read_b64_tr_b16: every 16 lanes read a
//     [4 keys][16 d] block) -- round 4 transposed while staging with 2-byte LDS writes: eight ds_write_b16 per 16 bytes and
//     the kernel's only LDS bank conflicts (6.8 M cycles per launch).
//     K + V = 61 KB -> two workgroups per CU (the f32 kernel needs 112 KB and runs one).
//   * S^T = K Q^T exactly as in the f32 kernel (lane = query column, registers = keys), so softmax is
//     in-register + one cross-half exchange, and register group 8c..8c+7 of a key tile, packed to bf16, IS the
//     B operand of O^T = V^T P^T for key chunk c.  The keys a lane half holds there are
//     {16c + 4h + (0..3)} u {16c + 8 + 4h + (0..3)}: the V^T fragment is two transpose reads, eight rows apart.
//   * O^T's C layout has lane = query again: 1/rowsum is a per-lane scalar and a lane stores 4 x 8 bytes per
//     32-wide slice of its row.
constexpr int BK_ROWB = 144;       // bytes per K row in LDS
constexpr int BV_PANEL_B = 224 * 64 + 64;   // one 32-column panel of V: 224 rows of 64 B (+ 64: the two panels' rows on different banks)
typedef short ab_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ab_s16x4 ab_lds_s16x4;

__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) { return f2bf2(a, b); }
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const u16* __restrict__ qkv, int64_t ldqkv, u16* __restrict__ out,
                                                          int64_t ldo, int L, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                                  // [NT*32][144 B]
  char* sV = smem + NT * 32 * BK_ROWB;              // [2 panels][224 rows][64 B]
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int W = heads * 64;
  const u16* base = qkv + (int64_t)b * L * ldqkv + h * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, hh = lane >> 5;

  // ---- stage K (row copy) and V (transposed); 8 lanes cover one 128-byte row; rows >= L are zero
  constexpr int NP = NT * 32 * 8;                   // 16-byte pieces per operand
  constexpr int NSTG = (NP + 255) / 256;
  uint4 stk[NSTG], stv[NSTG];
#pragma unroll
  for (int j = 0; j < NSTG; ++j) {
    const int i = t + j * 256;
    const int row = i >> 3, pc = i & 7;
    stk[j] = make_uint4(0, 0, 0, 0);
    stv[j] = stk[j];
    if (i < NP && row < L) {
      const u16* p = base + (int64_t)row * ldqkv + 8 * pc;
      stk[j] = *reinterpret_cast<const uint4*>(p + W);
      stv[j] = *reinterpret_cast<const uint4*>(p + 2 * W);
    }
  }
  // ---- Q fragments (B operand) of this wave's (at most two: NT <= 8) query blocks, requested BEHIND the K / V loads so that the
  // whole workgroup waits for memory once: query row qb*32 + li, d = 8*(2kk + hh) .. +7.  (Loaded at the head of each query
  // block they cost a second and a third exposed round trip per workgroup: 16.7 us of lifetime for ~3.5 us of arithmetic.)
  uint4 qpre[2][4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = min((wave + 4 * it) * 32 + li, L - 1);
    const u16* qp = base + (int64_t)q * ldqkv + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qpre[it][kk] = *reinterpret_cast<const uint4*>(qp + 16 * kk);
  }
#pragma unroll
  for (int j = 0; j < NSTG; ++j) {
    const int i = t + j * 256;
    const int row = i >> 3, pc = i & 7;
    if (i < NP) {
      *reinterpret_cast<uint4*>(sK + row * BK_ROWB + pc * 16) = stk[j];
      *reinterpret_cast<uint4*>(sV + (pc >> 2) * BV_PANEL_B + row * 64 + (pc & 3) * 16) = stv[j];
    }
  }
  __syncthreads();

  const int nqb = (L + 31) / 32;
  const int vtr = (4 * hh + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // transpose-read offset inside a 16-key step
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int qb = wave + 4 * it;
    if (qb >= nqb) break;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(&qpre[it][kk]);
    // ---- S^T tiles: keys x queries
    f32x16 st[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kt][e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kt * 32 + li) * BK_ROWB + (2 * kk + hh) * 16);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);     // one key tile of fragments in flight, not all 28 (register spills)
    }
    // ---- softmax over the keys of this lane's query: register r of tile kt is key kt*32 + (r&3) + 8(r>>2) + 4hh.
    // At the bf16 MFMA rate this VALU work weighs more than the MFMAs (SQ_ACTIVE_INST_VALU 35 % vs MFMA busy 18 %): the
    // maximum is taken over the RAW scores (the scale is positive), only the last key tile holds keys >= L, and scale,
    // shift and the base change are ONE (packed) fma in front of v_exp_f32: p = 2^((s - m) * 0.125 * log2 e).
    {
      constexpr int kt = NT - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        st[kt][r] = key < L ? st[kt][r] : -3.0e38f;
      }
    }
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(st[kt][r], st[kt][r + 1]));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float sc_ = 0.125f * 1.4426950408889634f, nmc = -m * sc_;
    f32x2v sum2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2v s2 = {st[kt][r], st[kt][r + 1]};
        const f32x2v a2 = __builtin_elementwise_fma(s2, (f32x2v){sc_, sc_}, (f32x2v){nmc, nmc});
        const f32x2v p2 = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
        st[kt][r] = p2[0]; st[kt][r + 1] = p2[1];
        sum2 += p2;
      }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // ---- O^T = V^T P^T : two 32-wide e tiles
    f32x16 o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 pk;
        pk.x = pack2_bf16(st[kt][8 * c + 0], st[kt][8 * c + 1]);
        pk.y = pack2_bf16(st[kt][8 * c + 2], st[kt][8 * c + 3]);
        pk.z = pack2_bf16(st[kt][8 * c + 4], st[kt][8 * c + 5]);
        pk.w = pack2_bf16(st[kt][8 * c + 6], st[kt][8 * c + 7]);
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(&pk);
        // lane i of a 16-lane group supplies key 4 hh + (i >> 2) (+ 8: second read) of the 16-key step, d = 16 ((lane >> 4) & 1) + 4 (i & 3)
        const char* v0 = sV + vtr + (kt * 32 + 16 * c) * 64;
        typedef short s16x8_ __attribute__((ext_vector_type(8)));
        bf16x8 vf[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const ab_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab_lds_s16x4*)(v0 + dt * BV_PANEL_B));
          const ab_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ab_lds_s16x4*)(v0 + dt * BV_PANEL_B + 8 * 64));
          const s16x8_ v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          vf[dt] = __builtin_bit_cast(bf16x8, v8);
        }
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf, o1, 0, 0, 0);
      }
    // ---- store: lane = query, register r of tile et is column e = et*32 + (r&3) + 8(r>>2) + 4hh
    const int qrow = qb * 32 + li;
    if (qrow < L) {
      u16* op = out + ((int64_t)b * L + qrow) * ldo + h * 64 + 4 * hh;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 w0, w1;
        w0.x = pack2_bf16(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv); w0.y = pack2_bf16(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
        w1.x = pack2_bf16(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv); w1.y = pack2_bf16(o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
        *reinterpret_cast<uint2*>(op + 8 * g4) = w0;
        *reinterpret_cast<uint2*>(op + 32 + 8 * g4) = w1;
      }
    }
  }
}

}  // namespace

extern "C" int acx_attention_bf16(acx_ctx* ctx, const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int32_t batch,
                                  int32_t L, int32_t heads, void* stream) {
  if (!qkv || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_attention_bf16: null pointer%s");
  if (batch <= 0) return ACX_OK;
  if (L <= 0 || L > 224 || heads <= 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_attention_bf16: need 0 < L <= 224%s");
  if (ldqkv % 8 || ldo % 4 || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7))
    return acx_fail(ctx, ACX_E_BADARG, "acx_attention_bf16: qkv 16-byte aligned with ld%%8==0, out 8-byte aligned with ld%%4==0%s");
  const int nt = (L + 31) / 32;
  const dim3 grid((unsigned)(batch * heads));
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, s);
#define ACX_ATTNB(NT)                                                                              \
  do {                                                                                             \
    const size_t lds = (size_t)NT * 32 * BK_ROWB + 2 * BV_PANEL_B;                                 \
    static bool done = false;                                                                      \
    if (!done) {                                                                                   \
      (void)hipFuncSetAttribute((const void*)attn_bf16_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      done = true;                                                                                 \
    }                                                                                              \
    hipLaunchKernelGGL((attn_bf16_kernel<NT>), grid, dim3(256), lds, s, (const u16*)qkv, ldqkv, (u16*)out, ldo, L, heads); \
  } while (0)
  switch (nt) {
    case 1: ACX_ATTNB(1); break;
    case 2: ACX_ATTNB(2); break;
    case 3: ACX_ATTNB(3); break;
    case 4: ACX_ATTNB(4); break;
    case 5: ACX_ATTNB(5); break;
    case 6: ACX_ATTNB(6); break;
    default: ACX_ATTNB(7); break;
  }
#undef ACX_ATTNB
  ACX_CHECK_LAUNCH(ctx, "acx_attention_bf16");
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
