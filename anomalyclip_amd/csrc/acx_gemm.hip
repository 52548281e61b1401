// acx_gemm -- the MFMA workhorse:  C[M,N] = epilogue( amap(A)[M,K] . W[N,K]^T )
//
// Replaces every dense contraction on the AnomalyCLIP hot path (see include/acx.h for the
// reference call sites).  Both operands are K-contiguous (PyTorch nn.Linear stores W as [N,K]),
// which is exactly the MFMA A/B fragment orientation: a lane needs consecutive k of ONE row.
//
// CDNA4 design
//   * block tile 128x128, 4 wavefronts (2x2), each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32
//     accumulators (64 acc VGPRs/lane).  2 blocks per CU (LDS 2 x 72 KiB, <=128 VGPR).
//   * K-step = 128 BYTES of K per row for both precisions (32 f32 / 64 bf16), staged through LDS
//     rows of 128 B + 16 B pad (stride 144 B = 9 x 16-B slots -> the 16-lane groups of
//     ds_read_b128 and the 8-lane groups of ds_write_b128 are bank-conflict free).
//   * ACX_PREC_F32 : v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 157 TF roof).  One
//     ds_read_b128 feeds FOUR MFMAs per operand by permuting K inside the step: lane half h of
//     MFMA (q,j) supplies k = 4*(2q+h)+j for A and B alike (a dot product does not care about
//     the order of k as long as both operands agree).
//   * ACX_PREC_BF16: v_mfma_f32_32x32x16_bf16, one ds_read_b128 (8 bf16) per operand per MFMA.
//   * register-staged double buffering (guide T14) as a hand-scheduled 4-phase K-step: fragments are
//     double-buffered in registers, the next tile's LDS writes ride in the shadow of MFMA phases 1-2,
//     the ONE barrier of the step sits between phases 2 and 3 (every wave already holds its phase-3
//     fragments), and the global loads two K-steps ahead plus the next tile's phase-0 fragment reads
//     are issued right after it, under phase 3's MFMAs.
//   * staging values live in NAMED registers and the epilogue computes every value before the first
//     store is issued (both are measured necessities with hipcc: arrays get demoted to scratch, and
//     a load pending across a predicated store makes every store wait for the previous one).
//   * skinny problems (few output tiles, long K) are split along K over gridDim.y with a fused
//     reduce+epilogue kernel (fixed summation order).
//   * A-operand prologue fused into the staging pass: f32->bf16 conversion, re-centring
//     (a - ncentroid[k]), the 3x3-conv row gather (implicit GEMM over the (gn,gl) token grid,
//     zero padding) and the reference's test-mode tiling gather.
//   * epilogue fused: bias, QuickGELU / LeakyReLU, axial positional embedding, residual add,
//     f32 or bf16 store.
//   * XCD-aware 1-D grid: block b runs on XCD b%8; the bijective remap gives every XCD a
//     contiguous run of tiles (n fastest) so the A tile of a row of tiles stays in that XCD's L2.
//
// Kernel families (all dispatched from acx_gemm / acx_gemm_tn at the bottom of this file):
//   gemm_kernel<...>         this file   4 waves per tile: every precision / row map / epilogue, split-K
//   gemm_f32_w8_kernel       _w8.h       8 waves per tile: branch-free f32 problems and the 3x3 convolutions
//   gemm_bf16_dma_kernel     _bf16.h     bf16 operands, global->LDS DMA staging, 128x128, two blocks per CU
//   gemm_bf16_ring_kernel    _bf16.h     bf16, persistent 256x256 tiles for the large-M shapes
//   gemm_tn_kernel / _w8     _tn.h       weight gradients C = A^T B (reduction over rows), split-M + reduce
#include "acx_internal.h"

#include <stdlib.h>
#include <type_traits>

#ifndef ACX_TRACE
#define ACX_TRACE 0
#endif

namespace {

// QuickGELU of every epilogue: v * sigmoid(1.702 v) with v_exp_f32 + v_rcp_f32 (1 ulp).  "1.f / x" would be the IEEE division
// sequence (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup): ten VALU instructions per element of a
// 256 x 256 tile, with the MFMA pipe idle behind them.
// (the empty asm pins the product as an f32 value HERE -- a plane epilogue that splits the result into hi | mid | lo would otherwise get
// the multiply contracted into its first residual, fma(v, s, -hi), and its planes would hold a value that is not the f32 result)
__device__ __forceinline__ float acx_quickgelu(float v) {
  float r = v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
  asm volatile("" : "+v"(r));
  return r;
}

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;                // LDS row stride in bytes (128 data + 16 pad)
constexpr int TILE_B = BM * ROWB;        // 18432
constexpr int NTHREADS = 256;
constexpr int ACX_SK_MAX_M = 320;        // acx_gemm sends f32 problems with at most this many rows to the few-row kernel

struct Args {
  acx_gemm_desc d;
  int tiles_n;
  long long* trace;        // debug timeline buffer (ACX_TRACE builds only)
  int ksplit, kchunk;      // split-K (FAST path, skinny problems): gridDim.y splits of kchunk K-steps each
  float* partial;          // [ksplit][M][N] raw partial sums (epilogue applied by splitk_reduce_kernel)
  const float* zeros;      // acx_gemm_desc.zero_page (conv taps outside the grid on the LDS-DMA kernels)
  unsigned int* counters;  // acx_gemm_desc.counters (few-row kernel's cross-workgroup K split: one arrival counter per tile)
  int tile0, ntiles;       // gemm_x6_p4_kernel: the launch covers the full 256 x 256 tiles tile0 .. tile0 + ntiles (ntiles = 0: all)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ uint4 pack_bf16x8(float4 a, float4 b) {
  uint4 r;
  r.x = f2bf2(a.x, a.y);
  r.y = f2bf2(a.z, a.w);
  r.z = f2bf2(b.x, b.y);
  r.w = f2bf2(b.z, b.w);
  return r;
}

// PREC: 0 f32 MFMA, 1 bf16 MFMA.  A_BF16: A stored as bf16 in global (PREC 1 only).
// FAST: identity row map, no a_sub / positional epilogue, K % KE == 0 -> branch-free staging
//       (out-of-range rows are CLAMPED to a valid row and masked at the store) and a compile-time
//       epilogue (ACT activation, RES residual).  FAST == 0 is the generic runtime-flag path.
template <int PREC, int A_BF16, int C_BF16, int FAST, int ACT, int RES>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const acx_gemm_desc& d = g.d;
  constexpr int KE = PREC == 0 ? 32 : 64;     // K elements per step
  constexpr int CE = PREC == 0 ? 4 : 8;       // elements per 16-B LDS chunk
  constexpr int NLA = (PREC == 1 && !A_BF16) ? 2 : 1;   // 16-B global loads per staged A chunk
  constexpr int WB = PREC == 0 ? 4 : 2;       // bytes per W element

  // ---- XCD-aware tile assignment (bijective remap, guide T1)
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tiles_n, tn = wg % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int chunk = t & 7, rbase = t >> 3;
#if ACX_TRACE
  const long long tr0 = wall_clock64();
  long long tr1 = 0, tr2 = 0;
#endif

  // ---- per-thread source rows for the 4 staged A rows and 4 staged W rows
  const int grid_sz = FAST ? 1 : d.gn * d.gl;
  long a_row[4];        // identity/testtile: source row (clamped); conv: base row of the tile
  int a_n[4], a_l[4];   // conv: grid coordinates
  bool a_ok[4], w_ok[4];
  const char* w_ptr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int m = m0 + rbase + 32 * r;
    a_ok[r] = m < d.M;
    m = a_ok[r] ? m : d.M - 1;
    a_row[r] = m;
    a_n[r] = a_l[r] = 0;
    if constexpr (!FAST) {
      if (d.amap == ACX_AMAP_TESTTILE) {
        const int per = grid_sz * d.seg;
        const int b = m / per, rem = m - b * per;
        const int sg = rem / grid_sz, rem2 = rem - sg * grid_sz;
        const int n = rem2 / d.gl, l = rem2 - n * d.gl;
        a_row[r] = (((long)b * d.gn + n) * d.seg + sg) * d.gl + l;
      } else if (d.amap == ACX_AMAP_TILETABLE) {
        // row (tile, n, l) reads source row table[2 tile] + n * table[2 tile + 1] + l: tiles of several videos, each with its
        // own segment size S (base = video row0 + s L, stride = S L), in ONE launch
        const int tile = m / grid_sz, rem = m - tile * grid_sz;
        const int n = rem / d.gl, l = rem - n * d.gl;
        a_row[r] = (long)d.tile_table[2 * tile] + (long)n * d.tile_table[2 * tile + 1] + l;
      } else if (d.amap == ACX_AMAP_CONV3X3) {
        const int tile = m / grid_sz, rem = m - tile * grid_sz;
        a_n[r] = rem / d.gl;
        a_l[r] = rem - a_n[r] * d.gl;
        a_row[r] = (long)tile * grid_sz;
      }
    }
    int n = n0 + rbase + 32 * r;
    w_ok[r] = n < d.N;
    n = w_ok[r] ? n : d.N - 1;
    w_ptr[r] = (const char*)d.W + ((size_t)n * d.ldw + chunk * CE) * WB;
  }

  // ---- staging registers: NAMED variables, not arrays (an array here is demoted to scratch memory by
  // hipcc's alloca handling as soon as anything indexes it from a lambda -- guide rule 20)
  float4 ra0, ra1, ra2, ra3;      // A chunk (f32: 4 floats; bf16-in-global: 8 bf16 bit-cast)
  float4 rb0, rb1, rb2, rb3;      // second half of the A chunk when f32 A is converted to bf16 (NLA == 2)
  uint4 rw0, rw1, rw2, rw3;       // W chunk
  rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);

  // FAST-path staging loads as a macro (straight-line code keeps `ra`/`rw` in VGPRs; the lambda form
  // made the register allocator demote them to scratch)
#define ACX_FAST_LOAD_ROW(r, k0)                                                                   \
  do {                                                                                             \
    const int kcol_ = (k0) + chunk * CE;                                                           \
    if constexpr (A_BF16) {                                                                        \
      const uint4 v_ = *reinterpret_cast<const uint4*>((const u16*)d.A + (size_t)a_row[r] * d.lda + kcol_); \
      ra##r = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); \
    } else {                                                                                       \
      const float* p_ = (const float*)d.A + (size_t)a_row[r] * d.lda + kcol_;                      \
      ra##r = ld4(p_);                                                                             \
      if constexpr (NLA == 2) rb##r = ld4(p_ + 4);                                                 \
    }                                                                                              \
    rw##r = *reinterpret_cast<const uint4*>(w_ptr[r] + (size_t)(k0) * WB);                         \
  } while (0)
#define ACX_FAST_LOAD(k0) \
  do { ACX_FAST_LOAD_ROW(0, k0); ACX_FAST_LOAD_ROW(1, k0); ACX_FAST_LOAD_ROW(2, k0); ACX_FAST_LOAD_ROW(3, k0); } while (0)

  // generic-path staging (runtime row map / a_sub / bounds): per-step uniforms then one macro per row
  int g_dn = 0, g_dl = 0, g_kc = 0, g_kw = 0;
  bool g_kin = true;
  unsigned g_ok = 0;           // generic loader predicates of the staged rows: bit r = A row, bit 4+r = W row
  float4 g_sub0 = make_float4(0.f, 0.f, 0.f, 0.f), g_sub1 = g_sub0;
#define ACX_GEN_SETUP(k0)                                                                          \
  do {                                                                                             \
    int kcol_ = (k0) + chunk * CE;                                                                 \
    g_dn = g_dl = 0;                                                                               \
    g_kc = kcol_;                                                                                  \
    if (d.amap == ACX_AMAP_CONV3X3) {                                                              \
      const int tap_ = (k0) / d.cin; /* uniform over the block (cin % KE == 0) */                  \
      g_dn = tap_ / 3 - 1;                                                                         \
      g_dl = tap_ - (tap_ / 3) * 3 - 1;                                                            \
      g_kc = kcol_ - tap_ * d.cin;                                                                 \
    }                                                                                              \
    g_kin = kcol_ < d.K;                                                                           \
    if (!g_kin) { g_kc = 0; kcol_ = chunk * CE; } /* clamp: loaded, then zeroed by the select */   \
    g_kw = kcol_ - chunk * CE;                                                                     \
    g_sub0 = g_sub1 = make_float4(0.f, 0.f, 0.f, 0.f);                                             \
    if (d.a_sub) { /* depends on k only: one load per step, not per row */                         \
      g_sub0 = ld4(d.a_sub + g_kc);                                                                \
      if constexpr (NLA == 2) g_sub1 = ld4(d.a_sub + g_kc + 4);                                    \
    }                                                                                              \
  } while (0)
#define ACX_GEN_LOAD_ROW(r)   /* issues the loads only; masking / a_sub happen in ACX_STORE_ROW (no vmcnt wait here) */ \
  do {                                                                                             \
    bool ok_ = a_ok[r] && g_kin;                                                                   \
    long srow_ = a_row[r];                                                                         \
    if (d.amap == ACX_AMAP_CONV3X3) {                                                              \
      int nn_ = a_n[r] + g_dn, ll_ = a_l[r] + g_dl;                                                \
      ok_ = ok_ && nn_ >= 0 && nn_ < d.gn && ll_ >= 0 && ll_ < d.gl;                               \
      nn_ = min(max(nn_, 0), d.gn - 1);                                                            \
      ll_ = min(max(ll_, 0), d.gl - 1);                                                            \
      srow_ += (long)nn_ * d.gl + ll_;                                                             \
    }                                                                                              \
    if constexpr (A_BF16) {                                                                        \
      const uint4 v_ = *reinterpret_cast<const uint4*>((const u16*)d.A + (size_t)srow_ * d.lda + g_kc); \
      ra##r = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); \
    } else {                                                                                       \
      const float* p_ = (const float*)d.A + (size_t)srow_ * d.lda + g_kc;                          \
      ra##r = ld4(p_);                                                                             \
      if constexpr (NLA == 2) rb##r = ld4(p_ + 4);                                                 \
    }                                                                                              \
    rw##r = *reinterpret_cast<const uint4*>(w_ptr[r] + (size_t)g_kw * WB);                        \
    g_ok = (g_ok & ~(0x11u << (r))) | (ok_ ? (1u << (r)) : 0u) | ((w_ok[r] && g_kin) ? (0x10u << (r)) : 0u); \
  } while (0)
#define ACX_GEN_LOAD(k0) \
  do { ACX_GEN_SETUP(k0); ACX_GEN_LOAD_ROW(0); ACX_GEN_LOAD_ROW(1); ACX_GEN_LOAD_ROW(2); ACX_GEN_LOAD_ROW(3); } while (0)

  // (macro, not a lambda: the row index must stay a compile-time constant or `ra`/`rw` are demoted
  //  to scratch memory -- guide rule 20)
#define ACX_STORE_ROW(stage, r)                                                        \
  do {                                                                                 \
    char* sA_ = smem + (stage) * 2 * TILE_B;                                           \
    char* sW_ = sA_ + TILE_B;                                                          \
    const int off_ = (rbase + 32 * (r)) * ROWB + chunk * 16;                           \
    float4 va_ = ra##r, vb_ = rb##r;                                                   \
    uint4 w_ = rw##r;                                                                  \
    if constexpr (!FAST) {   /* generic loader: recentre + zero-fill, one K-step after the loads were issued */ \
      if constexpr (!A_BF16) {                                                         \
        va_.x -= g_sub0.x; va_.y -= g_sub0.y; va_.z -= g_sub0.z; va_.w -= g_sub0.w;    \
        if constexpr (NLA == 2) { vb_.x -= g_sub1.x; vb_.y -= g_sub1.y; vb_.z -= g_sub1.z; vb_.w -= g_sub1.w; } \
      }                                                                                \
      if (!(g_ok & (1u << (r)))) { va_ = make_float4(0.f, 0.f, 0.f, 0.f); vb_ = va_; } \
      if (!(g_ok & (0x10u << (r)))) w_ = make_uint4(0, 0, 0, 0);                       \
    }                                                                                  \
    if constexpr (NLA == 2) {                                                          \
      *reinterpret_cast<uint4*>(sA_ + off_) = pack_bf16x8(va_, vb_);                   \
    } else {                                                                           \
      *reinterpret_cast<float4*>(sA_ + off_) = va_;                                    \
    }                                                                                  \
    *reinterpret_cast<uint4*>(sW_ + off_) = w_;                                        \
  } while (0)

  // ---- accumulators
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hh = lane >> 5;
  const int a_off = (wm * 64 + li) * ROWB + hh * 16;
  const int w_off = (wn * 64 + li) * ROWB + hh * 16;

  const int nk_total = (d.K + KE - 1) / KE;
  const int kbeg = (g.ksplit > 1 ? (int)blockIdx.y * g.kchunk : 0) * KE;       // first K element of this split
  const int nk = g.ksplit > 1 ? min(g.kchunk, nk_total - (int)blockIdx.y * g.kchunk) : nk_total;
  // ---- software-pipelined K loop (v2).  Fragment registers are double-buffered by hand (fa/fb sets X and Y);
  // the ONE barrier of a K-step sits between MFMA phases 2 and 3: by then this wave has written its share of the
  // next tile and already holds phase 3's fragments, so after the barrier it issues 16 MFMAs immediately and the
  // ds_reads of the NEXT tile's phase-0 fragments (and the global loads two tiles ahead) ride in their shadow.
  using frag_t = typename std::conditional<PREC == 0, float4, bf16x8>::type;
  frag_t xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
#define ACX_RD(SET, stage, q)                                                                             \
  do {                                                                                                    \
    const char* sA_ = smem + (stage) * 2 * TILE_B;                                                        \
    const char* sW_ = sA_ + TILE_B;                                                                       \
    SET##a0 = *reinterpret_cast<const frag_t*>(sA_ + a_off + (q) * 32);                                   \
    SET##a1 = *reinterpret_cast<const frag_t*>(sA_ + a_off + 32 * ROWB + (q) * 32);                       \
    SET##b0 = *reinterpret_cast<const frag_t*>(sW_ + w_off + (q) * 32);                                   \
    SET##b1 = *reinterpret_cast<const frag_t*>(sW_ + w_off + 32 * ROWB + (q) * 32);                       \
  } while (0)
#define ACX_MM(SET)                                                                                       \
  do {                                                                                                    \
    if constexpr (PREC == 0) {                                                                            \
      const float4& A0_ = reinterpret_cast<const float4&>(SET##a0);                                       \
      const float4& A1_ = reinterpret_cast<const float4&>(SET##a1);                                       \
      const float4& B0_ = reinterpret_cast<const float4&>(SET##b0);                                       \
      const float4& B1_ = reinterpret_cast<const float4&>(SET##b1);                                       \
      const float av_[2][4] = {{A0_.x, A0_.y, A0_.z, A0_.w}, {A1_.x, A1_.y, A1_.z, A1_.w}};               \
      const float bv_[2][4] = {{B0_.x, B0_.y, B0_.z, B0_.w}, {B1_.x, B1_.y, B1_.z, B1_.w}};               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                  \
          _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[mi][j], bv_[ni][j], acc[mi][ni], 0, 0, 0); \
    } else {                                                                                              \
      const bf16x8& A0_ = reinterpret_cast<const bf16x8&>(SET##a0);                                       \
      const bf16x8& A1_ = reinterpret_cast<const bf16x8&>(SET##a1);                                       \
      const bf16x8& B0_ = reinterpret_cast<const bf16x8&>(SET##b0);                                       \
      const bf16x8& B1_ = reinterpret_cast<const bf16x8&>(SET##b1);                                       \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0_, B0_, acc[0][0], 0, 0, 0);                  \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0_, B1_, acc[0][1], 0, 0, 0);                  \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1_, B0_, acc[1][0], 0, 0, 0);                  \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1_, B1_, acc[1][1], 0, 0, 0);                  \
    }                                                                                                     \
  } while (0)
#define ACX_LOAD_TILE(k0) do { if constexpr (FAST) ACX_FAST_LOAD(k0); else ACX_GEN_LOAD(k0); } while (0)

  ACX_LOAD_TILE(kbeg);
  ACX_STORE_ROW(0, 0); ACX_STORE_ROW(0, 1); ACX_STORE_ROW(0, 2); ACX_STORE_ROW(0, 3);
  __syncthreads();
  if (nk > 1) ACX_LOAD_TILE(kbeg + KE);
  ACX_RD(x, 0, 0);
#if ACX_TRACE
  tr1 = wall_clock64();
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nk;
    ACX_RD(y, cur, 1);
    ACX_MM(x);                                   // phase 0
    ACX_RD(x, cur, 2);
    if (more) { ACX_STORE_ROW(nxt, 0); ACX_STORE_ROW(nxt, 1); }
    ACX_MM(y);                                   // phase 1
    ACX_RD(y, cur, 3);
    if (more) { ACX_STORE_ROW(nxt, 2); ACX_STORE_ROW(nxt, 3); }
    ACX_MM(x);                                   // phase 2
    __syncthreads();                             // next tile complete in LDS; everyone holds its phase-3 fragments
    if (kt + 2 < nk) ACX_LOAD_TILE(kbeg + (kt + 2) * KE);
    if (more) ACX_RD(x, nxt, 0);
    ACX_MM(y);                                   // phase 3
  }
#undef ACX_RD
#undef ACX_MM
#undef ACX_LOAD_TILE

#if ACX_TRACE
  tr2 = wall_clock64();
#endif
  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // Structure matters: every value is fully computed (bias / residual / pos loads consumed) BEFORE the first
  // store of its group is issued.  A store issued between a pending load and its use makes hipcc wait
  // vmcnt(0) -- which on CDNA also waits for the previous STORE -- and serialises the 64 stores of a lane
  // into 64 HBM round trips (measured: 29 us of a 120 us tile).
  if (g.ksplit > 1) {      // raw partial tile; bias / activation / residual happen in splitk_reduce_kernel
    float* P = g.partial + (size_t)blockIdx.y * d.M * d.N;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + li;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + mi * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
          if (col < d.N && row < d.M) P[(size_t)row * d.N + col] = acc[mi][ni][r];
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + li;
    const bool cok = col < d.N;
    const int colc = cok ? col : d.N - 1;
    float bias = 0.f;
    if (d.bias) bias = d.bias[colc];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int rowb = m0 + wm * 64 + mi * 32 + 4 * hh;
      float outv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) outv[r] = 0.f;
      const bool has_res = FAST ? (RES != 0) : (d.residual != nullptr);
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rowb + (r & 3) + 8 * (r >> 2), d.M - 1);
          outv[r] = d.residual[(size_t)row * d.ldr + colc];
        }
      }
      if constexpr (!FAST) {
        if (d.pos0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rc = min(rowb + (r & 3) + 8 * (r >> 2), d.M - 1);
            const int l = rc % d.gl, n = (rc / d.gl) % d.gn;
            // reference order: (x + param_0) + param_1, then the residual (none on this GEMM)
            outv[r] += d.pos0[(size_t)n * d.N + colc] + d.pos1[(size_t)l * d.N + colc];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mi][ni][r] + bias;
        const int act = FAST ? ACT : d.act;
        if (act == ACX_ACT_QUICKGELU) v = acx_quickgelu(v);
        else if (act == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;
        outv[r] += v;
      }
      // all loads consumed: make that explicit so the store loop below carries no memory dependencies
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(outv[r]));   // pin: computed here, not inside the store's branch
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowb + (r & 3) + 8 * (r >> 2);
        if (cok && row < d.M) {
          if constexpr (C_BF16) ((u16*)d.C)[(size_t)row * d.ldc + col] = f2bf(outv[r]);
          else ((float*)d.C)[(size_t)row * d.ldc + col] = outv[r];
        }
      }
    }
  }
#if ACX_TRACE
  if (g.trace && t == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long* o = g.trace + (size_t)blockIdx.x * 6;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = wall_clock64(); o[4] = hwid; o[5] = xcc;
  }
#endif
}


#include "acx_gemm_w8.h"     // gemm_f32_w8_kernel
#include "acx_gemm_p256.h"   // gemm_f32_p256_kernel: persistent 256-wide strip stream
#include "acx_gemm_sk.h"     // gemm_f32_sk_kernel: few-row problems (text tower of a data-parallel rank), no K-loop barrier
#include "acx_gemm_bf16.h"   // gemm_bf16_dma_kernel, gemm_bf16_ring_kernel
#include "acx_gemm_p8.h"     // gemm_bf16_p8_kernel: the ring kernel's tile stream on a phase-interleaved schedule
#include "acx_gemm_x6.h"     // gemm_x6_p4_kernel: pairs = 6 products, one wave per SIMD, plane-reuse schedule
#include "acx_gemm_tn.h"     // gemm_tn_kernel, gemm_tn_w8_kernel, tn_reduce_kernel

template <int C_BF16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, acx_gemm_desc d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)d.M * d.N;
  if (i >= total) return;
  const int row = (int)(i / d.N), col = (int)(i - (int64_t)row * d.N);
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += part[(size_t)s * total + i];
  if (d.bias) v += d.bias[col];
  if (d.act == ACX_ACT_QUICKGELU) v = acx_quickgelu(v);
  else if (d.act == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;
  // same order as the unsplit kernel's epilogue: (residual + (pos0 + pos1)) + activation(acc + bias)
  float o = d.residual ? d.residual[(size_t)row * d.ldr + col] : 0.f;
  if (d.pos0) {
    const int l = row % d.gl, n = (row / d.gl) % d.gn;
    o += d.pos0[(size_t)n * d.N + col] + d.pos1[(size_t)l * d.N + col];
  }
  o += v;
  if constexpr (C_BF16 == 2) {               // ACX_BF16X3: three bf16 planes hi | mid | lo of the f32 value
    const size_t pe = (d.c_plane_rows ? (size_t)d.c_plane_rows : (size_t)d.M) * d.ldc, at = (size_t)row * d.ldc + col;
    const u16 h = f2bf(o);
    const float r1 = o - bf2f(h);
    const u16 m = f2bf(r1);
    ((u16*)d.C)[at] = h; ((u16*)d.C)[at + pe] = m; ((u16*)d.C)[at + 2 * pe] = f2bf(r1 - bf2f(m));
  } else if constexpr (C_BF16 == 1) ((u16*)d.C)[(size_t)row * d.ldc + col] = f2bf(o);
  else ((float*)d.C)[(size_t)row * d.ldc + col] = o;
}

// The same, four consecutive columns per lane (N, ldc, ldr multiples of 4, 16-byte aligned bias / residual, no positional
// epilogue): 16-byte loads of the pieces, 16- / 8-byte stores -- the pairs = 6 kernel's K split at a data-parallel rank's 4096
// rows pays this launch after every convolution (one column per lane: 24-28 us for 3 x 16.8 MB in, three planes out).
// OUT: 0 f32, 1 bf16, 2 three bf16 planes (ACX_BF16X3, or ACX_BF16X3P: K-panel layout).  Same arithmetic, same order.
template <int OUT>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ part, int splits, acx_gemm_desc d) {
  const int n4 = d.N >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total4 = (int64_t)d.M * n4;
  if (i >= total4) return;
  const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) * 4;
  const int64_t total = (int64_t)d.M * d.N;
  const float* p = part + (int64_t)row * d.N + col;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  int q = 0;
  for (; q + 4 <= splits; q += 4) {                      // four pieces in flight, added in piece order
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(p + (int64_t)(q + u) * total);
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
  }
  for (; q < splits; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)q * total);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  if (d.bias) { const float4 b4 = *reinterpret_cast<const float4*>(d.bias + col); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (d.act == ACX_ACT_QUICKGELU) v[e] = acx_quickgelu(v[e]);
    else if (d.act == ACX_ACT_LEAKYRELU) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
  }
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (d.residual) { const float4 r4 = *reinterpret_cast<const float4*>(d.residual + (size_t)row * d.ldr + col); o[0] = r4.x; o[1] = r4.y; o[2] = r4.z; o[3] = r4.w; }
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] += v[e];
  if constexpr (OUT == 2) {
    const size_t crows = d.c_plane_rows ? (size_t)d.c_plane_rows : (size_t)d.M;
    const size_t pe = crows * d.ldc;
    u16* dst = (u16*)d.C + (d.c_dtype == ACX_BF16X3P ? ((size_t)(col >> 5) * crows + row) * 32 + (col & 31) : (size_t)row * d.ldc + col);
    u16 h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = f2bf(o[e]);
      const float r1 = o[e] - bf2f(h[e]);
      m[e] = f2bf(r1);
      l[e] = f2bf(r1 - bf2f(m[e]));
    }
    *reinterpret_cast<uint2*>(dst) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(dst + pe) = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
    *reinterpret_cast<uint2*>(dst + 2 * pe) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  } else if constexpr (OUT == 1) {
    *reinterpret_cast<uint2*>((u16*)d.C + (size_t)row * d.ldc + col) =
        make_uint2((uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16), (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16));
  } else {
    *reinterpret_cast<float4*>((float*)d.C + (size_t)row * d.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

// Which problems the persistent strip-stream kernel (acx_gemm_p256.h) takes: plain row-major f32 A and C, no residual
// (with one, the epilogue's load-then-store wait costs what the larger tile gains: 123 vs 125 TFLOP/s on out-proj),
// >= 1024 tiles of 128x128 so every CU streams >= 8 strips.  linear() asks too: such a launch has no tail to split.
bool acx_gemm_takes_strip_stream(const acx_gemm_desc* d) {
  if (!ACX_DBG_SWITCH("P256", true)) return false;
  if (d->prec != ACX_PREC_F32 || d->a_dtype == ACX_BF16 || d->c_dtype == ACX_BF16) return false;
  if (d->a_sub || d->pos0 || d->K % 32 || d->residual) return false;
  if (d->amap == ACX_AMAP_CONV3X3) {
    // implicit-GEMM convolutions: wide outputs only (N = 256 at 32768 rows gives every CU one 128-row tile = 8 of its 16
    // waves; measured 134-135 TFLOP/s against 133-136 for gemm_f32_w8_kernel, with and without the residual: a tie, so
    // those stay where they were), power-of-two token grid, a caller-provided zero page for the taps outside the grid
    if (!d->zero_page || ((uintptr_t)d->zero_page & 15) || d->N < 512 || d->cin % 32 || d->act == ACX_ACT_QUICKGELU) return false;
    if ((d->gl & (d->gl - 1)) || ((d->gn * d->gl) & (d->gn * d->gl - 1)) || d->M % 64) return false;
  } else if (d->amap != ACX_AMAP_IDENTITY || d->act == ACX_ACT_LEAKYRELU) {
    return false;
  }
  const long tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
  return tiles >= 1024 && (size_t)d->M * d->lda < ((size_t)1 << 31) && (size_t)d->N * d->ldw < ((size_t)1 << 31);
}

// Column strips for a partly filled last round of tiles (acx_gemm, pairs = 6): cost of a 256 x 128 / 256 x 64 strip relative
// to a whole 256 x 256 tile (same A units, half / a quarter of the W units and MFMAs), and the switch (ACX_X6_STRIP=0 in tools
// builds; ACX_OPT_X6_STRIP_TAIL at run time)
// measured (profiles/r06_x6_strip_tail.txt): a round of 128-column strips takes 0.38-0.53 of a round of whole tiles at K = 768 (0.43-0.57
// at K = 3072), growing with the number of strips in flight; 64-column strips 0.29-0.33 (0.38-0.45).  Upper ends here: strips are
// taken for rem <= CUs / 2 (64-column ones for rem <= CUs / 4), which is where every measured case gains.
static inline double x6_strip_cost(int ni) { return ni == 2 ? 0.55 : 0.45; }
static inline bool x6_strip_enabled(const acx_ctx* ctx) { return ACX_DBG_SWITCH("X6_STRIP", true) && (!ctx || ctx->opt_x6_strip); }

// K split of the plane-reuse kernel: `tiles` output tiles of `nks` K-steps on ncu persistent workgroups.  Cost of s pieces =
// rounds(tiles s) x ceil(nks / s) K-steps (+ a term for the s partial images the reduce launch reads); every non-empty split
// is allowed (the last K range of a tile takes what is left).  A pure function of the shape: fixed summation order.
static int x6_choose_split(int tiles, int nks, int ncu, size_t image_bytes, size_t workspace_bytes, int min_steps, double* cost_us = nullptr) {
  int best = 1;
  double best_cost = 1e30;
  for (int s2 = 1; s2 <= 64; ++s2) {
    const int spi = (nks + s2 - 1) / s2;
    if (s2 > 1 && (spi < min_steps || spi * (s2 - 1) >= nks || (size_t)s2 * image_bytes > workspace_bytes)) continue;
    const int rounds = (tiles * s2 + ncu - 1) / ncu;
    const double cost = (double)rounds * (spi + 4.0) * 3.0 + (s2 > 1 ? (s2 + 1.0) * (double)image_bytes / 4.0e6 + 4.0 : 0.0);   // us
    if (cost < best_cost * 0.995) { best_cost = cost; best = s2; }
  }
  if (cost_us) *cost_us = best_cost;
  return best;
}

extern "C" int acx_gemm(acx_ctx* ctx, const acx_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: null pointer%s");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: empty shape%s");
  const int prec = d->prec;
  if (prec != ACX_PREC_F32 && prec != ACX_PREC_BF16)
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown precision%s");
  // ACX_F16: fp16 planes of the two-plane split (pairs = 3 only: the X3 = 2 instantiation of the plane-reuse kernel); handled with the
  // bf16 planes' checks below (16-bit elements)
  const int a_f16 = d->a_dtype == ACX_F16;
  if ((a_f16 && d->pairs != 3) || (d->c_dtype == ACX_F16X2P && !a_f16))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: fp16 planes (ACX_F16 / ACX_F16X2P) come with pairs = 3%s");
  const int a_bf16 = d->a_dtype == ACX_BF16 || a_f16, c_bf16 = d->c_dtype == ACX_BF16;
  if (prec == ACX_PREC_F32 && (a_bf16))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: PREC_F32 needs f32 A%s");
  const int kal = prec == ACX_PREC_F32 ? 4 : 8;
  if (d->K % kal || d->lda % (a_bf16 ? 8 : 4) || d->ldw % kal)
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: K/lda/ldw must be multiples of %s%ld elements", "", kal);
  if (((uintptr_t)d->A | (uintptr_t)d->W) & 15)
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: A/W must be 16-byte aligned%s");
  if (d->a_sub && (a_bf16 || ((uintptr_t)d->a_sub & 15)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: a_sub needs f32 A and 16-byte alignment%s");
  if (d->amap == ACX_AMAP_CONV3X3) {
    const int ke = (prec == ACX_PREC_F32 || d->pairs == 6 || d->pairs == 3) ? 32 : 64;   // (the plane-reuse kernel's K-steps are 32 wide)
    if (d->cin <= 0 || d->K != 9 * d->cin || d->cin % ke || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad conv3x3 geometry%s");
    if (d->a_sub) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: a_sub with conv3x3%s");
  } else if (d->amap == ACX_AMAP_TILETABLE) {
    if (!d->tile_table || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: TILETABLE needs tile_table and M %% (gn*gl) == 0%s");
  } else if (d->amap == ACX_AMAP_TESTTILE) {
    if (d->seg <= 0 || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl * d->seg))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad test-tile geometry%s");
  } else if (d->amap != ACX_AMAP_IDENTITY) {
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown amap%s");
  }
  if ((d->pos0 != nullptr) != (d->pos1 != nullptr) || (d->pos0 && (d->gn <= 0 || d->gl <= 0)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: pos0/pos1 need both pointers and a grid%s");

  // ---- pairs = 6, more 256 x 256 tiles than CUs, a last round that fills only part of the chip (the ViT's N = 768 products: 1182
  // tiles = 4.6 rounds at 512 frames, 591 = 2.3 at 256): the whole tile rows of the full rounds go out as one launch, the rest
  // as a second problem on its row range -- fewer tiles than CUs, so its K is split across workgroups and reduced by a launch
  // that also applies the epilogue (below; plane outputs: acx_gemm_desc.c_plane_rows).  Identity rows, caller-provided workspace; taken when
  // the split model says >= 1 % (x6_choose_split's units).  Opt-in (ACX_OPT_X6_TAIL_SPLIT): the tail rows sum K in another order than
  // the rows before them, and the default keeps identical rows of one launch bit-identical.  Decided before the profiling scope:
  // the two calls are two launches.
  if (d->pairs == 6 && (d->amap == ACX_AMAP_IDENTITY || d->amap == ACX_AMAP_CONV3X3) && d->workspace && prec == ACX_PREC_BF16 && a_bf16 &&
      !d->a_sub && !d->pos0 &&
      d->K % 32 == 0 && ctx && ctx->opt_x6_tail) {
    int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
    if (ctx && ctx->opt_x6_cus > 0 && ctx->opt_x6_cus < ncu) ncu = ctx->opt_x6_cus;
    const int tm = (d->M + 255) / 256, tn = (d->N + 255) / 256, xt = tm * tn, nks = d->K / 32;
    const int rounds = xt / ncu, rem = xt - rounds * ncu;
    if (rounds >= 1 && rem > 0) {
      int tm_main = (rounds * ncu) / tn;
      if (d->amap == ACX_AMAP_CONV3X3) {           // the tail must begin at a token-grid boundary (its taps never leave a grid)
        const int grid_tiles = (d->gn * d->gl + 255) / 256;
        tm_main = (d->gn * d->gl) % 256 == 0 ? tm_main / grid_tiles * grid_tiles : 0;
      }
      const int xtail = (tm - tm_main) * tn;
      const int64_t row0 = (int64_t)tm_main * 256, m_tail = d->M - row0;
      if (tm_main >= 1 && xtail < ncu && m_tail > 0) {
        double tail_us = 0.0;
        const int s_tail = x6_choose_split(xtail, nks, ncu, (size_t)m_tail * d->N * sizeof(float), d->workspace_bytes, 6, &tail_us);
        const double now_us = (rounds + 1) * (nks + 7.0) * 3.0, new_us = rounds * (nks + 7.0) * 3.0 + tail_us + 6.0;
#ifndef ACX_X6TAIL_THRESH
#define ACX_X6TAIL_THRESH 0.99
#endif
        if (s_tail > 1 && new_us < ACX_X6TAIL_THRESH * now_us) {
          acx_gemm_desc dm = *d, dt = *d;
          dm.M = (int)row0; dm.workspace = nullptr; dm.workspace_bytes = 0;
          const bool apanel = (d->panels & 1) != 0;
          dt.M = (int)m_tail;
          dt.A = (const char*)d->A + (apanel ? (size_t)row0 * 64 : (size_t)row0 * d->lda * 2);
          const bool c_planes = d->c_dtype == ACX_BF16X3 || d->c_dtype == ACX_BF16X3P;
          if (c_planes) dm.c_plane_rows = dt.c_plane_rows = d->c_plane_rows ? d->c_plane_rows : (int64_t)d->M;
          dt.C = (char*)d->C + (d->c_dtype == ACX_BF16X3P ? (size_t)row0 * 32 * 2
                                                          : (size_t)row0 * d->ldc * (d->c_dtype == ACX_F32 ? 4 : 2));
          if (d->residual) dt.residual = d->residual + (size_t)row0 * d->ldr;
          const int rc = acx_gemm(ctx, &dm, stream);
          return rc ? rc : acx_gemm(ctx, &dt, stream);
        }
      }
    }
  }

  Args g;
  g.d = *d;
  const int tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
#if ACX_TRACE
  g.trace = getenv("ACX_TRACE_PTR") ? (long long*)strtoull(getenv("ACX_TRACE_PTR"), nullptr, 0) : nullptr;
#else
  g.trace = nullptr;
#endif
  dim3 grid((unsigned)(tiles_m * g.tiles_n)), block(NTHREADS);
  g.ksplit = 1; g.kchunk = 0; g.partial = nullptr; g.counters = nullptr;
  g.tile0 = 0; g.ntiles = 0;
  g.zeros = (const float*)d->zero_page;
  const size_t lds = 4 * TILE_B;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_GEMM, (hipStream_t)stream);
  if (ctx && ctx->prof_on) ctx->prof_gemm_flops += 2.0 * d->M * (double)d->N * d->K;
  // hipFuncSetAttribute is a per-DEVICE setting: the "already done" flags of the launch macros below are indexed by the
  // context's device, so one process may drive several GPUs
  const int dev_slot = (ctx ? ctx->device : 0) & 63;
#define ACX_LAUNCH(P, AB, CB, F, ACT, RES)                                                          \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_kernel<P, AB, CB, F, ACT, RES>,                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_kernel<P, AB, CB, F, ACT, RES>), grid, block, lds, s, g);              \
  } while (0)
  const int ke = prec == ACX_PREC_F32 ? 32 : 64;
  const bool fast = d->amap == ACX_AMAP_IDENTITY && !d->a_sub && !d->pos0 && d->K % ke == 0 &&
                    d->act != ACX_ACT_LEAKYRELU;
  const int variant = (prec == ACX_PREC_F32 ? 0 : (a_bf16 ? 1 : 2)) * 2 + c_bf16;   // 0..5
#define ACX_FAST(P, AB, CB)                                                           \
  do {                                                                                \
    if (d->act == ACX_ACT_QUICKGELU) {                                                \
      if (d->residual) ACX_LAUNCH(P, AB, CB, 1, 1, 1); else ACX_LAUNCH(P, AB, CB, 1, 1, 0); \
    } else {                                                                          \
      if (d->residual) ACX_LAUNCH(P, AB, CB, 1, 0, 1); else ACX_LAUNCH(P, AB, CB, 1, 0, 0); \
    }                                                                                 \
  } while (0)
  // few-row f32 problems (and every problem that asks for the few-row fusions): 32x32 tiles, K split over the waves
  {
    const bool a_norm = d->a_norm_w != nullptr;
    const bool sk_fusion = d->a_act != ACX_ACT_NONE || d->gelu_grad_of != nullptr || a_norm;
    const bool sk_ok = fast && prec == ACX_PREC_F32 && !c_bf16 && !a_bf16 && d->K % SK_CH == 0 && d->N % 4 == 0 && d->ldc % 4 == 0 &&
                       !((uintptr_t)d->C & 15) && (!d->residual || (d->ldr % 4 == 0 && !((uintptr_t)d->residual & 15))) &&
                       (!d->gelu_grad_of || (d->ldg % 4 == 0 && !((uintptr_t)d->gelu_grad_of & 15) && !d->residual &&
                                             d->act == ACX_ACT_NONE)) &&
                       !(d->act == ACX_ACT_QUICKGELU && d->residual) && (d->a_act == ACX_ACT_NONE || d->a_act == ACX_ACT_QUICKGELU) &&
                       (size_t)d->M * d->lda < ((size_t)1 << 31) && (size_t)d->N * d->ldw < ((size_t)1 << 31);
    if (sk_fusion && !sk_ok)
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: a_act / gelu_grad_of / a_norm need the few-row f32 kernel (see acx_gemm_desc)%s");
    if (a_norm && (d->K != 512 || !d->a_norm_b || d->a_act != ACX_ACT_NONE || d->residual || d->gelu_grad_of || d->act != ACX_ACT_NONE ||
                   (((uintptr_t)d->a_norm_w | (uintptr_t)d->a_norm_b) & 15)))
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: a_norm needs K == 512, weight and bias, and no other fusion%s");
    // narrow outputs (N <= 512: out-proj, proj and the dX chain of the text tower) stay ahead of the 64x64-tile kernel up to
    // ~1300 rows -- 17-34 row tiles x 16 column tiles fill the chip where 64x64 tiles leave half of it idle
    // (profiles/r03_text_gemm.txt); wide outputs only up to the row limit
    const int sk_max_m = ctx ? ctx->opt_sk_max_m : ACX_SK_MAX_M;
    // ... except long K at more than ~770 rows when the caller brought a split-K workspace: the 64x64 kernel with K split 2-4
    // ways is ahead there (1078 rows, N = 512: K = 2048 37.1 vs 43.5 us, K = 1536 30.0 vs 33.2 us; at 539 rows it is not)
    const bool sk_narrow = sk_max_m > 0 && d->N <= 512 && d->M <= 4 * sk_max_m && !(d->workspace && d->M > 768 && d->K >= 1536);
    if (sk_ok && (sk_fusion || d->M <= sk_max_m || sk_narrow)) {
      dim3 kgrid((unsigned)(((d->M + 31) / 32) * ((d->N + 31) / 32)));
      // long K on few tiles: pieces of two chunks (512 floats = resident in one DMA burst) across workgroups, when the caller
      // brought the partial workspace and the arrival counters
      g.counters = (unsigned int*)d->counters;
      if (d->workspace && d->counters && d->K >= 1024 && d->K % 512 == 0) {
        const int pieces = d->K / 512;
        if ((int)kgrid.x * pieces <= 512 && (int)kgrid.x <= d->n_counters &&
            (size_t)pieces * kgrid.x * 4096 <= d->workspace_bytes && !((uintptr_t)d->workspace & 15)) {
          g.ksplit = pieces;
          g.kchunk = 2;
          g.partial = (float*)d->workspace;
          kgrid.y = (unsigned)pieces;
        }
      }
#define ACX_SKL(E, AG)                                                                              \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_f32_sk_kernel<E, AG>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_B);         \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_f32_sk_kernel<E, AG>), kgrid, dim3(512), (size_t)SK_LDS_B, s, g);      \
  } while (0)
      if (a_norm) {
        static bool attrn_dev_[64] = {}; bool& attrn_done = attrn_dev_[dev_slot];
        if (!attrn_done) {
          (void)hipFuncSetAttribute((const void*)gemm_f32_sk_kernel<SK_EPI_PLAIN, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)SK_LDS_B);
          attrn_done = true;
        }
        hipLaunchKernelGGL((gemm_f32_sk_kernel<SK_EPI_PLAIN, 0, 1>), kgrid, dim3(512), (size_t)SK_LDS_B, s, g);
        ACX_CHECK_LAUNCH(ctx, "acx_gemm");
        return ACX_OK;
      }
      const int epi = d->gelu_grad_of ? SK_EPI_GELUGRAD : d->residual ? SK_EPI_RES : d->act == ACX_ACT_QUICKGELU ? SK_EPI_QUICKGELU : SK_EPI_PLAIN;
      if (d->a_act == ACX_ACT_QUICKGELU) {
        switch (epi) {
          case SK_EPI_PLAIN: ACX_SKL(SK_EPI_PLAIN, 1); break;
          case SK_EPI_QUICKGELU: ACX_SKL(SK_EPI_QUICKGELU, 1); break;
          case SK_EPI_RES: ACX_SKL(SK_EPI_RES, 1); break;
          default: ACX_SKL(SK_EPI_GELUGRAD, 1); break;
        }
      } else {
        switch (epi) {
          case SK_EPI_PLAIN: ACX_SKL(SK_EPI_PLAIN, 0); break;
          case SK_EPI_QUICKGELU: ACX_SKL(SK_EPI_QUICKGELU, 0); break;
          case SK_EPI_RES: ACX_SKL(SK_EPI_RES, 0); break;
          default: ACX_SKL(SK_EPI_GELUGRAD, 0); break;
        }
      }
#undef ACX_SKL
      ACX_CHECK_LAUNCH(ctx, "acx_gemm");
      return ACX_OK;
    }
  }
  // ---- pairs = 6 (f32-accurate product from three bf16 planes per operand): the one-wave-per-SIMD plane-reuse kernel
  // (acx_gemm_x6.h) -- identity rows or the implicit 3x3 convolution; K split across workgroups when the tiles alone
  // would not fill the chip (caller-provided workspace)
  if ((d->pairs == 6 || d->pairs == 3) && ACX_DBG_SWITCH("X6P4", true)) {
    const bool conv = d->amap == ACX_AMAP_CONV3X3;
    const bool x3 = d->pairs == 3;               // the three leading products only (acx_gemm_x6.h, X3): identity rows
    if (x3 && (conv || c_bf16 || d->act == ACX_ACT_LEAKYRELU))
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: pairs = 3 takes identity rows, f32 or plane outputs, bias / QuickGELU / residual epilogues%s");
    const bool c_x3_ = d->c_dtype == ACX_BF16X3 || d->c_dtype == ACX_BF16X3P || d->c_dtype == ACX_F16X2P;
    if (a_f16 && c_x3_ && d->c_dtype != ACX_F16X2P)
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: fp16 operand planes write fp16 output planes (ACX_F16X2P)%s");
    const bool shape_ok = prec == ACX_PREC_BF16 && a_bf16 && (d->amap == ACX_AMAP_IDENTITY || conv) && !d->a_sub && !d->pos0 &&
        d->K % 32 == 0 && d->lda % 8 == 0 && d->ldw % 8 == 0 && d->N % 4 == 0 && d->ldc % 4 == 0 && (!d->residual || d->ldr % 4 == 0) &&
        !(((uintptr_t)d->C | (uintptr_t)d->residual | (uintptr_t)d->bias) & 15) && d->a_plane_stride > 0 && d->w_plane_stride > 0 &&
        !((d->a_plane_stride | d->w_plane_stride) & 15) && (size_t)d->M * d->lda * 2 < ((size_t)1 << 32) &&
        (size_t)d->N * d->ldw * 2 < ((size_t)1 << 32) && !(d->act == ACX_ACT_QUICKGELU && d->residual) &&
        !(d->act == ACX_ACT_LEAKYRELU && d->residual) && !(c_x3_ && (d->residual || d->N % 8 || d->ldc % 8)) && !(c_bf16 && d->residual) &&
        (!d->panels || (!conv && d->K % 32 == 0 && d->lda == d->K && d->ldw == d->K)) && ((d->c_dtype != ACX_BF16X3P && d->c_dtype != ACX_F16X2P) || d->ldc == d->N) &&
        (!conv || (d->zero_page && !((uintptr_t)d->zero_page & 15) && d->cin % 32 == 0 && !(d->gl & (d->gl - 1)) &&
                   !(d->gn & (d->gn - 1)) && d->M % 256 == 0));
    if (shape_ok) {
      int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
      if (ctx && ctx->opt_x6_cus > 0 && ctx->opt_x6_cus < ncu) ncu = ctx->opt_x6_cus;   // ACX_OPT_X6_CUS
      // narrow convolution outputs (N <= 128: c2 and the input gradient of c1 in the XD-Violence head, E = 128): 256 x 128 tiles
      // (the NI = 2 instantiation) instead of half-empty 256 x 256 ones; f32 outputs without an activation only (what the head asks for)
      const bool narrow = conv && d->N <= 128 && !c_x3_ && !c_bf16 && d->act == ACX_ACT_NONE;
      const int xt = ((d->M + 255) / 256) * (narrow ? 1 : (d->N + 255) / 256);
      const int nks = d->K / 32;
      int split = 1;
      double split_us = 0.0;
      if (d->workspace && xt < ncu && !(a_f16 && c_x3_))   // (no reduce launch writes fp16 planes: such launches keep whole K per tile)
        split = x6_choose_split(xt, nks, ncu, (size_t)d->M * d->N * sizeof(float), d->workspace_bytes, 6, &split_us);
      // A partly filled LAST round of tiles (N = 768 at 256 frames: 591 tiles = 2.3 rounds of 256 CUs) is cut into column STRIPS:
      // the full rounds go out as whole tiles, the remaining `rem` tiles as 2 rem strips of 128 columns or 4 rem strips of 64
      // (the NI = 2 / 1 instantiations, acx_gemm_x6.h) when that makes the last round shorter.  Strips leave every row's K order
      // and product order untouched -- results are bit-identical to whole tiles (test_gemm_x6_strip_tail_bit_identical) -- so
      // identical rows of one launch stay identical wherever they sit, which the K split of the tail (ACX_OPT_X6_TAIL_SPLIT) gave up.
      // Cost of a strip relative to a whole tile: measured (profiles/r06_x6_strip_tail.txt).  Fewer tiles than workgroups: strips
      // compete with the K split by the same cost model (x6_choose_split's units).
      int strip_ni = 4, tile0_tail = 0, rem_tail = 0;
      if (!conv && d->N % 256 == 0 && !c_bf16 && d->act != ACX_ACT_LEAKYRELU && x6_strip_enabled(ctx)) {
        const int rounds = xt / ncu, rem = xt - rounds * ncu;
        if (rem > 0) {
          const double c2 = x6_strip_cost(2), c1 = x6_strip_cost(1);
          const double cost2 = (double)((2 * rem + ncu - 1) / ncu) * c2, cost1 = (double)((4 * rem + ncu - 1) / ncu) * c1;
          const double best = cost1 < cost2 ? cost1 : cost2;
          const int force = ctx ? ctx->opt_x6_strip : 1;    // 2 / 3: always 128- / 64-column strips (measurements)
          const double strip_us = (rounds + best) * (nks + 4.0) * 3.0;
          if (force >= 2) { strip_ni = force == 2 ? 2 : 1; tile0_tail = rounds * ncu; rem_tail = rem; }
          else if (best < 0.93 && (split == 1 || (strip_us < split_us && !(ctx && ctx->opt_x6_tail)))) { strip_ni = cost1 < cost2 ? 1 : 2; tile0_tail = rounds * ncu; rem_tail = rem; }
          if (strip_ni < 4) split = 1;
        }
      }
      g.ksplit = split; g.kchunk = (nks + split - 1) / split; g.partial = split > 1 ? (float*)d->workspace : nullptr;
      const int items = (strip_ni < 4 ? tile0_tail : xt) * split;
      const dim3 xgrid((unsigned)(items < ncu ? items : ncu));
      if (strip_ni < 4) { g.tile0 = 0; g.ntiles = tile0_tail; }
#define ACX_X6L_(CM, ACT, RES, CV, NI_, GRID)                                                        \
  do {                                                                                              \
    if constexpr ((CV) == 0 && (CM) != 1 && (ACT) != 2) {   /* (the ViT's epilogues: f32 / plane outputs, bias, QuickGELU, residual) */ \
      if (x3 && a_f16) {                                                                            \
        static bool attr4_dev_[64] = {}; bool& attr4_done = attr4_dev_[dev_slot];                   \
        if (!attr4_done) {                                                                          \
          (void)hipFuncSetAttribute((const void*)gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 0, NI_, 0, 2>, \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS_B);     \
          attr4_done = true;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL((gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 0, NI_, 0, 2>), GRID, dim3(256), (size_t)X6_LDS_B, s, g); \
        break;                                                                                      \
      }                                                                                             \
      if (x3) {                                                                                     \
        static bool attr3_dev_[64] = {}; bool& attr3_done = attr3_dev_[dev_slot];                   \
        if (!attr3_done) {                                                                          \
          (void)hipFuncSetAttribute((const void*)gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 0, NI_, 0, 1>, \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS_B);     \
          attr3_done = true;                                                                        \
        }                                                                                           \
        hipLaunchKernelGGL((gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 0, NI_, 0, 1>), GRID, dim3(256), (size_t)X6_LDS_B, s, g); \
        break;                                                                                      \
      }                                                                                             \
    }                                                                                               \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                          \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_x6_p4_kernel<CM, ACT, RES, CV, 0, 0, NI_>,        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS_B);         \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_x6_p4_kernel<CM, ACT, RES, CV, 0, 0, NI_>), GRID, dim3(256), (size_t)X6_LDS_B, s, g); \
  } while (0)
#define ACX_X6L(CM, ACT, RES, CV) do { if (items > 0) ACX_X6L_(CM, ACT, RES, CV, 4, xgrid); } while (0)
#define ACX_X6SEL(CV)                                                                               \
  do {                                                                                              \
    if (split > 1) ACX_X6L(0, 0, 0, CV);                                                            \
    else if (c_x3_) { if (d->act == ACX_ACT_QUICKGELU) ACX_X6L(2, 1, 0, CV); else if (d->act == ACX_ACT_LEAKYRELU) ACX_X6L(2, 2, 0, CV); else ACX_X6L(2, 0, 0, CV); } \
    else if (c_bf16) { if (d->act == ACX_ACT_QUICKGELU) ACX_X6L(1, 1, 0, CV); else if (d->act == ACX_ACT_LEAKYRELU) ACX_X6L(1, 2, 0, CV); else ACX_X6L(1, 0, 0, CV); } \
    else if (d->residual) ACX_X6L(0, 0, 1, CV);                                                     \
    else if (d->act == ACX_ACT_QUICKGELU) ACX_X6L(0, 1, 0, CV);                                     \
    else if (d->act == ACX_ACT_LEAKYRELU) ACX_X6L(0, 2, 0, CV);                                     \
    else ACX_X6L(0, 0, 0, CV);                                                                      \
  } while (0)
      if (narrow) { if (d->residual && split == 1) ACX_X6L_(0, 0, 1, 1, 2, xgrid); else ACX_X6L_(0, 0, 0, 1, 2, xgrid); }
      else if (conv) ACX_X6SEL(1);
      else ACX_X6SEL(0);
      if (strip_ni < 4) {
        // the strips of the last round: identity rows only; epilogues of the ViT's products (plane outputs with / without QuickGELU,
        // f32 with a residual / QuickGELU / plain) -- anything else keeps whole tiles (x6_strip_epilogue_ok)
        g.tile0 = tile0_tail * (4 / strip_ni); g.ntiles = rem_tail * (4 / strip_ni);   // (in the strips' own numbering: N % 256 == 0)
        const int sitems = rem_tail * (4 / strip_ni);
        const dim3 sgrid((unsigned)(sitems < ncu ? sitems : ncu));
#define ACX_X6S(NI_)                                                                                 \
  do {                                                                                              \
    if (c_x3_) { if (d->act == ACX_ACT_QUICKGELU) ACX_X6L_(2, 1, 0, 0, NI_, sgrid); else ACX_X6L_(2, 0, 0, 0, NI_, sgrid); } \
    else if (d->residual) ACX_X6L_(0, 0, 1, 0, NI_, sgrid);                                         \
    else if (d->act == ACX_ACT_QUICKGELU) ACX_X6L_(0, 1, 0, 0, NI_, sgrid);                         \
    else ACX_X6L_(0, 0, 0, 0, NI_, sgrid);                                                          \
  } while (0)
        if (strip_ni == 2) ACX_X6S(2); else ACX_X6S(1);
#undef ACX_X6S
      }
#undef ACX_X6SEL
#undef ACX_X6L
#undef ACX_X6L_
      if (split > 1) {
        // (shape_ok: N, ldc, ldr multiples of 4, 16-byte aligned bias / residual / C, no positional epilogue)
        const int64_t total4 = (int64_t)d->M * (d->N / 4);
        const dim3 rgrid((unsigned)((total4 + 255) / 256));
        if (c_x3_) hipLaunchKernelGGL((splitk_reduce4_kernel<2>), rgrid, dim3(256), 0, s, (const float*)g.partial, split, *d);
        else if (c_bf16) hipLaunchKernelGGL((splitk_reduce4_kernel<1>), rgrid, dim3(256), 0, s, (const float*)g.partial, split, *d);
        else hipLaunchKernelGGL((splitk_reduce4_kernel<0>), rgrid, dim3(256), 0, s, (const float*)g.partial, split, *d);
      }
      ACX_CHECK_LAUNCH(ctx, "acx_gemm");
      return ACX_OK;
    }
    if (conv) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: pairs = 6 with CONV3X3 needs a power-of-two grid, cin %% 32 == 0, M %% 256 == 0, a zero page and 16-byte aligned planes%s");
  }
  const bool w8_conv = d->amap == ACX_AMAP_CONV3X3 && !d->a_sub && !d->pos0 && d->K % 32 == 0 && d->cin % 32 == 0 &&
                       prec == ACX_PREC_F32 && !c_bf16 && !a_bf16;
  // small branch-free f32 problems (text tower): 64x64 tiles, four blocks per CU, no split-K
  const bool s64 = ACX_DBG_SWITCH("S64", true);
  const int st_m = (d->M + 63) / 64, st_n = (d->N + 63) / 64;
  if (s64 && fast && prec == ACX_PREC_F32 && !c_bf16 && !a_bf16 && tiles_m * g.tiles_n <= 256 && st_m * st_n >= 96) {
    const size_t lds_s = 4 * TILE_S;
    dim3 sgrid((unsigned)(st_m * st_n));
    // long K on few tiles (text tower: N = 512, K = 1536 / 2048 at 539-1078 rows = 72-136 tiles of 48-64 serial K-steps):
    // split K so that the launch covers the chip about twice, at least 16 K-steps per piece
    if (d->workspace && st_m * st_n <= 256 && d->K >= 1024) {
      int split = 1;
      const int nkt = d->K / 32;
      while (st_m * st_n * split * 2 <= 1100 && nkt / (split * 2) >= 12 && split < 8) split *= 2;
      if (split > 1 && (size_t)split * d->M * d->N * sizeof(float) <= d->workspace_bytes) {
        g.kchunk = (nkt + split - 1) / split;
        g.ksplit = (nkt + g.kchunk - 1) / g.kchunk;
        g.partial = (float*)d->workspace;
        sgrid.y = (unsigned)g.ksplit;
      }
    }
#define ACX_S64L(ACT, RES)                                                                          \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_f32_s64_kernel<ACT, RES>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);            \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_f32_s64_kernel<ACT, RES>), sgrid, dim3(256), lds_s, s, g);             \
  } while (0)
    if (d->act == ACX_ACT_QUICKGELU) { if (d->residual) ACX_S64L(1, 1); else ACX_S64L(1, 0); }
    else { if (d->residual) ACX_S64L(0, 1); else ACX_S64L(0, 0); }
#undef ACX_S64L
    if (g.ksplit > 1) {
      const int64_t total = (int64_t)d->M * d->N;
      hipLaunchKernelGGL((splitk_reduce_kernel<0>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)g.partial,
                         g.ksplit, *d);
    }
    ACX_CHECK_LAUNCH(ctx, "acx_gemm");
    return ACX_OK;
  }
  const bool w8 = ACX_DBG_SWITCH("W8", true);   // ACX_W8=0 (debug builds) keeps the 4-wave kernels
  // ... and f32 problems on the generic row-mapped path (a_sub / positional epilogue / test tilings: the temporal model's input
  // projection, 64 tiles at a data-parallel rank's 4096 rows = a quarter of the chip for 59 us): same split, same reduce
  const bool gen_split = !fast && prec == ACX_PREC_F32 && !a_bf16 && !c_bf16 && d->amap != ACX_AMAP_CONV3X3 && d->K % ke == 0 &&
                         d->act != ACX_ACT_LEAKYRELU;
  if ((fast || (w8_conv && w8) || gen_split) && d->workspace) {
    // skinny problems (few tiles, long K): split K over gridDim.y so the chip is filled; partial sums are
    // combined in fixed order by splitk_reduce_kernel together with the epilogue
    const int tiles = tiles_m * g.tiles_n, nkt = d->K / ke;
    int split = 1;
    // bf16 pieces keep >= 8 K-steps: at the XD text tower's 539 rows x K = 512 a 2-way split + its 5 us reduce launch is
    // slower than the unsplit launch (head step 1.64 -> 1.51 ms); 16 is worse again (1.59)
    while (tiles * split * 2 <= 512 && nkt / (split * 2) >= (prec == ACX_PREC_BF16 ? 8 : 4) && split < 16) split *= 2;
    if (split > 1 && (size_t)split * d->M * d->N * sizeof(float) <= d->workspace_bytes) {
      g.ksplit = split;
      g.kchunk = (nkt + split - 1) / split;
      g.ksplit = (nkt + g.kchunk - 1) / g.kchunk;
      g.partial = (float*)d->workspace;
      grid.y = g.ksplit;
    }
  }
  // f32 FAST problems without split-K: the 8-wave variant (ACX_W8=0 keeps the 4-wave kernel)
  // (short split-K pieces of identity-map problems stay on the 4-wave kernel: no measurable difference)
  if (g.ksplit == 1 && acx_gemm_takes_strip_stream(d)) {
    // big f32 GEMMs without a residual: persistent strip-stream kernel, one 1024-thread block per CU (acx_gemm_p256.h)
    const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
    const dim3 pgrid((unsigned)ncu);                       // >= 1024 tiles of 128x128 => >= 8 strips per CU
    const size_t plds = 2 * P2_STAGE_B;
#define ACX_P2L(ACT, CV)                                                                            \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                          \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_f32_p256_kernel<ACT, 0, CV>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);             \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_f32_p256_kernel<ACT, 0, CV>), pgrid, dim3(1024), plds, s, g);          \
  } while (0)
    if (d->amap == ACX_AMAP_CONV3X3) { if (d->act == ACX_ACT_LEAKYRELU) ACX_P2L(2, 1); else ACX_P2L(0, 1); }
    else if (d->act == ACX_ACT_QUICKGELU) ACX_P2L(1, 0); else ACX_P2L(0, 0);
#undef ACX_P2L
    ACX_CHECK_LAUNCH(ctx, "acx_gemm");
    return ACX_OK;
  }
  if (w8 && (w8_conv || (fast && g.ksplit == 1)) && prec == ACX_PREC_F32 && !c_bf16 && !a_bf16) {
#define ACX_W8L(ACT, RES, CV)                                                                       \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_f32_w8_kernel<ACT, RES, CV>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_f32_w8_kernel<ACT, RES, CV>), grid, dim3(512), lds, s, g);             \
  } while (0)
    // split pieces meet inside the kernel (last-arriver reduction) when the caller brought arrival counters; else a reduce launch
    g.counters = (g.ksplit > 1 && d->counters && (int)grid.x <= d->n_counters) ? (unsigned int*)d->counters : nullptr;
    if (w8_conv) {
      if (d->act == ACX_ACT_LEAKYRELU) { if (d->residual) ACX_W8L(2, 1, 1); else ACX_W8L(2, 0, 1); }
      else if (d->act == ACX_ACT_QUICKGELU) { if (d->residual) ACX_W8L(1, 1, 1); else ACX_W8L(1, 0, 1); }
      else { if (d->residual) ACX_W8L(0, 1, 1); else ACX_W8L(0, 0, 1); }
    } else {
      if (d->act == ACX_ACT_QUICKGELU) { if (d->residual) ACX_W8L(1, 1, 0); else ACX_W8L(1, 0, 0); }
      else { if (d->residual) ACX_W8L(0, 1, 0); else ACX_W8L(0, 0, 0); }
    }
#undef ACX_W8L
    if (g.ksplit > 1 && !g.counters) {
      const int64_t total = (int64_t)d->M * d->N;
      hipLaunchKernelGGL((splitk_reduce_kernel<0>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)g.partial,
                         g.ksplit, *d);
    }
    ACX_CHECK_LAUNCH(ctx, "acx_gemm");
    return ACX_OK;
  }
  // bf16 operands already in global memory, no split: the LDS-DMA kernels (ACX_NO_DMA=1 keeps the register-staged
  // one, ACX_NO_RING=1 the 128x128 DMA kernel).  Large problems take the persistent 256x256 ring kernel.
  const bool no_dma = !ACX_DBG_SWITCH("DMA", true);
  const bool no_ring = !ACX_DBG_SWITCH("RING", true);
  const int rtiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
  const int ring_min = ctx ? ctx->opt_ring_min_tiles : 512;   // acx_set_option(ACX_OPT_RING_MIN_TILES); tests lower it
  const bool ring_ok = fast && g.ksplit == 1 && prec == ACX_PREC_BF16 && a_bf16 && !no_dma && !no_ring && d->lda % 8 == 0 &&
      d->ldw % 8 == 0 && d->K % 64 == 0 && d->N % 4 == 0 && d->ldc % 4 == 0 && (!d->residual || d->ldr % 4 == 0) &&
      !(((uintptr_t)d->C | (uintptr_t)d->residual | (uintptr_t)d->bias) & 15) && rtiles >= ring_min &&
      !(d->act == ACX_ACT_QUICKGELU && d->residual);
  const bool c_x3 = d->c_dtype == ACX_BF16X3;
  if (c_x3 && (!ring_ok || d->residual || ((d->K / 64) * (d->pairs > 1 ? d->pairs : 1)) % 2 || !ACX_DBG_SWITCH("P8", true)))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: c_dtype ACX_BF16X3 needs the persistent 256x256 bf16 kernel and no residual%s");
  // (pairs = 6 problems were taken by the plane-reuse kernel above; the PAIRS instantiation of the kernel below -- round 4's route,
  // six plain products one after the other -- is no longer built)
  if (d->pairs == 3) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: pairs = 3 needs bf16 planes, identity rows, K %% 32 == 0 (the plane-reuse kernel)%s");
  if (d->pairs > 1)
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: pairs = 6 needs bf16 planes (16-byte aligned strides), K %% 32 == 0, N %% 4 == 0 (plane output: N %% 8 == 0), identity rows or CONV3X3 on a power-of-two grid%s");
  if (ring_ok) {
    const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
    const dim3 rgrid((unsigned)(rtiles < ncu ? rtiles : ncu));
    // even K-tile count and 32-bit operand byte offsets: the phase-interleaved kernel; else the lock-step ring kernel
    const int npairs_ = d->pairs > 1 ? d->pairs : 1;
    const bool p8 = ((d->K / 64) * npairs_) % 2 == 0 && (size_t)d->M * d->lda * 2 < ((size_t)1 << 32) &&
                    (size_t)d->N * d->ldw * 2 < ((size_t)1 << 32) && ACX_DBG_SWITCH("P8", true);
    if ((d->pairs > 1 || c_x3) && !p8)
      return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: pairs = 6 / ACX_BF16X3 output need the persistent 256x256 kernel%s");
#ifdef ACX_DEBUG_SWITCHES
    // A/B only (tools build, ACX_P4=1): the one-wave-per-SIMD frame's PLAIN schedule (acx_gemm_x6.h: 64-k super-steps, 64 MFMAs per
    // wave and barrier).  Measured equal to or 3-15 % behind the p8 kernel on the ViT shapes, its K loop alone (no DMA, no stores)
    // at 0.46 of the bf16 roof (profiles/r05_gemm_plain_bf16_notes.txt): not a product route.
    if (p8 && !c_x3 && !(c_bf16 && d->residual) && d->act != ACX_ACT_LEAKYRELU && ACX_DBG_SWITCH("P4", false)) {
      g.d.panels = 0; g.ksplit = 1; g.partial = nullptr;
#define ACX_P4L(CM, ACT, RES)                                                                       \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                          \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 1>,              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS_B);         \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_x6_p4_kernel<CM, ACT, RES, 0, 0, 1>), rgrid, dim3(256), (size_t)X6_LDS_B, s, g); \
  } while (0)
      if (d->residual) ACX_P4L(0, 0, 1);
      else if (d->act == ACX_ACT_QUICKGELU) { if (c_bf16) ACX_P4L(1, 1, 0); else ACX_P4L(0, 1, 0); }
      else { if (c_bf16) ACX_P4L(1, 0, 0); else ACX_P4L(0, 0, 0); }
#undef ACX_P4L
      ACX_CHECK_LAUNCH(ctx, "acx_gemm");
      return ACX_OK;
    }
#endif
#define ACX_RING_L(CB, ACT, RES)                                                                    \
  do {                                                                                              \
    if (p8) {                                                                                       \
      static bool attr8_dev_[64] = {}; bool& attr8_done = attr8_dev_[dev_slot];                                                               \
      if (!attr8_done) {                                                                            \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_p8_kernel<CB, ACT, RES>,                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_B);       \
        attr8_done = true;                                                                          \
      }                                                                                             \
      hipLaunchKernelGGL((gemm_bf16_p8_kernel<CB, ACT, RES>), rgrid, dim3(512), (size_t)P8_LDS_B, s, g); \
      break;                                                                                        \
    }                                                                                               \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_ring_kernel<CB, ACT, RES>,                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * RG_STAGE_B + 8 * 4096)); \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<CB, ACT, RES>), rgrid, dim3(512), (size_t)(2 * RG_STAGE_B + 8 * 4096), s, g); \
  } while (0)
#define ACX_RING_SEL()                                                                              \
  do {                                                                                              \
    if (c_x3) { if (d->act == ACX_ACT_QUICKGELU) ACX_RING_L(2, 1, 0); else ACX_RING_L(2, 0, 0); }   \
    else if (d->residual) { if (c_bf16) ACX_RING_L(1, 0, 1); else ACX_RING_L(0, 0, 1); }            \
    else if (d->act == ACX_ACT_QUICKGELU) { if (c_bf16) ACX_RING_L(1, 1, 0); else ACX_RING_L(0, 1, 0); } \
    else { if (c_bf16) ACX_RING_L(1, 0, 0); else ACX_RING_L(0, 0, 0); }                             \
  } while (0)
    ACX_RING_SEL();
#undef ACX_RING_SEL
#undef ACX_RING_L
  } else
  if (prec == ACX_PREC_BF16 && a_bf16 && !no_dma && d->amap == ACX_AMAP_CONV3X3 && g.ksplit == 1 && d->zero_page &&
      !((uintptr_t)d->zero_page & 15) && d->cin % 64 == 0 && d->M % 128 == 0 && d->lda % 8 == 0 && d->ldw % 8 == 0 && !d->a_sub &&
      !d->pos0 && d->act != ACX_ACT_QUICKGELU && !(d->act == ACX_ACT_LEAKYRELU && d->residual)) {
    // bf16 implicit-GEMM convolutions (the XD-Violence head): the 128x128 LDS-DMA kernel with a per-tap source row
    const size_t dlds = 2 * DMA_STAGE_B;
    g.zeros = (const float*)d->zero_page;
#define ACX_DMAC(CB, ACT, RES)                                                                      \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<CB, ACT, RES, 1>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds);             \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<CB, ACT, RES, 1>), grid, block, dlds, s, g);           \
  } while (0)
    if (d->act == ACX_ACT_LEAKYRELU) { if (c_bf16) ACX_DMAC(1, 2, 0); else ACX_DMAC(0, 2, 0); }
    else if (d->residual) { if (c_bf16) ACX_DMAC(1, 0, 1); else ACX_DMAC(0, 0, 1); }
    else { if (c_bf16) ACX_DMAC(1, 0, 0); else ACX_DMAC(0, 0, 0); }
#undef ACX_DMAC
  } else
  if (fast && g.ksplit == 1 && prec == ACX_PREC_BF16 && a_bf16 && !no_dma && d->lda % 8 == 0 && d->ldw % 8 == 0) {
    const size_t dlds = 2 * DMA_STAGE_B;
#define ACX_DMA(CB, ACT, RES)                                                                       \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                                                                 \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<CB, ACT, RES>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds);             \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_dma_kernel<CB, ACT, RES>), grid, block, dlds, s, g);              \
  } while (0)
    const int sel = (c_bf16 ? 4 : 0) + (d->act == ACX_ACT_QUICKGELU ? 2 : 0) + (d->residual ? 1 : 0);
    switch (sel) {
      case 0: ACX_DMA(0, 0, 0); break;
      case 1: ACX_DMA(0, 0, 1); break;
      case 2: ACX_DMA(0, 1, 0); break;
      case 3: ACX_DMA(0, 1, 1); break;
      case 4: ACX_DMA(1, 0, 0); break;
      case 5: ACX_DMA(1, 0, 1); break;
      case 6: ACX_DMA(1, 1, 0); break;
      default: ACX_DMA(1, 1, 1); break;
    }
#undef ACX_DMA
  } else if (fast) {
    switch (variant) {
      case 0: ACX_FAST(0, 0, 0); break;
      case 1: ACX_FAST(0, 0, 1); break;
      case 2: ACX_FAST(1, 1, 0); break;
      case 3: ACX_FAST(1, 1, 1); break;
      case 4: ACX_FAST(1, 0, 0); break;
      default: ACX_FAST(1, 0, 1); break;
    }
  } else {
    switch (variant) {
      case 0: ACX_LAUNCH(0, 0, 0, 0, 0, 0); break;
      case 1: ACX_LAUNCH(0, 0, 1, 0, 0, 0); break;
      case 2: ACX_LAUNCH(1, 1, 0, 0, 0, 0); break;
      case 3: ACX_LAUNCH(1, 1, 1, 0, 0, 0); break;
      case 4: ACX_LAUNCH(1, 0, 0, 0, 0, 0); break;
      default: ACX_LAUNCH(1, 0, 1, 0, 0, 0); break;
    }
  }
#undef ACX_FAST
#undef ACX_LAUNCH
  if (g.ksplit > 1) {
    const int64_t total = (int64_t)d->M * d->N;
    const dim3 rgrid((unsigned)((total + 255) / 256));
    if (c_bf16) hipLaunchKernelGGL((splitk_reduce_kernel<1>), rgrid, dim3(256), 0, s, (const float*)g.partial, g.ksplit, *d);
    else hipLaunchKernelGGL((splitk_reduce_kernel<0>), rgrid, dim3(256), 0, s, (const float*)g.partial, g.ksplit, *d);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_gemm");
  return ACX_OK;
}

// Split count of the M reduction: a pure function of the shape (so results are run-to-run identical).
// Cost model in units of "rows of m streamed by one resident block": rounds(s) * (rows per split + fixed
// per-block overhead) + the partial-tile traffic of s splits; 512 = 256 CUs x 2 resident blocks.  A power-of-two
// rule leaves e.g. 144 tiles x 4 = 576 blocks = 2 rounds at 56 % occupancy; 7 splits give 1008 blocks (98 %).
static int tn_choose_splits(int64_t M, int64_t N1, int64_t N2) {
  const int64_t tiles = ((N1 + 127) / 128) * ((N2 + 127) / 128);
  const double row_us = 0.106;                                   // one 32-row K-step per ~3.4 us with 2 blocks/CU
  const double part_us = (double)N1 * N2 * 8.0 / 4.0e6;           // write + read of one partial image at ~4 TB/s
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 64; ++s) {
    const int64_t mps = ((M + s - 1) / s + 31) / 32 * 32;
    if (s > 1 && (mps < 256 || mps * (s - 1) >= M)) continue;      // too short, or a split would be empty
    const int64_t rounds = (tiles * s + 511) / 512;
    const double cost = (double)rounds * (double)(mps + 64) * row_us + (s > 1 ? s * part_us + 4.0 : 0.0);
    if (cost < best_cost * 0.995) { best_cost = cost; best = s; }
  }
  return best;
}

// Large weight gradients take the 256 x 256 LDS-DMA kernel: at least 8 tiles of output and enough rows that a split still
// streams >= 512 of them; its splits fill the CUs once (one workgroup per CU).
static bool tn_takes_p256(int64_t M, int64_t N1, int64_t N2, bool b_sub) {
  return !b_sub && N1 >= 192 && N2 >= 256 && M >= 4096 && ((N1 + 255) / 256) * ((N2 + 255) / 256) >= 8;
}
static int tn_p256_splits(int64_t M, int64_t N1, int64_t N2, int ncu) {
  const int tiles = (int)(((N1 + 255) / 256) * ((N2 + 255) / 256));
  int s = ncu / tiles;
  if (s < 1) s = 1;
  while (s > 1 && M / s < 512) --s;
  return s;
}
constexpr size_t TN_ZERO_B = 1024;               // zero page at the end of the workspace (the DMA kernel's padding source)

// ---- grouped small weight gradients (identity row map, no conv): see gemm_tn_w8_group_kernel
extern "C" size_t acx_gemm_tn_group_workspace_bytes(int32_t nprob, const acx_tn_problem* probs) {
  size_t need = 0;
  for (int i = 0; i < nprob; ++i) {
    const int splits = tn_choose_splits(probs[i].M, probs[i].N1, probs[i].N2);
    if (splits > 1) need += ((size_t)splits * probs[i].N1 * probs[i].N2 * sizeof(float) + 255) / 256 * 256;
  }
  return need;
}

extern "C" int acx_gemm_tn_group(acx_ctx* ctx, int32_t nprob, const acx_tn_problem* probs, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (nprob <= 0) return ACX_OK;
  if (!probs) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_group: null problem table%s");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = 4 * 32 * TN_ROWF * sizeof(float);
  const int dev_slot = (ctx ? ctx->device : 0) & 63;
  static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_w8_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  int i = 0;
  size_t woff = 0;
  while (i < nprob) {
    TnGroup G;
    TnReduceGroup R;
    memset(&G, 0, sizeof(G));
    memset(&R, 0, sizeof(R));
    int k = 0, blocks = 0, rblocks = 0, nr = 0;
    double flops = 0.0;
    for (; i < nprob && k < TN_GROUP_MAX; ++i, ++k) {
      const acx_tn_problem& q = probs[i];
      if (!q.A || !q.B || !q.C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_group: null pointer%s");
      if (q.M <= 0 || q.N1 <= 0 || q.N2 <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_group: empty shape%s");
      if (q.N1 % 4 || q.N2 % 4 || q.lda % 4 || q.ldb % 4 || (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) ||
          (q.b_sub && ((uintptr_t)q.b_sub & 15)))
        return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_group: N1/N2/lda/ldb multiples of 4, 16-byte aligned pointers%s");
      const int splits = tn_choose_splits(q.M, q.N1, q.N2);
      const int tiles = ((q.N1 + 127) / 128) * ((q.N2 + 127) / 128);
      float* dst = (float*)q.C;
      if (splits > 1) {
        const size_t need = ((size_t)splits * q.N1 * q.N2 * sizeof(float) + 255) / 256 * 256;
        if (!workspace || woff + need > workspace_bytes) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_gemm_tn_group: workspace too small%s");
        dst = (float*)((char*)workspace + woff);
        woff += need;
        R.part[nr] = dst; R.out[nr] = (float*)q.C; R.n4[nr] = (long long)q.N1 * q.N2 / 4; R.splits[nr] = splits;
        R.blk0[nr] = rblocks;
        rblocks += (int)((R.n4[nr] + 255) / 256);
        ++nr;
      }
      TnArgs& g = G.p[k];
      g.A = (const float*)q.A; g.B = (const float*)q.B; g.C = dst;
      g.M = q.M; g.N1 = q.N1; g.N2 = q.N2; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.N2;
      g.b_sub = (const float*)q.b_sub; g.conv = 0; g.gn = g.gl = g.cin = 0;
      g.m_per_split = ((q.M + splits - 1) / splits + 31) / 32 * 32;
      g.sh_gl = g.sh_grid = -1;
      G.tiles[k] = tiles;
      G.blk0[k] = blocks;
      blocks += tiles * splits;
      flops += 2.0 * q.M * (double)q.N1 * q.N2;
    }
    G.blk0[k] = blocks; G.n = k;
    R.blk0[nr] = rblocks; R.n = nr;
    AcxProfScope prof__(ctx, ACX_K_GEMM_TN, s);
    if (ctx && ctx->prof_on) { ctx->prof_gemm_flops += flops; ctx->prof_tn_flops += flops; }
    hipLaunchKernelGGL(gemm_tn_w8_group_kernel, dim3((unsigned)blocks), dim3(512), lds, s, G);
    if (nr > 0) hipLaunchKernelGGL(tn_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, R);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_gemm_tn_group");
  return ACX_OK;
}

// ---- weight gradient as an f32-accurate product of bf16 planes: C[N1, N2] = sum_m A[m, n1] B[m, n2], A and B as three bf16 planes
// each (acx_split_bf16x3 of dY and of the layer input), the TN instantiation of the plane-reuse kernel (acx_gemm_x6.h)
// Tile geometry by shape: 256 x 256 (N1, N2 multiples of 256; conv: cin % 256 == 0 -- the UCF / ShanghaiTech heads), 256 x 128
// (N2 a multiple of 128 only, or cin == 128: a 128-column tile lies inside one tap -- c1 of the XD-Violence head: [512, 9 x 128]),
// 128 x 256 with the 1 x 4 wave grid (N1 a multiple of 128 only -- c2 of the XD head: [128, 9 x 512]).  0: not supported.
static int tn_x6_mode(int64_t N1, int64_t N2, int conv, int cin) {
  if (N1 % 256 == 0 && N2 % 256 == 0 && (!conv || cin % 256 == 0)) return 1;
  if (N1 % 256 == 0 && N2 % 128 == 0 && (!conv || cin % 128 == 0)) return 2;
  if (N1 % 128 == 0 && N2 % 256 == 0 && (!conv || cin % 256 == 0)) return 3;
  return 0;
}
static int tn_x6_tiles(int mode, int64_t N1, int64_t N2) {
  return mode == 1 ? (int)((N1 / 256) * (N2 / 256)) : mode == 2 ? (int)((N1 / 256) * (N2 / 128)) : mode == 3 ? (int)((N1 / 128) * (N2 / 256)) : 0;
}
static int tn_x6_splits(int tiles, int64_t M, int64_t N1, int64_t N2, int ncu, size_t workspace_bytes) {
  return x6_choose_split(tiles, (int)((M + 31) / 32), ncu, (size_t)N1 * N2 * sizeof(float), workspace_bytes, 8);
}
extern "C" size_t acx_gemm_tn_x6_workspace_bytes(int32_t M, int32_t N1, int32_t N2) {
  // (the narrowest geometry any mode could pick for this shape: an upper bound on the pieces)
  const int tiles = (int)(((N1 + 255) / 256) * ((N2 + 255) / 256));
  if (tiles <= 0) return 0;
  int s = 512 / tiles;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (size_t)s * N1 * N2 * sizeof(float);
}
extern "C" int acx_gemm_tn_x6(acx_ctx* ctx, const void* A3, int64_t a_plane_stride, int32_t lda, const void* B3, int64_t b_plane_stride,
                              int32_t ldb, float* C, int32_t ldc, int32_t M, int32_t N1, int32_t N2, int32_t conv, int32_t gn, int32_t gl,
                              int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, void* stream) {
  if (!A3 || !B3 || !C || !zero_page) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_x6: null pointer (the zero page is required)%s");
  if (M <= 0 || N1 <= 0 || N2 <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_x6: empty shape%s");
  const int mode = tn_x6_mode(N1, N2, conv, cin);
  if (!mode || lda % 8 || ldb % 8 || ldc != N2 || ((a_plane_stride | b_plane_stride) & 15) ||
      (((uintptr_t)A3 | (uintptr_t)B3 | (uintptr_t)C | (uintptr_t)zero_page) & 15))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm_tn_x6: N1 a multiple of 256 and N2 of 128, or N1 of 128 and N2 of 256 (conv: cin a multiple of the tile width); lda / ldb multiples of 8, dense C, 16-byte aligned planes%s");
  if (conv && (cin <= 0 || N2 != 9 * cin || gn <= 0 || gl <= 0 || (gl & (gl - 1)) || (gn & (gn - 1)) || M % (gn * gl)))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm_tn_x6: conv needs N2 == 9 cin, a power-of-two grid, whole tiles%s");
  int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  if (ctx && ctx->opt_x6_cus > 0 && ctx->opt_x6_cus < ncu) ncu = ctx->opt_x6_cus;   // ACX_OPT_X6_CUS
  const int tiles = tn_x6_tiles(mode, N1, N2);
  const int split = workspace ? tn_x6_splits(tiles, M, N1, N2, ncu, workspace_bytes) : 1;
  Args g;
  memset(&g, 0, sizeof(g));
  g.d.A = A3; g.d.W = B3; g.d.C = C;
  g.d.M = N1; g.d.N = N2; g.d.K = M; g.d.lda = lda; g.d.ldw = ldb; g.d.ldc = ldc;
  g.d.a_dtype = ACX_BF16; g.d.c_dtype = ACX_F32; g.d.prec = ACX_PREC_BF16; g.d.pairs = 6;
  g.d.a_plane_stride = a_plane_stride; g.d.w_plane_stride = b_plane_stride;
  g.d.amap = conv ? ACX_AMAP_CONV3X3 : ACX_AMAP_IDENTITY; g.d.gn = gn; g.d.gl = gl; g.d.cin = cin;
  g.ksplit = split; g.kchunk = ((int)((M + 31) / 32) + split - 1) / split; g.partial = split > 1 ? (float*)workspace : nullptr;
  g.zeros = (const float*)zero_page;
  const int items = tiles * split;
  const dim3 xgrid((unsigned)(items < ncu ? items : ncu));
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_GEMM_TN, s);
  if (ctx && ctx->prof_on) { ctx->prof_gemm_flops += 2.0 * M * (double)N1 * N2; ctx->prof_tn_flops += 2.0 * M * (double)N1 * N2; }
  const int dev_slot = (ctx ? ctx->device : 0) & 63;
#define ACX_TNX6(CV, NI_, W14_)                                                                      \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];                          \
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_x6_p4_kernel<0, 0, 0, CV, 1, 0, NI_, W14_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6_LDS_B); attr_done = true; } \
    hipLaunchKernelGGL((gemm_x6_p4_kernel<0, 0, 0, CV, 1, 0, NI_, W14_>), xgrid, dim3(256), (size_t)X6_LDS_B, s, g); \
  } while (0)
  if (mode == 1) { if (conv) ACX_TNX6(1, 4, 0); else ACX_TNX6(0, 4, 0); }
  else if (mode == 2) { if (conv) ACX_TNX6(1, 2, 0); else ACX_TNX6(0, 2, 0); }
  else { if (conv) ACX_TNX6(1, 2, 1); else ACX_TNX6(0, 2, 1); }
#undef ACX_TNX6
  if (split > 1) {
    const int64_t n4 = (int64_t)N1 * N2 / 4;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, n4, split);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_gemm_tn_x6");
  return ACX_OK;
}

extern "C" size_t acx_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2) {
  const int splits = tn_choose_splits(M, N1, N2);
  size_t need = splits > 1 ? (size_t)splits * N1 * N2 * sizeof(float) : 0;
  if (tn_takes_p256(M, N1, N2, false)) {        // sized for a 256-CU device; fewer CUs mean fewer splits
    const int sp = tn_p256_splits(M, N1, N2, 256);
    const size_t need2 = (sp > 1 ? (size_t)sp * N1 * N2 * sizeof(float) : 0) + TN_ZERO_B;
    if (need2 > need) need = need2;
  }
  return need;
}

extern "C" int acx_gemm_tn(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                           int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                           int32_t cin, void* workspace, size_t workspace_bytes, void* stream) {
  return acx_gemm_tn_zp(ctx, A, lda, B, ldb, C, ldc, M, N1, N2, b_sub, conv, gn, gl, cin, workspace, workspace_bytes, nullptr, stream);
}

// splits_out != nullptr (acx_gemm_tn_parts): no reduce launch -- *splits_out = the number of K-split partial images left in
// `workspace` ([splits][N1][N2] f32, image order = tn_reduce_kernel's summation order); 1: C holds the result itself
static int gemm_tn_impl(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                        int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                        int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, int32_t* splits_out, void* stream);
extern "C" int acx_gemm_tn_zp(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                              int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                              int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, void* stream) {
  return gemm_tn_impl(ctx, A, lda, B, ldb, C, ldc, M, N1, N2, b_sub, conv, gn, gl, cin, workspace, workspace_bytes, zero_page, nullptr, stream);
}
extern "C" int acx_gemm_tn_parts(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                                 int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                                 int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, int32_t* splits_out,
                                 void* stream) {
  if (!splits_out) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn_parts: null splits_out%s");
  return gemm_tn_impl(ctx, A, lda, B, ldb, C, ldc, M, N1, N2, b_sub, conv, gn, gl, cin, workspace, workspace_bytes, zero_page, splits_out, stream);
}
static int gemm_tn_impl(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                        int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                        int32_t cin, void* workspace, size_t workspace_bytes, const void* zero_page, int32_t* splits_out, void* stream) {
  if (!A || !B || !C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: null pointer%s");
  if (M <= 0 || N1 <= 0 || N2 <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: empty shape%s");
  if (N1 % 4 || N2 % 4 || lda % 4 || ldb % 4 || ldc != N2 || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: N1/N2/lda/ldb must be multiples of 4, C dense, 16-byte aligned%s");
  if (conv && (cin <= 0 || N2 != 9 * cin || cin % 4 || gn <= 0 || gl <= 0 || M % (gn * gl)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: bad conv geometry%s");
  if (b_sub && ((uintptr_t)b_sub & 15)) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: b_sub alignment%s");
  if (tn_takes_p256(M, N1, N2, b_sub != nullptr) && workspace && M >= (ctx ? ctx->opt_tn_p256_min_rows : 4096)) {
    const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
    int splits = tn_p256_splits(M, N1, N2, ncu > 256 ? 256 : ncu);
    const size_t part = (size_t)N1 * N2 * sizeof(float);
    const size_t ztail = zero_page ? 0 : TN_ZERO_B;      // a caller-owned zero page (>= 1 KB, never written) saves the clear launch
    while (splits > 1 && (size_t)splits * part + ztail > workspace_bytes) --splits;
    const size_t zoff = splits > 1 ? (size_t)splits * part : 0;
    if (zoff + ztail <= workspace_bytes && !(zoff & 15) && !((uintptr_t)zero_page & 15)) {
      hipStream_t s = (hipStream_t)stream;
      AcxProfScope prof__(ctx, ACX_K_GEMM_TN, s);
      if (ctx && ctx->prof_on) { ctx->prof_gemm_flops += 2.0 * M * (double)N1 * N2; ctx->prof_tn_flops += 2.0 * M * (double)N1 * N2; }
      const float* zeros = zero_page ? (const float*)zero_page : (const float*)((char*)workspace + zoff);
      // a KERNEL clears the page: as a hipMemsetAsync node inside a captured graph the clear was observed to run unordered
      // with the consumer (stale workspace bytes read as padding, run-to-run different gradients under graph replay)
      if (!zero_page)
        hipLaunchKernelGGL(tn_zero_page_kernel, dim3(1), dim3(TN_ZERO_B / 16), 0, s, reinterpret_cast<float4*>(const_cast<float*>(zeros)));
      TnArgs g;
      g.A = A; g.B = B; g.C = splits > 1 ? (float*)workspace : C;
      g.M = M; g.N1 = N1; g.N2 = N2; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
      g.b_sub = nullptr; g.conv = conv; g.gn = gn; g.gl = gl; g.cin = cin;
      g.m_per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
      g.sh_gl = g.sh_grid = -1;
      if (conv && !(gl & (gl - 1)) && !((gn * gl) & (gn * gl - 1))) {
        g.sh_gl = __builtin_ctz((unsigned)gl);
        g.sh_grid = __builtin_ctz((unsigned)(gn * gl));
      }
      const int dev_slot = (ctx ? ctx->device : 0) & 63;
      static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];
      if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_p256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TP_LDS_B);
        attr_done = true;
      }
      const int tiles = ((N1 + 255) / 256) * ((N2 + 255) / 256);
      hipLaunchKernelGGL(gemm_tn_p256_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(1024), (size_t)TP_LDS_B, s, g, zeros);
      if (splits_out) *splits_out = splits;
      if (splits > 1 && !splits_out) {
        const int64_t n4 = (int64_t)N1 * N2 / 4;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, n4,
                           splits);
      }
      ACX_CHECK_LAUNCH(ctx, "acx_gemm_tn");
      return ACX_OK;
    }
  }
  const int tiles = ((N1 + 127) / 128) * ((N2 + 127) / 128);
  const int splits = tn_choose_splits(M, N1, N2);
  const size_t need = splits > 1 ? (size_t)splits * N1 * N2 * sizeof(float) : 0;
  if (need > workspace_bytes || (need && !workspace)) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_gemm_tn: workspace too small%s");
  TnArgs g;
  g.A = A; g.B = B; g.C = splits > 1 ? (float*)workspace : C;
  g.M = M; g.N1 = N1; g.N2 = N2; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.b_sub = b_sub; g.conv = conv; g.gn = gn; g.gl = gl; g.cin = cin;
  g.m_per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
  g.sh_gl = g.sh_grid = -1;
  if (conv && !(gl & (gl - 1)) && !((gn * gl) & (gn * gl - 1))) {
    g.sh_gl = __builtin_ctz((unsigned)gl);
    g.sh_grid = __builtin_ctz((unsigned)(gn * gl));
  }
  const size_t lds = 4 * 32 * TN_ROWF * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_GEMM_TN, s);
  if (ctx && ctx->prof_on) { ctx->prof_gemm_flops += 2.0 * M * (double)N1 * N2; ctx->prof_tn_flops += 2.0 * M * (double)N1 * N2; }
  const int dev_slot = (ctx ? ctx->device : 0) & 63;            // kernel attributes are per device
  static bool attr_dev_[64] = {}; bool& attr_done = attr_dev_[dev_slot];
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const bool tn_w8 = ACX_DBG_SWITCH("TN_W8", true);
  if (tn_w8) {
    static bool attr8_dev_[64] = {}; bool& attr8_done = attr8_dev_[dev_slot];
    if (!attr8_done) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr8_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_w8_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(512), lds, s, g);
  } else
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(NTHREADS), lds, s, g);
  if (splits_out) *splits_out = splits;
  if (splits > 1 && !splits_out) {
    const int64_t n4 = (int64_t)N1 * N2 / 4;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, n4,
                       splits);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_gemm_tn");
  return ACX_OK;
}

