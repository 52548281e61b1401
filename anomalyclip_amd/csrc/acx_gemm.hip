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
#include "acx_internal.h"

#include <stdlib.h>
#include <type_traits>

#ifndef ACX_TRACE
#define ACX_TRACE 0
#endif

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;                // LDS row stride in bytes (128 data + 16 pad)
constexpr int TILE_B = BM * ROWB;        // 18432
constexpr int NTHREADS = 256;

struct Args {
  acx_gemm_desc d;
  int tiles_n;
  long long* trace;        // debug timeline buffer (ACX_TRACE builds only)
  int ksplit, kchunk;      // split-K (FAST path, skinny problems): gridDim.y splits of kchunk K-steps each
  float* partial;          // [ksplit][M][N] raw partial sums (epilogue applied by splitk_reduce_kernel)
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
        if (act == ACX_ACT_QUICKGELU) v = v * (1.f / (1.f + __expf(-1.702f * v)));
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


// =====================================================================================================
// f32 GEMM, 8 waves per 128x128 tile (FAST problems only: identity rows, K % 32 == 0, f32 in / out)
//
// Same LDS image, K-permutation and K-step schedule as gemm_kernel, but the tile is shared by 8 waves of 64 x 32
// instead of 4 waves of 64 x 64: two co-resident blocks then put FOUR waves on every SIMD (<= 128 VGPRs each), so
// a wave that sits at the K-step barrier or on an LDS round trip leaves three others to feed the matrix pipe
// instead of one.  Costs 1.5x the LDS fragment reads per flop (still < 10 % of the LDS port).
// CONV = 1: the implicit-GEMM 3x3 convolution of the axial feed-forwards (A rows are token rows shifted by the tap of
// the current K-step, zero outside the (gn, gl) grid; cin % 32 == 0 so a K-step never straddles two taps).
template <int ACT, int RES, int CONV>
__global__ __launch_bounds__(512, 2) void gemm_f32_w8_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tiles_n, tn = wg % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int chunk = t & 7, rbase = t >> 3;                  // 64 staged rows per pass, 2 passes per operand
  const float* pa0 = (const float*)d.A + (size_t)min(m0 + rbase, d.M - 1) * d.lda + chunk * 4;
  const float* pa1 = (const float*)d.A + (size_t)min(m0 + rbase + 64, d.M - 1) * d.lda + chunk * 4;
  const float* pw0 = (const float*)d.W + (size_t)min(n0 + rbase, d.N - 1) * d.ldw + chunk * 4;
  const float* pw1 = (const float*)d.W + (size_t)min(n0 + rbase + 64, d.N - 1) * d.ldw + chunk * 4;
  float4 ra0, ra1, rw0, rw1;
  // conv: grid coordinates of the two staged token rows, base row of their tile, validity of the staged taps
  int cn0 = 0, cl0 = 0, cn1 = 0, cl1 = 0;
  long cb0 = 0, cb1 = 0;
  bool ok0 = true, ok1 = true;
  if constexpr (CONV != 0) {
    const int gsz = d.gn * d.gl;
    const int r0_ = min(m0 + rbase, d.M - 1), r1_ = min(m0 + rbase + 64, d.M - 1);
    const int t0_ = r0_ / gsz, t1_ = r1_ / gsz;
    cb0 = (long)t0_ * gsz; cb1 = (long)t1_ * gsz;
    cn0 = (r0_ - t0_ * gsz) / d.gl; cl0 = (r0_ - t0_ * gsz) - cn0 * d.gl;
    cn1 = (r1_ - t1_ * gsz) / d.gl; cl1 = (r1_ - t1_ * gsz) - cn1 * d.gl;
  }
#define W8_LOAD(k0)                                                                                \
  do {                                                                                             \
    if constexpr (CONV != 0) {                                                                     \
      const int tap_ = (k0) / d.cin;                    /* uniform */                             \
      const int dn_ = tap_ / 3 - 1, dl_ = tap_ - (tap_ / 3) * 3 - 1, kc_ = (k0) - tap_ * d.cin + chunk * 4; \
      const int n0_ = cn0 + dn_, l0_ = cl0 + dl_, n1_ = cn1 + dn_, l1_ = cl1 + dl_;                 \
      ok0 = (unsigned)n0_ < (unsigned)d.gn && (unsigned)l0_ < (unsigned)d.gl;                      \
      ok1 = (unsigned)n1_ < (unsigned)d.gn && (unsigned)l1_ < (unsigned)d.gl;                      \
      const long s0_ = cb0 + (long)min(max(n0_, 0), d.gn - 1) * d.gl + min(max(l0_, 0), d.gl - 1); \
      const long s1_ = cb1 + (long)min(max(n1_, 0), d.gn - 1) * d.gl + min(max(l1_, 0), d.gl - 1); \
      ra0 = ld4((const float*)d.A + (size_t)s0_ * d.lda + kc_);                                    \
      ra1 = ld4((const float*)d.A + (size_t)s1_ * d.lda + kc_);                                    \
    } else {                                                                                       \
      ra0 = ld4(pa0 + (k0)); ra1 = ld4(pa1 + (k0));                                                \
    }                                                                                              \
    rw0 = ld4(pw0 + (k0)); rw1 = ld4(pw1 + (k0));                                                  \
  } while (0)
#define W8_STORE(stage, r)                                                             \
  do {                                                                                 \
    char* sA_ = smem + (stage) * 2 * TILE_B;                                           \
    const int off_ = (rbase + 64 * (r)) * ROWB + chunk * 16;                           \
    float4 va_ = ra##r;                                                                \
    if constexpr (CONV != 0) { if (!ok##r) va_ = make_float4(0.f, 0.f, 0.f, 0.f); }    \
    *reinterpret_cast<float4*>(sA_ + off_) = va_;                                      \
    *reinterpret_cast<float4*>(sA_ + TILE_B + off_) = rw##r;                           \
  } while (0)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, hh = lane >> 5;
  const int a_off = (wm * 64 + li) * ROWB + hh * 16;
  const int w_off = (wn * 32 + li) * ROWB + hh * 16;
  float4 xa0, xa1, xb, ya0, ya1, yb;
#define W8_RD(SET, stage, q)                                                                              \
  do {                                                                                                    \
    const char* sA_ = smem + (stage) * 2 * TILE_B;                                                        \
    SET##a0 = *reinterpret_cast<const float4*>(sA_ + a_off + (q) * 32);                                   \
    SET##a1 = *reinterpret_cast<const float4*>(sA_ + a_off + 32 * ROWB + (q) * 32);                       \
    SET##b = *reinterpret_cast<const float4*>(sA_ + TILE_B + w_off + (q) * 32);                           \
  } while (0)
#define W8_MM(SET)                                                                                        \
  do {                                                                                                    \
    const float a0_[4] = {SET##a0.x, SET##a0.y, SET##a0.z, SET##a0.w};                                    \
    const float a1_[4] = {SET##a1.x, SET##a1.y, SET##a1.z, SET##a1.w};                                    \
    const float b_[4] = {SET##b.x, SET##b.y, SET##b.z, SET##b.w};                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[j], b_[j], acc[0], 0, 0, 0);                      \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[j], b_[j], acc[1], 0, 0, 0);                      \
    }                                                                                                     \
  } while (0)

  const int nk = d.K / 32;
  W8_LOAD(0);
  W8_STORE(0, 0); W8_STORE(0, 1);
  __syncthreads();
  if (nk > 1) W8_LOAD(32);
  W8_RD(x, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nk;
    W8_RD(y, cur, 1);
    W8_MM(x);                                    // phase 0
    W8_RD(x, cur, 2);
    if (more) W8_STORE(nxt, 0);
    W8_MM(y);                                    // phase 1
    W8_RD(y, cur, 3);
    if (more) W8_STORE(nxt, 1);
    W8_MM(x);                                    // phase 2
    __syncthreads();                             // next tile complete in LDS; everyone holds its phase-3 fragments
    if (kt + 2 < nk) W8_LOAD((kt + 2) * 32);
    if (more) W8_RD(x, nxt, 0);
    W8_MM(y);                                    // phase 3
  }
#undef W8_MM
#undef W8_RD
#undef W8_STORE
#undef W8_LOAD

  // ---- epilogue (structure of gemm_kernel's: everything computed, then stored)
  const int col = n0 + wn * 32 + li;
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
    if constexpr (RES != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(rowb + (r & 3) + 8 * (r >> 2), d.M - 1);
        outv[r] = d.residual[(size_t)row * d.ldr + colc];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[mi][r] + bias;
      if constexpr (ACT == ACX_ACT_QUICKGELU) v = v * (1.f / (1.f + __expf(-1.702f * v)));
      if constexpr (ACT == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;
      outv[r] += v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(outv[r]));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowb + (r & 3) + 8 * (r >> 2);
      if (cok && row < d.M) ((float*)d.C)[(size_t)row * d.ldc + col] = outv[r];
    }
  }
}

// =====================================================================================================
// bf16 GEMM with LDS-DMA staging:  C = epilogue(A[M,K] W[N,K]^T), A and W bf16 in global, K % 64 == 0.
//
// At the bf16 MFMA rate a 128x128x64 K-step is 16 x v_mfma_f32_32x32x16_bf16 = 512 matrix-pipe cycles per wave,
// an eighth of the f32 kernel's -- the register-staged loop above then spends its time in ds_write_b128 (79 B/clk)
// and staging VGPR traffic.  Here both operands go global -> LDS directly (global_load_lds_dwordx4: no staging
// registers, no LDS write instructions): a wave instruction moves 8 rows x 128 B = 1 KB to wave-uniform
// LDS base + lane*16, so the LDS image is row-linear [row][128 B] and the bank swizzle is applied on the SOURCE
// side: LDS position p of row r holds global 16-B chunk p ^ ((r >> 1) & 7); the reader of chunk c of row r
// looks at position c ^ ((r >> 1) & 7) -- conflict-free for every ds_read_b128 lane group, and the 8 lanes of
// a row still cover one full 128-B line.  Two 32 KB stages per block, two blocks per CU; per K-step: issue the
// DMA of the next tile, wait for this tile's (counted vmcnt, never 0 inside the loop), raw s_barrier, 16
// ds_read_b128 + 16 MFMA, raw s_barrier (the stage is overwritten by the DMA issued in the next iteration).
constexpr int DMA_ROWB = 128;                    // LDS row = one K-step of bf16
constexpr int DMA_OP_B = 128 * DMA_ROWB;         // one operand tile: 16 KB
constexpr int DMA_STAGE_B = 2 * DMA_OP_B;        // A | W

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int C_BF16, int ACT, int RES>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_dma_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tiles_n, tn = wg % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, hh = lane >> 5;

  // ---- DMA sources: instruction j (0..3) of wave w moves tile rows (4w + j)*8 .. +7; lane -> (row, position)
  const char* ga0; const char* ga1; const char* ga2; const char* ga3;
  const char* gw0; const char* gw1; const char* gw2; const char* gw3;
#define DMA_SRC(j)                                                                                 \
  do {                                                                                             \
    const int row_ = (4 * wave + (j)) * 8 + (lane >> 3);                                           \
    const int c_ = (lane & 7) ^ ((row_ >> 1) & 7);                                                 \
    ga##j = (const char*)d.A + ((size_t)min(m0 + row_, d.M - 1) * d.lda) * 2 + c_ * 16;            \
    gw##j = (const char*)d.W + ((size_t)min(n0 + row_, d.N - 1) * d.ldw) * 2 + c_ * 16;            \
  } while (0)
  DMA_SRC(0); DMA_SRC(1); DMA_SRC(2); DMA_SRC(3);
#undef DMA_SRC
  const int dma_off = 4 * wave * 1024;             // this wave's 4 KB slice of each operand image
#define DMA_ISSUE(stage, kt_)                                                                      \
  do {                                                                                             \
    char* sA_ = smem + (stage) * DMA_STAGE_B + dma_off;                                            \
    char* sW_ = sA_ + DMA_OP_B;                                                                    \
    const size_t ko_ = (size_t)(kt_) * 128;                                                        \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga0 + ko_), (lds_void_t*)(sA_), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga1 + ko_), (lds_void_t*)(sA_ + 1024), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga2 + ko_), (lds_void_t*)(sA_ + 2048), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga3 + ko_), (lds_void_t*)(sA_ + 3072), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw0 + ko_), (lds_void_t*)(sW_), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw1 + ko_), (lds_void_t*)(sW_ + 1024), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw2 + ko_), (lds_void_t*)(sW_ + 2048), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw3 + ko_), (lds_void_t*)(sW_ + 3072), 16, 0, 0); \
  } while (0)

  // ---- fragment addresses: row (wm*64 + mi*32 + li), chunk 2*kk + hh at position chunk ^ ((li >> 1) & 7)
  const int sw = (li >> 1) & 7;
  const int a_base = (wm * 64 + li) * DMA_ROWB;
  const int w_base = DMA_OP_B + (wn * 64 + li) * DMA_ROWB;
  const int o0 = ((0 + hh) ^ sw) * 16, o1 = ((2 + hh) ^ sw) * 16, o2 = ((4 + hh) ^ sw) * 16, o3 = ((6 + hh) ^ sw) * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = d.K / 64;
  DMA_ISSUE(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      DMA_ISSUE(cur ^ 1, kt + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's 8 DMAs of tile kt have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                            // ... and everybody else's
    const char* sS = smem + cur * DMA_STAGE_B;
#define DMA_FRAG(base, o) (*reinterpret_cast<const bf16x8*>(sS + (base) + (o)))
#define DMA_RD(S, o)                                                                               \
  do {                                                                                             \
    fA0##S = DMA_FRAG(a_base, o); fA1##S = DMA_FRAG(a_base + 32 * DMA_ROWB, o);                    \
    fB0##S = DMA_FRAG(w_base, o); fB1##S = DMA_FRAG(w_base + 32 * DMA_ROWB, o);                    \
  } while (0)
#define DMA_MM(S)                                                                                  \
  do {                                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA0##S, fB0##S, acc[0][0], 0, 0, 0);       \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA0##S, fB1##S, acc[0][1], 0, 0, 0);       \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA1##S, fB0##S, acc[1][0], 0, 0, 0);       \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA1##S, fB1##S, acc[1][1], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
    // fragment reads run two k-substeps ahead of their MFMAs (4 named register sets): the matrix pipe does not
    // wait on an LDS round trip between the four k-substeps
    bf16x8 fA0P, fA1P, fB0P, fB1P, fA0Q, fA1Q, fB0Q, fB1Q, fA0R, fA1R, fB0R, fB1R, fA0S, fA1S, fB0S, fB1S;
    DMA_RD(P, o0); DMA_RD(Q, o1);
    __builtin_amdgcn_sched_barrier(0);
    DMA_RD(R, o2); DMA_MM(P);
    DMA_RD(S, o3); DMA_MM(Q);
    DMA_MM(R); DMA_MM(S);
#undef DMA_MM
#undef DMA_RD
#undef DMA_FRAG
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // stage `cur` is free for the DMA of tile kt + 2
  }
#undef DMA_ISSUE

  // ---- epilogue (same structure as gemm_kernel's: compute everything, then store)
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
      if constexpr (RES != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rowb + (r & 3) + 8 * (r >> 2), d.M - 1);
          outv[r] = d.residual[(size_t)row * d.ldr + colc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mi][ni][r] + bias;
        if constexpr (ACT == ACX_ACT_QUICKGELU) v = v * (1.f / (1.f + __expf(-1.702f * v)));
        outv[r] += v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(outv[r]));
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
}

// =====================================================================================================
// bf16 GEMM, 256x256 block tile, persistent, LDS-DMA double buffer  (the large-M ViT shapes in bf16 mode)
//
// One 512-thread block per CU walks its share of the output tiles.  K-step = 64 bf16 = one full 128-B line per
// row (a 64-B half line per step re-fetches every line twice: measured 5.1 TB/s of fill traffic, DMA-bound).
// A stage = A[256][128 B] | W[256][128 B] = 64 KB, two stages.  8 waves as 2(M) x 4(N); a wave owns 128 x 64 of
// C = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 and issues 32 MFMAs per K-step (1024 pipe cycles, 2048 per
// SIMD: about one L2 round trip, which is the lead the DMA of the next step gets).  Per K-step:
//     s_barrier                           -> DMA(s) complete for every wave, stage of step s-1 free
//     issue DMA(s+1)                      -> 8 x global_load_lds_dwordx4 per wave; s+1 may belong to the NEXT tile,
//                                            so its first lines land while this tile's epilogue runs
//     4 x { 6 ds_read_b128 || 8 MFMA }    -> fragment sets alternate between two named register sets
//     s_waitcnt vmcnt(0)                  -> this wave's DMA(s+1) landed (before any epilogue store is issued)
//     [last K-step of a tile: epilogue through 8 x 4 KB of wave-private LDS -> full-line 16-byte stores]
// Source-side swizzle as in gemm_bf16_dma_kernel (position p of row r <- chunk p ^ ((r >> 1) & 7)).
constexpr int RG_TM = 256, RG_TN = 256;
constexpr int RG_ROWB = 128;
constexpr int RG_OP_B = 256 * RG_ROWB;      // 32 KB per operand image
constexpr int RG_STAGE_B = 2 * RG_OP_B;     // 64 KB

#define RG_GLDS(a, la) __builtin_amdgcn_global_load_lds((gbl_void_t*)(a), (lds_void_t*)(la), 16, 0, 0)

template <int C_BF16, int ACT, int RES>
__global__ __launch_bounds__(512, 1) void gemm_bf16_ring_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, hh = lane >> 5;
  const int tiles_n = (d.N + RG_TN - 1) / RG_TN, tiles_m = (d.M + RG_TM - 1) / RG_TM;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  // XCD-aware start: within a round of G tiles each XCD (blockIdx & 7) owns a contiguous chunk
  const int xcd = blockIdx.x & 7, qq = G >> 3, rr = G & 7;
  const int b0 = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + ((int)blockIdx.x >> 3);
  const int my_tiles = b0 < ntiles ? (ntiles - b0 + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int nk = d.K / 64;
  const int S = my_tiles * nk;

  // ---- DMA cursor: (tile jd, K-step hd) of the next stage to issue, source pointers of tile jd at k = 0.
  // instruction i (0..3) of wave w moves tile rows (4w + i)*8 .. +7 of each operand; lane -> (row, position)
  const int dr = (4 * wave) * 8 + (lane >> 3);
  const char *pa0, *pa1, *pa2, *pa3, *pw0, *pw1, *pw2, *pw3;
  int jd = 0, hd = 0, std_ = 0;
#define RG_SRC1(i, m0_, n0_)                                                                       \
  do {                                                                                             \
    const int row_ = dr + 8 * (i);                                                                 \
    const int c_ = (lane & 7) ^ ((row_ >> 1) & 7);                                                 \
    pa##i = (const char*)d.A + ((size_t)min((m0_) + row_, d.M - 1) * d.lda) * 2 + c_ * 16;         \
    pw##i = (const char*)d.W + ((size_t)min((n0_) + row_, d.N - 1) * d.ldw) * 2 + c_ * 16;         \
  } while (0)
#define RG_SET_SRC(j)                                                                              \
  do {                                                                                             \
    const int L_ = b0 + min((j), my_tiles - 1) * G;    /* past the end: re-read the last tile (never consumed) */ \
    const int tm_ = L_ / tiles_n, tn_ = L_ - tm_ * tiles_n;                                        \
    RG_SRC1(0, tm_ * RG_TM, tn_ * RG_TN); RG_SRC1(1, tm_ * RG_TM, tn_ * RG_TN);                    \
    RG_SRC1(2, tm_ * RG_TM, tn_ * RG_TN); RG_SRC1(3, tm_ * RG_TM, tn_ * RG_TN);                    \
  } while (0)
#define RG_DMA()                                                                                   \
  do {                                                                                             \
    char* sA_ = smem + std_ * RG_STAGE_B + (4 * wave) * 1024;                                      \
    char* sW_ = sA_ + RG_OP_B;                                                                     \
    const size_t ko_ = (size_t)hd * 128;                                                           \
    RG_GLDS(pa0 + ko_, sA_);        RG_GLDS(pa1 + ko_, sA_ + 1024);                          \
    RG_GLDS(pa2 + ko_, sA_ + 2048); RG_GLDS(pa3 + ko_, sA_ + 3072);                          \
    RG_GLDS(pw0 + ko_, sW_);        RG_GLDS(pw1 + ko_, sW_ + 1024);                          \
    RG_GLDS(pw2 + ko_, sW_ + 2048); RG_GLDS(pw3 + ko_, sW_ + 3072);                          \
    std_ ^= 1;                                                                                     \
    if (++hd == nk) { hd = 0; ++jd; RG_SET_SRC(jd); }                                              \
  } while (0)
  RG_SET_SRC(0);

  // ---- fragment addresses inside a stage: row, chunk 2*kk + hh at position chunk ^ ((li >> 1) & 7)
  const int sw = (li >> 1) & 7;
  const int fa = (wm * 128 + li) * RG_ROWB;                 // + mi * 32 * RG_ROWB
  const int fw = RG_OP_B + (wn * 64 + li) * RG_ROWB;        // + ni * 32 * RG_ROWB
  const int oK0 = ((0 + hh) ^ sw) * 16, oK1 = ((2 + hh) ^ sw) * 16, oK2 = ((4 + hh) ^ sw) * 16, oK3 = ((6 + hh) ^ sw) * 16;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  bf16x8 pA0, pA1, pA2, pA3, pB0, pB1, qA0, qA1, qA2, qA3, qB0, qB1;
#define RG_FRAG(stage, base, o) (*reinterpret_cast<const bf16x8*>(smem + (stage) * RG_STAGE_B + (base) + (o)))
#define RG_RD(S, stage, o)                                                                         \
  do {                                                                                             \
    S##A0 = RG_FRAG(stage, fa, o);                    S##A1 = RG_FRAG(stage, fa + 32 * RG_ROWB, o); \
    S##A2 = RG_FRAG(stage, fa + 64 * RG_ROWB, o);     S##A3 = RG_FRAG(stage, fa + 96 * RG_ROWB, o); \
    S##B0 = RG_FRAG(stage, fw, o);                    S##B1 = RG_FRAG(stage, fw + 32 * RG_ROWB, o); \
    __builtin_amdgcn_sched_barrier(0);   /* keep the reads AHEAD of the MFMAs they overlap with */   \
  } while (0)
// Operands are SWAPPED (W fragment first): the accumulator then holds C^T, i.e. a lane owns ONE row of C and 4 x 4
// consecutive columns per 32x32 tile, so the epilogue stores (and the residual loads) are 16-byte accesses -- a
// quarter of the store instructions of the natural layout (narrow stores cost up to 46 % of this kernel).
// The first MFMA of a fragment set is issued BEFORE the next set's reads: hipcc waits lgkmcnt(0) (never a partial
// count) in front of the first use of a set, and at that point only reads issued >= 7 MFMAs earlier are outstanding.
#define RG_MM_HEAD(S)                                                                              \
  do {                                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B0, S##A0, acc[0][0], 0, 0, 0);         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#define RG_MM_TAIL(S)                                                                              \
  do {                                                                                             \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B1, S##A0, acc[0][1], 0, 0, 0);         \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B0, S##A1, acc[1][0], 0, 0, 0);         \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B1, S##A1, acc[1][1], 0, 0, 0);         \
    acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B0, S##A2, acc[2][0], 0, 0, 0);         \
    acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B1, S##A2, acc[2][1], 0, 0, 0);         \
    acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B0, S##A3, acc[3][0], 0, 0, 0);         \
    acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(S##B1, S##A3, acc[3][1], 0, 0, 0);         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)

  RG_DMA();                         // DMA(0)
  int cur = 0, h = 0, j = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int s = 0; s < S; ++s) {
    __builtin_amdgcn_s_barrier();                            // everybody's DMA(s) is complete; the other stage is free
    RG_DMA();                                                // DMA(s + 1)
    __builtin_amdgcn_sched_barrier(0);
    RG_RD(p, cur, oK0);
    RG_MM_HEAD(p); RG_RD(q, cur, oK1); RG_MM_TAIL(p);
    RG_MM_HEAD(q); RG_RD(p, cur, oK2); RG_MM_TAIL(q);
    RG_MM_HEAD(p); RG_RD(q, cur, oK3); RG_MM_TAIL(p);
    RG_MM_HEAD(q); RG_MM_TAIL(q);
    // this wave's DMA(s+1) has had the whole K-step to land.  Waiting for it HERE, before the epilogue's stores are
    // issued, keeps those stores out of the wait (vmcnt cannot tell loads from stores): they get the next K-step to
    // drain instead of stalling it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (++h == nk) {
      // ---- epilogue of tile j (rows >= M / cols >= N masked), then restart the accumulators
      const int L = b0 + j * G;
      const int tm = L / tiles_n, tn = L - tm * tiles_n;
      const int m0 = tm * RG_TM, n0 = tn * RG_TN;
      // Each 32x32 accumulator tile goes through this wave's private 4 KB of LDS (XOR-swizzled 128-B rows) and
      // comes back row-major: lane l then owns 4 consecutive columns (l & 7) of row (l >> 3) + 8*pass, and one
      // store instruction writes 8 complete 128-B lines (narrow or per-row-scattered stores cost 30-46 % here).
      // No global load may sit between stores and its use (vmcnt cannot tell them apart and the wait would cover
      // the stores' full round trip): both bias vectors are fetched first, and the residual of two 32 x 32 tiles
      // (8 float4 per lane) is requested in one batch before those tiles' stores.
      char* scr = smem + 2 * RG_STAGE_B + wave * 4096;
      const int rl = lane >> 3, cj = lane & 7;
      const int col0 = n0 + wn * 64 + 4 * cj, col1 = col0 + 32;           // 4 consecutive columns (N % 4 == 0)
      float4 bb0 = make_float4(0.f, 0.f, 0.f, 0.f), bb1 = bb0;
      if (d.bias) {
        bb0 = *reinterpret_cast<const float4*>(d.bias + (col0 < d.N ? col0 : 0));
        bb1 = *reinterpret_cast<const float4*>(d.bias + (col1 < d.N ? col1 : 0));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = ni ? col1 : col0;
        const bool cok = col < d.N;
        const float4 b4 = ni ? bb1 : bb0;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
        float4 res[2][4];
        if constexpr (RES != 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
              res[u][ps] = *reinterpret_cast<const float4*>(
                  d.residual + (size_t)min(m0 + wm * 128 + (2 * mp + u) * 32 + rl + 8 * ps, d.M - 1) * d.ldr + (cok ? col : 0));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int mi = 2 * mp + u;
          const int row0 = m0 + wm * 128 + mi * 32 + rl;
#pragma unroll
          for (int k = 0; k < 4; ++k)       // accumulator (C^T layout): row li, columns 8k + 4hh .. +3 = chunk 2k + hh
            *reinterpret_cast<float4*>(scr + li * 128 + (((2 * k + hh) ^ (li & 7)) * 16)) =
                make_float4(acc[mi][ni][4 * k], acc[mi][ni][4 * k + 1], acc[mi][ni][4 * k + 2], acc[mi][ni][4 * k + 3]);
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int rr_ = rl + 8 * ps;
            float4 v = *reinterpret_cast<const float4*>(scr + rr_ * 128 + ((cj ^ (rr_ & 7)) * 16));
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            if constexpr (ACT == ACX_ACT_QUICKGELU) {
              v.x = v.x * (1.f / (1.f + __expf(-1.702f * v.x))); v.y = v.y * (1.f / (1.f + __expf(-1.702f * v.y)));
              v.z = v.z * (1.f / (1.f + __expf(-1.702f * v.z))); v.w = v.w * (1.f / (1.f + __expf(-1.702f * v.w)));
            }
            if constexpr (RES != 0) { v.x += res[u][ps].x; v.y += res[u][ps].y; v.z += res[u][ps].z; v.w += res[u][ps].w; }
            const int row = row0 + 8 * ps;
            if (cok && row < d.M) {
              if constexpr (C_BF16) {
                uint2 pk;
                pk.x = f2bf2(v.x, v.y);
                pk.y = f2bf2(v.z, v.w);
                u32x2* dst = reinterpret_cast<u32x2*>((u16*)d.C + (size_t)row * d.ldc + col);
                if constexpr (RES != 0) *dst = *reinterpret_cast<const u32x2*>(&pk);     // residual stream: re-read soon
                else __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pk), dst);   // streamed: keep A / W in L2
              } else {
                f32x4* dst = reinterpret_cast<f32x4*>((float*)d.C + (size_t)row * d.ldc + col);
                if constexpr (RES != 0) *dst = *reinterpret_cast<const f32x4*>(&v);
                else __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&v), dst);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        }
        }
      }
      h = 0; ++j;
    }
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may still be writing this block's LDS at exit
#undef RG_MM_HEAD
#undef RG_MM_TAIL
#undef RG_RD
#undef RG_FRAG
#undef RG_DMA
#undef RG_SET_SRC
#undef RG_SRC1
}

// =====================================================================================================
// acx_gemm_tn -- weight-gradient GEMM:  C[N1,N2] = sum_m A[m,n1] * bmap(B)[m,n2]      (exact f32 MFMA)
//
// Both operands are stored with the REDUCTION index m as the slow (row) index -- dY [M,N1] and
// X [M,N2] exactly as the forward pass left them -- so no transposes are materialised.  A K-step is
// 32 rows of m; the LDS image is [32 m][128 n] (+pad), a lane of the 32x32x2 MFMA reads its operand
// with ds_read_b32 (32 consecutive floats per half-wave: conflict free for any row stride).
// bmap: identity, or the 3x3-conv gather (column k = tap*cin + ci reads row shift_tap(m), zero outside
// the (gn,gl) grid) for the conv weight gradient, optionally minus a per-column vector (b_sub) for the
// selector's direction gradient.  The M reduction is split over gridDim.y; partial tiles go to a
// workspace and are summed in fixed order by tn_reduce_kernel (deterministic, no atomics).
constexpr int TN_ROWF = 132;   // floats per LDS row (128 + 4 pad keeps 16-B alignment of the b128 writes)

struct TnArgs {
  const float* A; const float* B; float* C;   // C: [splits][N1][ldc] partials (or the result when splits == 1)
  int M, N1, N2, lda, ldb, ldc;
  const float* b_sub;
  int conv, gn, gl, cin;
  int m_per_split;
  int sh_gl, sh_grid;      // log2(gl), log2(gn*gl) when both are powers of two, else -1
};

__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(const TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);           // [stage][A|B][32][TN_ROWF]
  constexpr int TILE_F = 32 * TN_ROWF;
  const int tiles_n2 = (g.N2 + 127) / 128;
  const int tm = blockIdx.x / tiles_n2, tn = blockIdx.x % tiles_n2;
  const int n1_0 = tm * 128, n2_0 = tn * 128;
  const int split = blockIdx.y;
  const int m_begin = split * g.m_per_split;
  const int m_end = min(g.M, m_begin + g.m_per_split);

  const int t = threadIdx.x;
  const int c16 = t & 31, r0 = t >> 5;                  // chunk column (4 floats), base row (0..7) + 8*r
  // column validity / conv tap of this thread's chunk (fixed over the K loop)
  const int ca = n1_0 + 4 * c16, cb = n2_0 + 4 * c16;
  const bool a_cok = ca < g.N1, b_cok = cb < g.N2;      // N1, N2 multiples of 4 (checked on the host)
  int tap_dn = 0, tap_dl = 0, b_col = cb;
  if (g.conv && b_cok) {
    const int tap = cb / g.cin;
    tap_dn = tap / 3 - 1;
    tap_dl = tap - (tap / 3) * 3 - 1;
    b_col = cb - tap * g.cin;
  }
  float4 bsub = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.b_sub && b_cok) bsub = *reinterpret_cast<const float4*>(g.b_sub + cb);
  const int grid_sz = g.conv ? g.gn * g.gl : 1;

  // TN_LOAD_ROW only ISSUES the two 16-byte loads of staging row r and records their predicates; masking and
  // the b_sub recentre happen in TN_STORE_ROW, a whole K-step later, so nothing waits on memory in between.
  float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
  unsigned ok_a = 0, ok_b = 0;
#define TN_LOAD_ROW(r, mbase)                                                                      \
  do {                                                                                             \
    const int m_ = (mbase) + r0 + 8 * (r);                                                         \
    const bool mok_ = m_ < m_end;                                                                  \
    const int mc_ = mok_ ? m_ : m_end - 1;                                                         \
    sa##r = *reinterpret_cast<const float4*>(g.A + (size_t)mc_ * g.lda + (a_cok ? ca : 0));        \
    long srow_ = mc_;                                                                              \
    bool bok_ = mok_ && b_cok;                                                                     \
    if (g.conv) {                                                                                  \
      int tile_, rem_, nn_, ll_;                                                                   \
      if (g.sh_gl >= 0) { /* power-of-two grid (every shipped config: 32 x 16): shifts, no division */ \
        tile_ = mc_ >> g.sh_grid; rem_ = mc_ & (grid_sz - 1);                                      \
        nn_ = (rem_ >> g.sh_gl) + tap_dn; ll_ = (rem_ & (g.gl - 1)) + tap_dl;                      \
      } else {                                                                                     \
        tile_ = mc_ / grid_sz; rem_ = mc_ - tile_ * grid_sz;                                       \
        nn_ = rem_ / g.gl + tap_dn; ll_ = rem_ % g.gl + tap_dl;                                    \
      }                                                                                            \
      bok_ = bok_ && nn_ >= 0 && nn_ < g.gn && ll_ >= 0 && ll_ < g.gl;                             \
      nn_ = min(max(nn_, 0), g.gn - 1);                                                            \
      ll_ = min(max(ll_, 0), g.gl - 1);                                                            \
      srow_ = (long)tile_ * grid_sz + (long)nn_ * g.gl + ll_;                                      \
    }                                                                                              \
    sb##r = *reinterpret_cast<const float4*>(g.B + (size_t)srow_ * g.ldb + (b_cok ? b_col : 0));   \
    ok_a = (ok_a & ~(1u << (r))) | ((mok_ && a_cok) ? (1u << (r)) : 0u);                            \
    ok_b = (ok_b & ~(1u << (r))) | (bok_ ? (1u << (r)) : 0u);                                       \
  } while (0)
#define TN_STORE_ROW(stage, r)                                                                     \
  do {                                                                                             \
    float* pa_ = sm + (stage) * 2 * TILE_F + (r0 + 8 * (r)) * TN_ROWF + 4 * c16;                   \
    float4 va_ = sa##r, vb_ = sb##r;                                                               \
    vb_.x -= bsub.x; vb_.y -= bsub.y; vb_.z -= bsub.z; vb_.w -= bsub.w;                             \
    if (!(ok_a & (1u << (r)))) va_ = make_float4(0.f, 0.f, 0.f, 0.f);                              \
    if (!(ok_b & (1u << (r)))) vb_ = make_float4(0.f, 0.f, 0.f, 0.f);                              \
    *reinterpret_cast<float4*>(pa_) = va_;                                                         \
    *reinterpret_cast<float4*>(pa_ + TILE_F) = vb_;                                                \
  } while (0)

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

  const int nk = (m_end - m_begin + 31) / 32;
  if (nk > 0) {
    TN_LOAD_ROW(0, m_begin); TN_LOAD_ROW(1, m_begin); TN_LOAD_ROW(2, m_begin); TN_LOAD_ROW(3, m_begin);
    TN_STORE_ROW(0, 0); TN_STORE_ROW(0, 1); TN_STORE_ROW(0, 2); TN_STORE_ROW(0, 3);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    const float* pa = sm + cur * 2 * TILE_F + hh * TN_ROWF + wm * 64 + li;
    const float* pb = sm + cur * 2 * TILE_F + TILE_F + hh * TN_ROWF + wn * 64 + li;
    // operand fragments run two MFMA groups (~512 pipe cycles) ahead of their use in a 4-deep register ring,
    // and the next tile's global loads (with their conv index arithmetic) are spread between the groups, so
    // neither an LDS round trip nor the address VALU work ever sits in front of an idle matrix pipe.
    float fa0A, fa1A, fb0A, fb1A, fa0B, fa1B, fb0B, fb1B, fa0C, fa1C, fb0C, fb1C, fa0D, fa1D, fb0D, fb1D;
#define TN_RD(S, s2)                                                                               \
  do {                                                                                             \
    fa0##S = pa[2 * (s2) * TN_ROWF]; fa1##S = pa[2 * (s2) * TN_ROWF + 32];                          \
    fb0##S = pb[2 * (s2) * TN_ROWF]; fb1##S = pb[2 * (s2) * TN_ROWF + 32];                          \
  } while (0)
#define TN_MM(S)                                                                                   \
  do {                                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0##S, fb0##S, acc[0][0], 0, 0, 0);          \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0##S, fb1##S, acc[0][1], 0, 0, 0);          \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1##S, fb0##S, acc[1][0], 0, 0, 0);          \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1##S, fb1##S, acc[1][1], 0, 0, 0);          \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
    const int mb = m_begin + (kt + 1) * 32;
    TN_RD(A, 0); TN_RD(B, 1);
    __builtin_amdgcn_sched_barrier(0);
    TN_RD(C, 2);  TN_MM(A);
    TN_RD(D, 3);  TN_MM(B);
    TN_LOAD_ROW(0, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(A, 4);  TN_MM(C);
    TN_RD(B, 5);  TN_MM(D);
    TN_RD(C, 6);  TN_MM(A);
    TN_LOAD_ROW(1, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(D, 7);  TN_MM(B);
    TN_RD(A, 8);  TN_MM(C);
    TN_RD(B, 9);  TN_MM(D);
    TN_LOAD_ROW(2, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(C, 10); TN_MM(A);
    TN_RD(D, 11); TN_MM(B);
    TN_RD(A, 12); TN_MM(C);
    TN_LOAD_ROW(3, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(B, 13); TN_MM(D);
    TN_RD(C, 14); TN_MM(A);
    TN_RD(D, 15); TN_MM(B);
    TN_MM(C);
    TN_MM(D);
#undef TN_RD
#undef TN_MM
    if (more) { TN_STORE_ROW(cur ^ 1, 0); TN_STORE_ROW(cur ^ 1, 1); TN_STORE_ROW(cur ^ 1, 2); TN_STORE_ROW(cur ^ 1, 3); }
    __syncthreads();
  }
  float* Cs = g.C + (size_t)split * g.N1 * g.ldc;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n2_0 + wn * 64 + ni * 32 + li;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n1_0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < g.N1 && col < g.N2) Cs[(size_t)row * g.ldc + col] = acc[mi][ni][r];
      }
  }
#undef TN_LOAD_ROW
#undef TN_STORE_ROW
}

// 8-wave variant (64 x 32 of C per wave, four waves per SIMD with two resident blocks): same LDS image and K-step.
__global__ __launch_bounds__(512, 2) void gemm_tn_w8_kernel(const TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);           // [stage][A|B][32][TN_ROWF]
  constexpr int TILE_F = 32 * TN_ROWF;
  const int tiles_n2 = (g.N2 + 127) / 128;
  const int tm = blockIdx.x / tiles_n2, tn = blockIdx.x % tiles_n2;
  const int n1_0 = tm * 128, n2_0 = tn * 128;
  const int split = blockIdx.y;
  const int m_begin = split * g.m_per_split;
  const int m_end = min(g.M, m_begin + g.m_per_split);

  const int t = threadIdx.x;
  const int c16 = t & 31, r0 = t >> 5;                  // chunk column (4 floats), base row (0..15) + 16*r
  // column validity / conv tap of this thread's chunk (fixed over the K loop)
  const int ca = n1_0 + 4 * c16, cb = n2_0 + 4 * c16;
  const bool a_cok = ca < g.N1, b_cok = cb < g.N2;      // N1, N2 multiples of 4 (checked on the host)
  int tap_dn = 0, tap_dl = 0, b_col = cb;
  if (g.conv && b_cok) {
    const int tap = cb / g.cin;
    tap_dn = tap / 3 - 1;
    tap_dl = tap - (tap / 3) * 3 - 1;
    b_col = cb - tap * g.cin;
  }
  float4 bsub = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.b_sub && b_cok) bsub = *reinterpret_cast<const float4*>(g.b_sub + cb);
  const int grid_sz = g.conv ? g.gn * g.gl : 1;

  // TN_LOAD_ROW only ISSUES the two 16-byte loads of staging row r and records their predicates; masking and
  // the b_sub recentre happen in TN_STORE_ROW, a whole K-step later, so nothing waits on memory in between.
  float4 sa0, sa1, sb0, sb1;
  unsigned ok_a = 0, ok_b = 0;
#define TN_LOAD_ROW(r, mbase)                                                                      \
  do {                                                                                             \
    const int m_ = (mbase) + r0 + 16 * (r);                                                         \
    const bool mok_ = m_ < m_end;                                                                  \
    const int mc_ = mok_ ? m_ : m_end - 1;                                                         \
    sa##r = *reinterpret_cast<const float4*>(g.A + (size_t)mc_ * g.lda + (a_cok ? ca : 0));        \
    long srow_ = mc_;                                                                              \
    bool bok_ = mok_ && b_cok;                                                                     \
    if (g.conv) {                                                                                  \
      int tile_, rem_, nn_, ll_;                                                                   \
      if (g.sh_gl >= 0) { /* power-of-two grid (every shipped config: 32 x 16): shifts, no division */ \
        tile_ = mc_ >> g.sh_grid; rem_ = mc_ & (grid_sz - 1);                                      \
        nn_ = (rem_ >> g.sh_gl) + tap_dn; ll_ = (rem_ & (g.gl - 1)) + tap_dl;                      \
      } else {                                                                                     \
        tile_ = mc_ / grid_sz; rem_ = mc_ - tile_ * grid_sz;                                       \
        nn_ = rem_ / g.gl + tap_dn; ll_ = rem_ % g.gl + tap_dl;                                    \
      }                                                                                            \
      bok_ = bok_ && nn_ >= 0 && nn_ < g.gn && ll_ >= 0 && ll_ < g.gl;                             \
      nn_ = min(max(nn_, 0), g.gn - 1);                                                            \
      ll_ = min(max(ll_, 0), g.gl - 1);                                                            \
      srow_ = (long)tile_ * grid_sz + (long)nn_ * g.gl + ll_;                                      \
    }                                                                                              \
    sb##r = *reinterpret_cast<const float4*>(g.B + (size_t)srow_ * g.ldb + (b_cok ? b_col : 0));   \
    ok_a = (ok_a & ~(1u << (r))) | ((mok_ && a_cok) ? (1u << (r)) : 0u);                            \
    ok_b = (ok_b & ~(1u << (r))) | (bok_ ? (1u << (r)) : 0u);                                       \
  } while (0)
#define TN_STORE_ROW(stage, r)                                                                     \
  do {                                                                                             \
    float* pa_ = sm + (stage) * 2 * TILE_F + (r0 + 16 * (r)) * TN_ROWF + 4 * c16;                   \
    float4 va_ = sa##r, vb_ = sb##r;                                                               \
    vb_.x -= bsub.x; vb_.y -= bsub.y; vb_.z -= bsub.z; vb_.w -= bsub.w;                             \
    if (!(ok_a & (1u << (r)))) va_ = make_float4(0.f, 0.f, 0.f, 0.f);                              \
    if (!(ok_b & (1u << (r)))) vb_ = make_float4(0.f, 0.f, 0.f, 0.f);                              \
    *reinterpret_cast<float4*>(pa_) = va_;                                                         \
    *reinterpret_cast<float4*>(pa_ + TILE_F) = vb_;                                                \
  } while (0)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, hh = lane >> 5;

  const int nk = (m_end - m_begin + 31) / 32;
  if (nk > 0) {
    TN_LOAD_ROW(0, m_begin); TN_LOAD_ROW(1, m_begin);
    TN_STORE_ROW(0, 0); TN_STORE_ROW(0, 1);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    const float* pa = sm + cur * 2 * TILE_F + hh * TN_ROWF + wm * 64 + li;
    const float* pb = sm + cur * 2 * TILE_F + TILE_F + hh * TN_ROWF + wn * 32 + li;
    // operand fragments run two MFMA groups (~512 pipe cycles) ahead of their use in a 4-deep register ring,
    // and the next tile's global loads (with their conv index arithmetic) are spread between the groups, so
    // neither an LDS round trip nor the address VALU work ever sits in front of an idle matrix pipe.
    float fa0A, fa1A, fb0A, fa0B, fa1B, fb0B, fa0C, fa1C, fb0C, fa0D, fa1D, fb0D;
#define TN_RD(S, s2)                                                                               \
  do {                                                                                             \
    fa0##S = pa[2 * (s2) * TN_ROWF]; fa1##S = pa[2 * (s2) * TN_ROWF + 32];                          \
    fb0##S = pb[2 * (s2) * TN_ROWF];                                                               \
  } while (0)
#define TN_MM(S)                                                                                   \
  do {                                                                                             \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0##S, fb0##S, acc[0], 0, 0, 0);                \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1##S, fb0##S, acc[1], 0, 0, 0);                \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
    const int mb = m_begin + (kt + 1) * 32;
    TN_RD(A, 0); TN_RD(B, 1);
    __builtin_amdgcn_sched_barrier(0);
    TN_RD(C, 2);  TN_MM(A);
    TN_RD(D, 3);  TN_MM(B);
    TN_LOAD_ROW(0, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(A, 4);  TN_MM(C);
    TN_RD(B, 5);  TN_MM(D);
    TN_RD(C, 6);  TN_MM(A);
    TN_LOAD_ROW(1, mb);          // unconditional (rows clamp to m_end-1): straight-line code keeps vmcnt exact
    TN_RD(D, 7);  TN_MM(B);
    TN_RD(A, 8);  TN_MM(C);
    TN_RD(B, 9);  TN_MM(D);
    TN_RD(C, 10); TN_MM(A);
    TN_RD(D, 11); TN_MM(B);
    TN_RD(A, 12); TN_MM(C);
    TN_RD(B, 13); TN_MM(D);
    TN_RD(C, 14); TN_MM(A);
    TN_RD(D, 15); TN_MM(B);
    TN_MM(C);
    TN_MM(D);
#undef TN_RD
#undef TN_MM
    if (more) { TN_STORE_ROW(cur ^ 1, 0); TN_STORE_ROW(cur ^ 1, 1); }
    __syncthreads();
  }
  float* Cs = g.C + (size_t)split * g.N1 * g.ldc;
  {
    const int col = n2_0 + wn * 32 + li;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n1_0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < g.N1 && col < g.N2) Cs[(size_t)row * g.ldc + col] = acc[mi][r];
      }
  }
#undef TN_LOAD_ROW
#undef TN_STORE_ROW
}

// out[i] = sum_s part[s][i]   (fixed order)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                        int64_t n4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = reinterpret_cast<const float4*>(part)[i];
  for (int k = 1; k < splits; ++k) {
    const float4 v = reinterpret_cast<const float4*>(part)[(int64_t)k * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = s;
}


// C = epilogue(sum_s partial[s])  for the split-K path (fixed summation order)
template <int C_BF16>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, acx_gemm_desc d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)d.M * d.N;
  if (i >= total) return;
  const int row = (int)(i / d.N), col = (int)(i - (int64_t)row * d.N);
  float v = 0.f;
  for (int s = 0; s < splits; ++s) v += part[(size_t)s * total + i];
  if (d.bias) v += d.bias[col];
  if (d.act == ACX_ACT_QUICKGELU) v = v * (1.f / (1.f + __expf(-1.702f * v)));
  else if (d.act == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;
  if (d.residual) v += d.residual[(size_t)row * d.ldr + col];
  if constexpr (C_BF16) ((u16*)d.C)[(size_t)row * d.ldc + col] = f2bf(v);
  else ((float*)d.C)[(size_t)row * d.ldc + col] = v;
}

}  // namespace

extern "C" int acx_gemm(acx_ctx* ctx, const acx_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: null pointer%s");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: empty shape%s");
  const int prec = d->prec;
  if (prec != ACX_PREC_F32 && prec != ACX_PREC_BF16)
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown precision%s");
  const int a_bf16 = d->a_dtype == ACX_BF16, c_bf16 = d->c_dtype == ACX_BF16;
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
    const int ke = prec == ACX_PREC_F32 ? 32 : 64;
    if (d->cin <= 0 || d->K != 9 * d->cin || d->cin % ke || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad conv3x3 geometry%s");
    if (d->a_sub) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: a_sub with conv3x3%s");
  } else if (d->amap == ACX_AMAP_TESTTILE) {
    if (d->seg <= 0 || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl * d->seg))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad test-tile geometry%s");
  } else if (d->amap != ACX_AMAP_IDENTITY) {
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown amap%s");
  }
  if ((d->pos0 != nullptr) != (d->pos1 != nullptr) || (d->pos0 && (d->gn <= 0 || d->gl <= 0)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: pos0/pos1 need both pointers and a grid%s");

  Args g;
  g.d = *d;
  const int tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
  g.trace = getenv("ACX_TRACE_PTR") ? (long long*)strtoull(getenv("ACX_TRACE_PTR"), nullptr, 0) : nullptr;   // ACX_TRACE builds only
  dim3 grid((unsigned)(tiles_m * g.tiles_n)), block(NTHREADS);
  g.ksplit = 1; g.kchunk = 0; g.partial = nullptr;
  const size_t lds = 4 * TILE_B;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_GEMM, (hipStream_t)stream);
  if (ctx && ctx->prof_on) ctx->prof_gemm_flops += 2.0 * d->M * (double)d->N * d->K;
#define ACX_LAUNCH(P, AB, CB, F, ACT, RES)                                                          \
  do {                                                                                              \
    static bool attr_done = false;                                                                  \
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
  if (fast && d->workspace) {
    // skinny problems (few tiles, long K): split K over gridDim.y so the chip is filled; partial sums are
    // combined in fixed order by splitk_reduce_kernel together with the epilogue
    const int tiles = tiles_m * g.tiles_n, nkt = d->K / ke;
    int split = 1;
    while (tiles * split * 2 <= 512 && nkt / (split * 2) >= 4 && split < 16) split *= 2;
    if (split > 1 && (size_t)split * d->M * d->N * sizeof(float) <= d->workspace_bytes) {
      g.ksplit = split;
      g.kchunk = (nkt + split - 1) / split;
      g.ksplit = (nkt + g.kchunk - 1) / g.kchunk;
      g.partial = (float*)d->workspace;
      grid.y = g.ksplit;
    }
  }
  // f32 FAST problems without split-K: the 8-wave variant (ACX_W8=0 keeps the 4-wave kernel)
  static const bool w8 = getenv("ACX_W8") ? atoi(getenv("ACX_W8")) != 0 : true;
  const bool w8_conv = d->amap == ACX_AMAP_CONV3X3 && !d->a_sub && !d->pos0 && d->K % 32 == 0 && d->cin % 32 == 0;
  if (w8 && (fast || w8_conv) && g.ksplit == 1 && prec == ACX_PREC_F32 && !c_bf16 && !a_bf16) {
#define ACX_W8L(ACT, RES, CV)                                                                       \
  do {                                                                                              \
    static bool attr_done = false;                                                                  \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_f32_w8_kernel<ACT, RES, CV>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_f32_w8_kernel<ACT, RES, CV>), grid, dim3(512), lds, s, g);             \
  } while (0)
    if (w8_conv) {
      if (d->act == ACX_ACT_LEAKYRELU) { if (d->residual) ACX_W8L(2, 1, 1); else ACX_W8L(2, 0, 1); }
      else if (d->act == ACX_ACT_QUICKGELU) { if (d->residual) ACX_W8L(1, 1, 1); else ACX_W8L(1, 0, 1); }
      else { if (d->residual) ACX_W8L(0, 1, 1); else ACX_W8L(0, 0, 1); }
    } else {
      if (d->act == ACX_ACT_QUICKGELU) { if (d->residual) ACX_W8L(1, 1, 0); else ACX_W8L(1, 0, 0); }
      else { if (d->residual) ACX_W8L(0, 1, 0); else ACX_W8L(0, 0, 0); }
    }
#undef ACX_W8L
    ACX_CHECK_LAUNCH(ctx, "acx_gemm");
    return ACX_OK;
  }
  // bf16 operands already in global memory, no split: the LDS-DMA kernels (ACX_NO_DMA=1 keeps the register-staged
  // one, ACX_NO_RING=1 the 128x128 DMA kernel).  Large problems take the persistent 256x256 ring kernel.
  static const bool no_dma = getenv("ACX_NO_DMA") != nullptr;
  static const bool no_ring = getenv("ACX_NO_RING") != nullptr;
  const int rtiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
  const int ring_min = getenv("ACX_RING_MIN_TILES") ? atoi(getenv("ACX_RING_MIN_TILES")) : 512;   // (tests lower it)
  if (fast && g.ksplit == 1 && prec == ACX_PREC_BF16 && a_bf16 && !no_dma && !no_ring && d->lda % 8 == 0 && d->ldw % 8 == 0 &&
      d->K % 64 == 0 && d->N % 4 == 0 && d->ldc % 4 == 0 && (!d->residual || d->ldr % 4 == 0) &&
      !(((uintptr_t)d->C | (uintptr_t)d->residual | (uintptr_t)d->bias) & 15) && rtiles >= ring_min && !(d->act == ACX_ACT_QUICKGELU && d->residual)) {
    int ncu = 256;
    { hipDeviceProp_t prop; int dev = 0;
      static int cached = 0;
      if (!cached && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount;
      if (cached > 0) ncu = cached; }
    const dim3 rgrid((unsigned)(rtiles < ncu ? rtiles : ncu));
#define ACX_RING_L(CB, ACT, RES)                                                                    \
  do {                                                                                              \
    static bool attr_done = false;                                                                  \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_ring_kernel<CB, ACT, RES>,                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * RG_STAGE_B + 8 * 4096)); \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<CB, ACT, RES>), rgrid, dim3(512), (size_t)(2 * RG_STAGE_B + 8 * 4096), s, g); \
  } while (0)
#define ACX_RING_SEL()                                                                              \
  do {                                                                                              \
    if (d->residual) { if (c_bf16) ACX_RING_L(1, 0, 1); else ACX_RING_L(0, 0, 1); }                 \
    else if (d->act == ACX_ACT_QUICKGELU) { if (c_bf16) ACX_RING_L(1, 1, 0); else ACX_RING_L(0, 1, 0); } \
    else { if (c_bf16) ACX_RING_L(1, 0, 0); else ACX_RING_L(0, 0, 0); }                             \
  } while (0)
    ACX_RING_SEL();
#undef ACX_RING_SEL
#undef ACX_RING_L
  } else
  if (fast && g.ksplit == 1 && prec == ACX_PREC_BF16 && a_bf16 && !no_dma && d->lda % 8 == 0 && d->ldw % 8 == 0) {
    const size_t dlds = 2 * DMA_STAGE_B;
#define ACX_DMA(CB, ACT, RES)                                                                       \
  do {                                                                                              \
    static bool attr_done = false;                                                                  \
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

extern "C" size_t acx_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2) {
  const int splits = tn_choose_splits(M, N1, N2);
  return splits > 1 ? (size_t)splits * N1 * N2 * sizeof(float) : 0;
}

extern "C" int acx_gemm_tn(acx_ctx* ctx, const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc,
                           int32_t M, int32_t N1, int32_t N2, const float* b_sub, int32_t conv, int32_t gn, int32_t gl,
                           int32_t cin, void* workspace, size_t workspace_bytes, void* stream) {
  if (!A || !B || !C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: null pointer%s");
  if (M <= 0 || N1 <= 0 || N2 <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: empty shape%s");
  if (N1 % 4 || N2 % 4 || lda % 4 || ldb % 4 || ldc != N2 || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: N1/N2/lda/ldb must be multiples of 4, C dense, 16-byte aligned%s");
  if (conv && (cin <= 0 || N2 != 9 * cin || cin % 4 || gn <= 0 || gl <= 0 || M % (gn * gl)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: bad conv geometry%s");
  if (b_sub && ((uintptr_t)b_sub & 15)) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm_tn: b_sub alignment%s");
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
  AcxProfScope prof__(ctx, ACX_K_GEMM, s);
  if (ctx && ctx->prof_on) ctx->prof_gemm_flops += 2.0 * M * (double)N1 * N2;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  static const bool tn_w8 = getenv("ACX_TN_W8") ? atoi(getenv("ACX_TN_W8")) != 0 : true;
  if (tn_w8) {
    static bool attr8_done = false;
    if (!attr8_done) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr8_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_w8_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(512), lds, s, g);
  } else
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(NTHREADS), lds, s, g);
  if (splits > 1) {
    const int64_t n4 = (int64_t)N1 * N2 / 4;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, C, n4,
                       splits);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_gemm_tn");
  return ACX_OK;
}
