// acx_gemm_tn -- weight-gradient GEMM kernels (4-wave and 8-wave) and their fixed-order reduce
// (included by acx_gemm.hip inside its anonymous namespace; shares Args / tile constants / helpers defined there)
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
// body of the 8-wave weight-gradient kernel for output tile `bx_` and m-split `by_` (shared by the single-problem and the
// grouped launch)
__device__ __forceinline__ void tn_w8_body(const TnArgs& g, const int bx_, const int by_) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);           // [stage][A|B][32][TN_ROWF]
  constexpr int TILE_F = 32 * TN_ROWF;
  const int tiles_n2 = (g.N2 + 127) / 128;
  const int tm = bx_ / tiles_n2, tn = bx_ % tiles_n2;
  const int n1_0 = tm * 128, n2_0 = tn * 128;
  const int split = by_;
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
  // rows r0 and r0 + 16 of every K-step: with gl | 16 on a power-of-two grid their column coordinate is a constant
  const bool fastconv = g.conv && g.sh_gl >= 0 && g.gl <= 16 && (m_begin & 31) == 0;
  const int ll_raw = (r0 & (g.gl - 1)) + tap_dl;
  const bool ll_ok = ll_raw >= 0 && ll_raw < g.gl;
  const int ll_c = min(max(ll_raw, 0), max(g.gl - 1, 0));

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
      if (fastconv) {  /* gl | 16, power-of-two grid, m_begin % 32 == 0: the column coordinate of this thread's rows */ \
        /* never changes over the K loop (precomputed, clamped), only the grid row moves */        \
        int nn_ = ((mc_ >> g.sh_gl) & (g.gn - 1)) + tap_dn;                                        \
        bok_ = bok_ && ll_ok && (unsigned)nn_ < (unsigned)g.gn;                                    \
        nn_ = min(max(nn_, 0), g.gn - 1);                                                          \
        srow_ = (long)(mc_ & ~(grid_sz - 1)) + nn_ * g.gl + ll_c;                                  \
      } else {                                                                                     \
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
      }                                                                                            \
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

__global__ __launch_bounds__(512, 2) void gemm_tn_w8_kernel(const TnArgs g) { tn_w8_body(g, (int)blockIdx.x, (int)blockIdx.y); }

// Several small weight-gradient problems in ONE launch (the temporal model's to_out / to_q|kv / projection gradients of a
// data-parallel rank: 0.5-1.6 GFLOP each, 22-25 us per launch + a 6 us reduce when issued one by one): the flat block index
// is cut by a prefix table into (problem, tile, split); every problem runs exactly the blocks -- and therefore the arithmetic --
// of its own single launch.
constexpr int TN_GROUP_MAX = 8;
struct TnGroup {
  TnArgs p[TN_GROUP_MAX];
  int tiles[TN_GROUP_MAX];
  int blk0[TN_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(512, 2) void gemm_tn_w8_group_kernel(const TnGroup G) {
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i)
    if (i < G.n && G.blk0[i] <= b) k = i;
  const int local = b - G.blk0[k];
  const int tiles = G.tiles[k];
  const int by = local / tiles;
  tn_w8_body(G.p[k], local - by * tiles, by);
}

struct TnReduceGroup {
  const float* part[TN_GROUP_MAX];
  float* out[TN_GROUP_MAX];
  long long n4[TN_GROUP_MAX];
  int splits[TN_GROUP_MAX];
  int blk0[TN_GROUP_MAX + 1];
  int n;
};
// out_k[i] = sum_s part_k[s][i] for every problem of a group, one launch (fixed order)
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(const TnReduceGroup G) {
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < TN_GROUP_MAX; ++i)
    if (i < G.n && G.blk0[i] <= b) k = i;
  const long long i = (long long)(b - G.blk0[k]) * 256 + threadIdx.x;
  const long long n4 = G.n4[k];
  if (i >= n4) return;
  const float4* part = reinterpret_cast<const float4*>(G.part[k]);
  float4 s = part[i];
  for (int q = 1; q < G.splits[k]; ++q) {
    const float4 v = part[(long long)q * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(G.out[k])[i] = s;
}

// ---- 256 x 256 tiles, one 16-wave workgroup per CU, operands by LDS-DMA ("strip" geometry of gemm_f32_p256_kernel)
// The 128 x 128 kernels above re-read A once per N2 tile and B once per N1 tile: at the conv weight gradient of the UCF
// head (M = 32768, N1 = 1024, N2 = 2304) that is 4.8 GB through L2 per launch = 3.3 TB/s at the measured 1.48 ms, and
// the kernel sits at 0.66 of the f32 MFMA roof next to a forward convolution of the same flops at 0.80.  A 256 x 256 tile
// halves the bytes per flop in both directions:
//   * 16 waves (4 x 4, 64 x 64 of C each = 2 x 2 accumulators of v_mfma_f32_32x32x2_f32), four per SIMD, <= 128 VGPRs;
//   * a K-step is 32 rows of m; a wave instruction of global_load_lds_dwordx4 moves ONE m-row of 256 columns (1 KB,
//     fully coalesced), wave w stages rows 2w and 2w+1 of A and of B; LDS rows are 1152 B apart (256 floats + 128 B) so
//     that the two half-waves of a ds_read_b32 fragment read (rows 2s and 2s+1) fall into different bank halves;
//   * rows past the end of the split read A from a ZERO buffer (the product vanishes), taps outside the (gn, gl) grid read
//     B from it -- the conv gather is a per-lane source address, there is no im2col and no predicate on the LDS side;
//   * two stages (144 KB); per K-step: issue the DMA of step kt+1, wait for this wave's step kt, barrier, 16 x (4 fragment
//     reads + 4 MFMA), barrier.
constexpr int TP_ROWB = 1152;
constexpr int TP_OP_B = 32 * TP_ROWB;            // 36,864
constexpr int TP_STAGE_B = 2 * TP_OP_B;          // A | B
constexpr int TP_LDS_B = 2 * TP_STAGE_B;         // 147,456

__global__ __launch_bounds__(1024) void gemm_tn_p256_kernel(const TnArgs g, const float* __restrict__ zeros) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tiles_n2 = (g.N2 + 255) / 256;
  const int tm = blockIdx.x / tiles_n2, tn = blockIdx.x - tm * tiles_n2;
  const int n1_0 = tm * 256, n2_0 = tn * 256;
  const int split = blockIdx.y;
  const int m_begin = split * g.m_per_split;
  const int m_end = min(g.M, m_begin + g.m_per_split);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, hh = lane >> 5;

  // ---- DMA sources: lane -> columns 4 lane .. 4 lane + 3 of the tile (clamped into the matrix: masked at the store)
  const int ca = min(n1_0 + 4 * lane, g.N1 - 4), cb = min(n2_0 + 4 * lane, g.N2 - 4);
  int tap_dn = 0, tap_dl = 0, b_col = cb;
  if (g.conv) {
    const int tap = cb / g.cin;
    tap_dn = tap / 3 - 1;
    tap_dl = tap - (tap / 3) * 3 - 1;
    b_col = cb - tap * g.cin;
  }
  const int grid_sz = g.conv ? g.gn * g.gl : 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(p2_lds_t*)smem;
#define TP_DMA1(gptr, ldsaddr)                                                                     \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
#define TP_ISSUE(stage, kt_)                                                                       \
  do {                                                                                             \
    _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                             \
      const int m_ = m_begin + (kt_) * 32 + 2 * wave + rr;                                         \
      const bool mok_ = m_ < m_end;                                                                \
      const int mc_ = mok_ ? m_ : m_end - 1;                                                       \
      const float* pa_ = mok_ ? g.A + (size_t)mc_ * g.lda + ca : zeros;                            \
      const float* pb_;                                                                            \
      if (g.conv) {                                                                                \
        int tile_, rem_, n0_;                                                                      \
        if (g.sh_gl >= 0) { tile_ = mc_ >> g.sh_grid; rem_ = mc_ & (grid_sz - 1); n0_ = rem_ >> g.sh_gl; }   \
        else { tile_ = mc_ / grid_sz; rem_ = mc_ - tile_ * grid_sz; n0_ = rem_ / g.gl; }           \
        const int nn_ = n0_ + tap_dn, ll_ = rem_ - n0_ * g.gl + tap_dl;                            \
        const bool ok_ = (unsigned)nn_ < (unsigned)g.gn && (unsigned)ll_ < (unsigned)g.gl;         \
        pb_ = ok_ ? g.B + ((size_t)tile_ * grid_sz + (size_t)nn_ * g.gl + ll_) * g.ldb + b_col : zeros; \
      } else {                                                                                     \
        pb_ = g.B + (size_t)mc_ * g.ldb + cb;                                                      \
      }                                                                                            \
      const unsigned s_ = lds0 + (stage) * TP_STAGE_B + (2 * wave + rr) * TP_ROWB;                 \
      TP_DMA1(pa_, s_);                                                                            \
      TP_DMA1(pb_, s_ + TP_OP_B);                                                                  \
    }                                                                                              \
  } while (0)

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  const int nk = (m_end - m_begin + 31) / 32;
  const int fa = hh * TP_ROWB + (wm * 64 + li) * 4;
  const int fb = TP_OP_B + hh * TP_ROWB + (wn * 64 + li) * 4;
  if (nk > 0) TP_ISSUE(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      TP_ISSUE(cur ^ 1, kt + 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // this wave's 4 rows of step kt have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                            // ... and everybody else's
    const char* sS = smem + cur * TP_STAGE_B;
    float a0 = *reinterpret_cast<const float*>(sS + fa), a1 = *reinterpret_cast<const float*>(sS + fa + 128);
    float b0 = *reinterpret_cast<const float*>(sS + fb), b1 = *reinterpret_cast<const float*>(sS + fb + 128);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (s2 + 1 < 16) {
        const int o = 2 * (s2 + 1) * TP_ROWB;
        na0 = *reinterpret_cast<const float*>(sS + fa + o); na1 = *reinterpret_cast<const float*>(sS + fa + o + 128);
        nb0 = *reinterpret_cast<const float*>(sS + fb + o); nb1 = *reinterpret_cast<const float*>(sS + fb + o + 128);
      }
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // stage `cur` is free for the DMA of step kt + 2
  }
#undef TP_ISSUE
#undef TP_DMA1
  float* Cs = g.C + (size_t)split * g.N1 * g.ldc;
#define TP_ST(ACC, mi, ni)                                                                         \
  do {                                                                                             \
    const int col = n2_0 + wn * 64 + (ni) * 32 + li;                                               \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                               \
      const int row = n1_0 + wm * 64 + (mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;                \
      if (row < g.N1 && col < g.N2) Cs[(size_t)row * g.ldc + col] = ACC[r];                        \
    }                                                                                              \
  } while (0)
  TP_ST(acc00, 0, 0); TP_ST(acc01, 0, 1); TP_ST(acc10, 1, 0); TP_ST(acc11, 1, 1);
#undef TP_ST
}

__global__ void tn_zero_page_kernel(float4* __restrict__ page) { page[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); }

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
