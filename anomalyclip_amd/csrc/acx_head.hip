// HBM-bound / small kernels of the AnomalyCLIP head and the frame front-end:
//   acx_vit_patches, acx_text_directions, acx_selector_project, acx_selector_bn, acx_bn_stats,
//   acx_axial_attention, acx_class_probs, acx_cast_bf16, acx_colsum
// Every kernel makes one coalesced pass over its input with 16-byte accesses where the layout
// allows; none of them is reshaped into a GEMM.
#include "acx_internal.h"

namespace {

// ------------------------------------------------------------------ patch im2col
// out[(f*g*g + gy*g + gx), c*P*P + ky*P + kx] = frames[f, c, gy*P+ky, gx*P+kx]   (P % 4 == 0)
template <int OUT_BF16>
__global__ __launch_bounds__(256) void patches_kernel(const float* __restrict__ frames, void* __restrict__ out,
                                                      int64_t total4, int R, int P, int g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int K = 3 * P * P, K4 = K / 4;
  const int64_t row = i / K4;
  const int k = (int)(i - row * K4) * 4;
  const int c = k / (P * P), rem = k - c * P * P, ky = rem / P, kx = rem - ky * P;
  const int64_t f = row / (g * g);
  const int tok = (int)(row - f * g * g), gy = tok / g, gx = tok - gy * g;
  const float4 v = *reinterpret_cast<const float4*>(
      frames + ((f * 3 + c) * R + gy * P + ky) * (int64_t)R + gx * P + kx);
  if constexpr (OUT_BF16 >= 2) {
    // three bf16 planes hi | mid | lo of the f32 pixel in K-panel layout (OUT_BF16 == 3, ACX_BF16X2P: hi and mid only) (ACX_BF16X3P: plane = [K / 32][rows][32]): the A operand
    // of the patch embedding as a pairs = 6 product (ACX_PREC_F32X6)
    const float x[4] = {v.x, v.y, v.z, v.w};
    u16 h[4], m[4], l[4];
    if constexpr (OUT_BF16 == 4) {                     // (ACX_F16X2P: two fp16 planes)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const uint32_t ph_ = f2h2(x[e], x[e + 1]);
        const uint32_t pl_ = f2h2(x[e] - h2f_lo(ph_), x[e + 1] - h2f_hi(ph_));
        h[e] = (u16)(ph_ & 0xffffu); h[e + 1] = (u16)(ph_ >> 16);
        m[e] = (u16)(pl_ & 0xffffu); m[e + 1] = (u16)(pl_ >> 16);
        l[e] = l[e + 1] = 0;
      }
    } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = f2bf(x[e]);
      if ((h[e] & 0x7fffu) == 0x7f80u && (__float_as_uint(x[e]) & 0x7fffffffu) < 0x7f800000u) h[e] = (u16)(__float_as_uint(x[e]) >> 16);
      const float r1 = x[e] - bf2f(h[e]);
      m[e] = f2bf(r1);
      l[e] = f2bf(r1 - bf2f(m[e]));
    }
    }
    const int64_t rows = total4 / K4, plane = rows * K;
    u16* d0 = (u16*)out + ((int64_t)(k >> 5) * rows + row) * 32 + (k & 31);
    *reinterpret_cast<uint2*>(d0) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(d0 + plane) = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
    if constexpr (OUT_BF16 == 2)
      *reinterpret_cast<uint2*>(d0 + 2 * plane) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  } else if constexpr (OUT_BF16 == 1) {
    uint2 pk;
    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>((u16*)out + row * K + k) = pk;
  } else {
    *reinterpret_cast<float4*>((float*)out + row * K + k) = v;
  }
}

// ------------------------------------------------------------------ text directions
__global__ __launch_bounds__(256) void text_dirs_kernel(const float* __restrict__ text, const float* __restrict__ nc,
                                                        float* __restrict__ dirs, int D, int normal_id) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const int src = c < normal_id ? c : c + 1;      // selector_model.py:44-50
  float ss = 0.f;
  for (int e = threadIdx.x; e < D; e += 256) {
    const float v = text[(size_t)src * D + e] - nc[e];
    ss += v * v;
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  for (int e = threadIdx.x; e < D; e += 256)
    dirs[(size_t)c * D + e] = (text[(size_t)src * D + e] - nc[e]) / norm;
}

// ------------------------------------------------------------------ selector projection
// raw[r, c] = (x[r,:] - nc) . dirs[c,:]      one wavefront per row, directions resident in LDS
template <int VPL>
__global__ __launch_bounds__(256) void selector_project_kernel(const float* __restrict__ x, const float* __restrict__ nc,
                                                               const float* __restrict__ dirs, float* __restrict__ raw,
                                                               int64_t rows, int C1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sd = reinterpret_cast<float*>(smem);     // [C1][D]
  constexpr int D = 64 * VPL;
  for (int i = threadIdx.x; i < C1 * D / 4; i += 256)
    reinterpret_cast<float4*>(sd)[i] = reinterpret_cast<const float4*>(dirs)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cen[VPL];
  load_row<VPL>(nc, lane, cen);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[VPL];
    load_row<VPL>(x + row * D, lane, v);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] -= cen[i];
    float mine = 0.f;
    for (int c = 0; c < C1; ++c) {
      float dd[VPL];
      load_row<VPL>(sd + c * D, lane, dd);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) acc += v[i] * dd[i];
      acc = wave_sum(acc);
      if (lane == c) mine = acc;
    }
    if (lane < C1) raw[row * C1 + lane] = mine;
  }
}

// ------------------------------------------------------------------ selector projection on the f32 MFMA
// raw[r, c] = (x[r,:] - nc) . dirs[c,:] is a skinny GEMM (K = D, N = C-1 <= 64): the wave-per-row kernel above reads all
// C-1 directions from LDS for every row (26 KB of LDS reads + 78 cross-lane adds per 2 KB of HBM at C-1 = 13), which
// bounds it at ~1.5 TB/s.  Here a wave owns 16 rows and NT <= 4 column tiles of 16 directions and runs
// v_mfma_f32_16x16x4_f32: lane (i = lane & 15, q = lane >> 4) loads 32 contiguous bytes of row i per 32-wide K-step
// straight from HBM into registers (every wave-level load covers 16 full 128-B lines), the SAME k-permutation is applied
// to the direction fragments it reads from LDS (rows padded by 16 B: conflict-free ds_read_b128), and the centroid is
// subtracted in registers like the reference does before its matmul (selector_model.py:54,62).  All K-steps' loads of a
// 16-row group are issued before the first MFMA (32 KB in flight per wave).
// STATS: per-column (sum, sum of squares) of the rows this block produced, accumulated in f64 in a fixed order and
// written to part[block][2][16*NT]; bn_finalize_kernel adds the blocks in order -> training BatchNorm1d statistics
// without re-reading raw.
// (non-temporal loads measured no different: 17.5 us either way; the rows are re-read by the direction gradient, so they
// keep the default policy)
__device__ __forceinline__ float4 selm_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <int D, int NT, bool STATS>
__global__ __launch_bounds__(256, 2) void selector_project_mfma_kernel(const float* __restrict__ x, const float* __restrict__ nc,
                                                                       const float* __restrict__ dirs, float* __restrict__ raw,
                                                                       int64_t rows, int C1, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LD = D + 4;                      // floats per direction row in LDS (16-B pad)
  constexpr int KS = D / 32;                     // K-steps
  constexpr int CH = KS >= 8 ? 4 : KS / 2;       // K-steps per staged chunk
  constexpr int NCH = KS / CH;                   // even for every supported D
  static_assert(NCH % 2 == 0 && NCH * CH == KS, "chunking");
  float* sd = reinterpret_cast<float*>(smem);    // [16*NT][LD], rows >= C1 are zero
  float* sc = sd + 16 * NT * LD;                 // [D] centroid
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, q = lane >> 4;
  const int64_t ngroups = (rows + 15) >> 4;
  // the first group's first chunk of x is requested BEFORE the directions are staged: the HBM latency of the row stream
  // runs under the LDS fill (a wave processes one or two groups per launch at the benchmark shapes)
  float4 a0[CH][2], a1[CH][2];
  const int64_t grp0 = (int64_t)blockIdx.x * 4 + wave;
  {
    int64_t row = grp0 * 16 + li;
    if (row >= rows) row = rows - 1;
    const float* xr = x + row * D + 8 * q;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      a0[j][0] = selm_ld(xr + 32 * j);
      a0[j][1] = selm_ld(xr + 32 * j + 4);
    }
  }
  for (int i = threadIdx.x; i < 16 * NT * (D / 4); i += 256) {
    const int r = i / (D / 4), k4 = i - r * (D / 4);
    const float4 v = r < C1 ? reinterpret_cast<const float4*>(dirs + (size_t)r * D)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(sd + r * LD + 4 * k4) = v;
  }
  for (int i = threadIdx.x; i < D / 4; i += 256) reinterpret_cast<float4*>(sc)[i] = reinterpret_cast<const float4*>(nc)[i];
  __syncthreads();
  double s_[NT], q_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { s_[t] = 0.0; q_[t] = 0.0; }
  for (int64_t grp = grp0; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    int64_t row = grp * 16 + li;
    if (row >= rows) row = rows - 1;                                   // clamped; masked at the store
    const float* xr = x + row * D + 8 * q;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K in chunks of CH steps, two register sets in ping-pong: chunk c+1 is in flight while chunk c feeds the MFMAs
    // (all-K-up-front costs 128 VGPRs of staging at D = 512 and spills)
#define SELM_LOAD(A, c)                                                        \
  _Pragma("unroll") for (int j = 0; j < CH; ++j) {                             \
    A[j][0] = selm_ld(xr + 32 * ((c) * CH + j));                               \
    A[j][1] = selm_ld(xr + 32 * ((c) * CH + j) + 4);                           \
  }
#define SELM_MMA(A, c)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < CH; ++j) {                                                       \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                      \
      const int k0 = 32 * ((c) * CH + j) + 8 * q + 4 * h;                                                \
      const float4 c4 = *reinterpret_cast<const float4*>(sc + k0);                                       \
      const float4 xa = make_float4(A[j][h].x - c4.x, A[j][h].y - c4.y, A[j][h].z - c4.z, A[j][h].w - c4.w); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                   \
        const float4 b4 = *reinterpret_cast<const float4*>(sd + (16 * t + li) * LD + k0);                \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, b4.x, acc[t], 0, 0, 0);                      \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, b4.y, acc[t], 0, 0, 0);                      \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, b4.z, acc[t], 0, 0, 0);                      \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, b4.w, acc[t], 0, 0, 0);                      \
      }                                                                                                  \
      /* keep the direction fragments of later K-steps out of this one: with the chunk loop fully unrolled (D = 256) the */ \
      /* scheduler hoisted every ds_read_b128 of a chunk above its first MFMA (NT = 3, 4: 108-176 spilled VGPRs)        */ \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
  }
    if (grp != grp0) { SELM_LOAD(a0, 0) }
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {
      SELM_LOAD(a1, c + 1)
      SELM_MMA(a0, c)
      if (c + 2 < NCH) { SELM_LOAD(a0, c + 2) }
      SELM_MMA(a1, c + 1)
    }
#undef SELM_LOAD
#undef SELM_MMA
    // lane holds column 16 t + li of rows grp*16 + 4 q + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t rr = grp * 16 + 4 * q + r;
      if (rr < rows) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int c = 16 * t + li;
          if (c < C1) raw[rr * C1 + c] = acc[t][r];
          if constexpr (STATS) { s_[t] += (double)acc[t][r]; q_[t] += (double)acc[t][r] * (double)acc[t][r]; }
        }
      }
    }
  }
  if constexpr (STATS) {
    __syncthreads();                                                   // directions no longer needed: reuse LDS
    double* red = reinterpret_cast<double*>(smem);                     // [4 waves][2][16*NT]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double s = s_[t], qq = q_[t];
      s += __shfl_xor(s, 16, 64);  qq += __shfl_xor(qq, 16, 64);      // the four row quarters (fixed order)
      s += __shfl_xor(s, 32, 64);  qq += __shfl_xor(qq, 32, 64);
      if (q == 0) { red[(wave * 2 + 0) * 16 * NT + 16 * t + li] = s; red[(wave * 2 + 1) * 16 * NT + 16 * t + li] = qq; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * 16 * NT) {
      const int k = threadIdx.x / (16 * NT), c = threadIdx.x - k * 16 * NT;
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += red[(w * 2 + k) * 16 * NT + c];
      part[(size_t)blockIdx.x * 2 * 16 * NT + threadIdx.x] = v;
    }
  }
}

// LDS-DMA variant of the kernel above (D % 64 == 0, up to two column tiles): the A fragment of the 16x16x4 MFMA gives a
// row only FOUR lanes, so a direct global_load_dwordx4 covers 64 B of each of 16 rows -- every 128-B line is touched by two
// instructions and the kernel sits at 6.2 B/clk/CU of half-line accesses against the ~10 B/clk a CU's vector-memory path
// sustains (3.9 TB/s).  Here a wave streams its 16-row groups through a private LDS ring instead: a K chunk is 64 floats
// (256 B per row), one global_load_lds_dwordx4 moves 4 rows x 256 B (sixteen lanes per row: full lines), the 16-B slot of
// row r holding source chunk (slot ^ r) so that the fragment reads (ds_read_b128, 16 rows x one slot) are conflict-free;
// two 4 KB stages per wave, the chunks of consecutive groups form one stream (the next group's first chunks are in flight
// during the stores), one workgroup of eight waves per CU.
typedef __attribute__((address_space(3))) void selm_lds_t;
constexpr int SELD_ST = 2;                      // ring stages per wave (three measured no faster: 16.6 vs 15.5 us)
template <int D, int NT, bool STATS>
__global__ __launch_bounds__(512) void selector_project_dma_kernel(const float* __restrict__ x, const float* __restrict__ nc,
                                                                   const float* __restrict__ dirs, float* __restrict__ raw,
                                                                   int64_t rows, int C1, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int LD = D + 4;
  constexpr int NCK = D / 64;                    // 64-float chunks per row
  char* ring = smem;                             // [8 waves][SELD_ST stages][16 rows][256 B]
  float* sd = reinterpret_cast<float*>(smem + 8 * SELD_ST * 4096);   // [16*NT][LD]
  float* sc = sd + 16 * NT * LD;                 // [D]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int64_t ngroups = (rows + 15) >> 4;
  const int64_t gstride = (int64_t)gridDim.x * 8;
  const int64_t grp0 = (int64_t)blockIdx.x * 8 + wave;
  const int64_t mygroups = grp0 < ngroups ? (ngroups - grp0 + gstride - 1) / gstride : 0;
  const int64_t S = mygroups * NCK;              // this wave's chunk stream
  const unsigned ring0 = (unsigned)(uintptr_t)(selm_lds_t*)ring + wave * (SELD_ST * 4096);
  const int drow = lane >> 4, dpos = lane & 15;  // DMA: instruction i moves rows 4 i .. 4 i + 3; lane -> (row, 16-B slot)
#define SELD_DMA1(gptr, ldsaddr)                                                                   \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
#define SELD_ISSUE(s_)                                                                             \
  do {                                                                                             \
    const int64_t g_ = grp0 + ((s_) / NCK) * gstride;                                              \
    const int ck_ = (int)((s_) % NCK);                                                             \
    const unsigned st_ = ring0 + (unsigned)((s_) % SELD_ST) * 4096;                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
      const int r_ = 4 * i + drow;                                                                 \
      int64_t row_ = g_ * 16 + r_;                                                                 \
      if (row_ >= rows) row_ = rows - 1;                                                           \
      SELD_DMA1(x + row_ * D + 64 * ck_ + 4 * (dpos ^ r_), st_ + i * 1024);                        \
    }                                                                                              \
  } while (0)
  if (S > 0) SELD_ISSUE((int64_t)0);
  if (S > 1) SELD_ISSUE((int64_t)1);
  if (SELD_ST >= 3 && S > 2) SELD_ISSUE((int64_t)2);
  // directions / centroid staged while the first chunks are in flight
  for (int i = threadIdx.x; i < 16 * NT * (D / 4); i += 512) {
    const int r = i / (D / 4), k4 = i - r * (D / 4);
    const float4 v = r < C1 ? reinterpret_cast<const float4*>(dirs + (size_t)r * D)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(sd + r * LD + 4 * k4) = v;
  }
  for (int i = threadIdx.x; i < D / 4; i += 512) reinterpret_cast<float4*>(sc)[i] = reinterpret_cast<const float4*>(nc)[i];
  __syncthreads();
  double s_[NT], q_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { s_[t] = 0.0; q_[t] = 0.0; }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* myring = ring + wave * (SELD_ST * 4096) + li * 256;
  for (int64_t s = 0; s < S; ++s) {
    // chunk s landed; up to SELD_ST - 1 younger chunks (4 DMAs each) may still be in flight
    if (SELD_ST >= 3 && s + 2 < S) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ck = (int)(s % NCK);
    const char* st = myring + (s % SELD_ST) * 4096;
    float4 xa[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) xa[j][h] = *reinterpret_cast<const float4*>(st + (((8 * j + 2 * q + h) ^ li) * 16));
    // the stage is in registers: refill it -- at a group end AFTER the group's stores (gfx9 stores count in vmcnt: issued
    // behind the next DMA they would make the next counted wait cover that DMA too)
    const bool last = ck == NCK - 1;
    if (!last && s + SELD_ST < S) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SELD_ISSUE(s + SELD_ST);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k0 = 64 * ck + 32 * j + 8 * q + 4 * h;
        const float4 c4 = *reinterpret_cast<const float4*>(sc + k0);
        const float4 a4 = make_float4(xa[j][h].x - c4.x, xa[j][h].y - c4.y, xa[j][h].z - c4.z, xa[j][h].w - c4.w);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 b4 = *reinterpret_cast<const float4*>(sd + (16 * t + li) * LD + k0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc[t], 0, 0, 0);
        }
      }
    }
    if (last) {                                                        // group complete: store, restart the accumulators
      const int64_t grp = grp0 + (s / NCK) * gstride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t rr = grp * 16 + 4 * q + r;
        if (rr < rows) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int c = 16 * t + li;
            if (c < C1) raw[rr * C1 + c] = acc[t][r];
            if constexpr (STATS) { s_[t] += (double)acc[t][r]; q_[t] += (double)acc[t][r] * (double)acc[t][r]; }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s + SELD_ST < S) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SELD_ISSUE(s + SELD_ST);
      }
    }
  }
#undef SELD_ISSUE
#undef SELD_DMA1
  if constexpr (STATS) {
    __syncthreads();                                                   // ring / directions no longer needed: reuse LDS
    double* red = reinterpret_cast<double*>(smem);                     // [8 waves][2][16*NT]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double sv = s_[t], qq = q_[t];
      sv += __shfl_xor(sv, 16, 64);  qq += __shfl_xor(qq, 16, 64);
      sv += __shfl_xor(sv, 32, 64);  qq += __shfl_xor(qq, 32, 64);
      if (q == 0) { red[(wave * 2 + 0) * 16 * NT + 16 * t + li] = sv; red[(wave * 2 + 1) * 16 * NT + 16 * t + li] = qq; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * 16 * NT) {
      const int k = threadIdx.x / (16 * NT), c = threadIdx.x - k * 16 * NT;
      double v = 0.0;
      for (int w = 0; w < 8; ++w) v += red[(w * 2 + k) * 16 * NT + c];
      part[(size_t)blockIdx.x * 2 * 16 * NT + threadIdx.x] = v;
    }
  }
}

// ------------------------------------------------------------------ batch-norm statistics (two stages, fixed order)
// stage 1: block b owns a contiguous slab of rows; thread (c = t % C1, rl = t / C1) walks rows rl, rl + RL, ... of the slab
// (consecutive threads read consecutive addresses), f64 sums of x and x^2 (K2 = 0) or of dl and dl * xhat (K2 = 1);
// part[b][2][CP].  stage 2 (bn_finalize_kernel / bn_sums_finalize_kernel): one thread per (kind, column) adds the
// blocks' partials in block order.  Replaces one-workgroup-per-column kernels (13 workgroups on a 256-CU chip).
template <int K2>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int C1,
                                                         int CP, int64_t rows_per_block, double* __restrict__ part) {
  __shared__ double red[2][256];
  const int RL = 256 / C1;
  const int c = threadIdx.x % C1, rl = threadIdx.x / C1;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  double s = 0.0, qq = 0.0;
  if (rl < RL) {
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      const double v = (double)a[r * C1 + c];
      double g = 0.0;
      if constexpr (K2 != 0) g = (double)b[r * C1 + c];
      acx_bn_acc<K2>(s, qq, v, g);
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = qq;
  __syncthreads();
  if (threadIdx.x < 2 * C1) {
    const int k = threadIdx.x / C1, cc = threadIdx.x - k * C1;
    double v = 0.0;
    for (int l = 0; l < RL; ++l) v += red[k][l * C1 + cc];
    part[((size_t)blockIdx.x * 2 + k) * CP + cc] = v;
  }
}
// mean / biased / unbiased variance from the (sum, sum of squares) partials: var = E[x^2] - mean^2 in f64 (the column
// values are O(1): 53 bits leave ~1e-13 after the cancellation)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int nblocks, int CP, int C1, int64_t rows,
                                                          float* __restrict__ mean, float* __restrict__ var_b, float* __restrict__ var_u) {
  __shared__ double red[256];
  __shared__ double tot[128];
  const double v = bn_pair_total(part, nblocks, CP, C1, red);
  if ((int)threadIdx.x < 2 * C1) tot[threadIdx.x] = v;
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= C1) return;
  const double s = tot[c], qq = tot[C1 + c];
  const double n = (double)rows, m = s / n;
  double m2 = qq - s * m;                                              // sum (x - m)^2
  if (m2 < 0.0) m2 = 0.0;
  mean[c] = (float)m;
  var_b[c] = (float)(m2 / n);
  var_u[c] = rows > 1 ? (float)(m2 / (n - 1.0)) : 0.f;
}

__global__ __launch_bounds__(256) void selector_bn_kernel(const float* __restrict__ raw, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float* __restrict__ logits,
                                                          int64_t ldl, int64_t total, int C1, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t r = i / C1;
  const int c = (int)(i - r * C1);
  logits[r * ldl + c] = acx_bn_apply(raw[i], mean[c], var[c], eps);
}

// ------------------------------------------------------------------ axial attention core
// thread = (line, head, query i); K/V of the block's lines in LDS, broadcast reads.
template <int T, int E>
__global__ __launch_bounds__(256) void axial_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                         int tiles, int gn, int gl, int heads, int axis,
                                                         int64_t nlines) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int He = heads * E;
  const int lpb = 256 / (T * heads);            // lines per block
  float* sK = reinterpret_cast<float*>(smem);   // [lpb][T][He]
  float* sV = sK + lpb * T * He;
  const int t = threadIdx.x;
  const int64_t line0 = (int64_t)blockIdx.x * lpb;
  const int ld = 3 * He;
  const int other = axis == 0 ? gl : gn;        // number of lines per tile
  auto row_of = [&](int64_t line, int j) -> int64_t {
    const int64_t tile = line / other;
    const int o = (int)(line - tile * other);
    return axis == 0 ? (tile * gn + j) * gl + o : (tile * gn + o) * gl + j;
  };
  // cooperative K/V staging
  const int f4_per_tok = He / 4;
  for (int i = t; i < lpb * T * f4_per_tok; i += 256) {
    const int c4 = i % f4_per_tok, tokl = i / f4_per_tok;
    const int lb = tokl / T, j = tokl - lb * T;
    const int64_t line = line0 + lb;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (line < nlines) {
      const float* p = qkv + row_of(line, j) * ld + 4 * c4;
      kv = *reinterpret_cast<const float4*>(p + He);
      vv = *reinterpret_cast<const float4*>(p + 2 * He);
    }
    reinterpret_cast<float4*>(sK)[i] = kv;
    reinterpret_cast<float4*>(sV)[i] = vv;
  }
  __syncthreads();
  const int i = t % T, h = (t / T) % heads, lb = t / (T * heads);
  const int64_t line = line0 + lb;
  if (line >= nlines) return;
  const int64_t row = row_of(line, i);
  float q[E];
  const float scale = E == 32 ? 0.17677669529663687f : 0.25f;    // e^-0.5
#pragma unroll
  for (int c = 0; c < E / 4; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(qkv + row * ld + h * E + 4 * c);
    q[4 * c] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
  }
  float s[T];
  float mx = -INFINITY;
  const float* kb = sK + (lb * T) * He + h * E;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < E / 4; ++c) {
      const float4 kk = *reinterpret_cast<const float4*>(kb + j * He + 4 * c);
      acc += q[4 * c] * kk.x + q[4 * c + 1] * kk.y + q[4 * c + 2] * kk.z + q[4 * c + 3] * kk.w;
    }
    s[j] = acc * scale;
    mx = fmaxf(mx, s[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    s[j] = __expf(s[j] - mx);
    sum += s[j];
  }
  const float inv = 1.f / sum;
  float o[E];
#pragma unroll
  for (int e = 0; e < E; ++e) o[e] = 0.f;
  const float* vb = sV + (lb * T) * He + h * E;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float p = s[j] * inv;
#pragma unroll
    for (int c = 0; c < E / 4; ++c) {
      const float4 vv = *reinterpret_cast<const float4*>(vb + j * He + 4 * c);
      o[4 * c] += p * vv.x; o[4 * c + 1] += p * vv.y; o[4 * c + 2] += p * vv.z; o[4 * c + 3] += p * vv.w;
    }
  }
#pragma unroll
  for (int c = 0; c < E / 4; ++c)
    *reinterpret_cast<float4*>(out + row * He + h * E + 4 * c) =
        make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
}

// ------------------------------------------------------------------ eval post-processing
__global__ __launch_bounds__(256) void class_probs_kernel(const float* __restrict__ sim, const float* __restrict__ scores,
                                                          float* __restrict__ probs, int64_t rows, int C1) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float mx = -INFINITY;
  for (int c = 0; c < C1; ++c) mx = fmaxf(mx, sim[r * C1 + c]);
  float sum = 0.f;
  for (int c = 0; c < C1; ++c) sum += expf(sim[r * C1 + c] - mx);
  const float k = scores[r] / sum;
  for (int c = 0; c < C1; ++c) probs[r * C1 + c] = expf(sim[r * C1 + c] - mx) * k;
}

// x = hi + mid + lo with three bf16 (8 significant bits each: 24 in all); x - hi and (x - hi) - mid are exact in f32
// F16: two fp16 planes hi | lo of scale * x (scale: a power of two -- exact; the ACX_PREC_F16X3 weights are lifted into fp16's normal
// range by it) instead of three bf16 planes; the third plane is not written
template <int PANEL, int F16 = 0>
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, int64_t ld, u16* __restrict__ dst,
                                                           int64_t plane, int64_t rows, int cols4, float scale = 1.f) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const int64_t r = i / cols4;
  const int c4 = (int)(i - r * cols4);
  const float4 v = *reinterpret_cast<const float4*>(src + r * ld + 4 * c4);
  const float x[4] = {v.x, v.y, v.z, v.w};
  u16 h[4], m[4], l[4];
  if constexpr (F16 != 0) {
    const int64_t o = PANEL ? ((int64_t)(c4 >> 3) * rows + r) * 32 + 4 * (c4 & 7) : r * (int64_t)cols4 * 4 + 4 * c4;
    uint2 ph, pl;
    ph.x = f2h2(x[0] * scale, x[1] * scale); ph.y = f2h2(x[2] * scale, x[3] * scale);
    pl.x = f2h2(x[0] * scale - h2f_lo(ph.x), x[1] * scale - h2f_hi(ph.x));
    pl.y = f2h2(x[2] * scale - h2f_lo(ph.y), x[3] * scale - h2f_hi(ph.y));
    *reinterpret_cast<uint2*>(dst + o) = ph;
    *reinterpret_cast<uint2*>(dst + plane + o) = pl;
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = f2bf(x[k]);
    // |x| above bf16's largest finite value (3.3895e38) rounds to bf16 infinity although x is finite: truncate instead (the
    // remainder then has <= 16 significant bits: mid + lo still hold it exactly)
    if ((h[k] & 0x7fffu) == 0x7f80u && (__float_as_uint(x[k]) & 0x7fffffffu) < 0x7f800000u) h[k] = (u16)(__float_as_uint(x[k]) >> 16);
    const float r1 = x[k] - bf2f(h[k]);
    m[k] = f2bf(r1);
    const float r2 = r1 - bf2f(m[k]);
    l[k] = f2bf(r2);
  }
  // PANEL: K-panel layout [cols / 32][rows][32] (ACX_BF16X3P)
  const int64_t o = PANEL ? ((int64_t)(c4 >> 3) * rows + r) * 32 + 4 * (c4 & 7) : r * (int64_t)cols4 * 4 + 4 * c4;
  uint2 pk;
  pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
  *reinterpret_cast<uint2*>(dst + o) = pk;
  pk.x = (uint32_t)m[0] | ((uint32_t)m[1] << 16); pk.y = (uint32_t)m[2] | ((uint32_t)m[3] << 16);
  *reinterpret_cast<uint2*>(dst + plane + o) = pk;
  pk.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16); pk.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
  *reinterpret_cast<uint2*>(dst + 2 * plane + o) = pk;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    uint2 pk;
    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(dst + i) = pk;
  } else {
    for (int64_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
  }
}

// column sums: block handles a slab of rows, thread = 4 columns; one atomicAdd per column per block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ acc,
                                                     int64_t rows, int D, int rows_per_block) {
  const int c4 = threadIdx.x % (D / 4);
  const int rsub = threadIdx.x / (D / 4), nsub = 256 / (D / 4);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rsub < nsub) {
    for (int64_t r = r0 + rsub; r < r1; r += nsub) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * D + 4 * c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    atomicAdd(acc + 4 * c4, s.x);
    atomicAdd(acc + 4 * c4 + 1, s.y);
    atomicAdd(acc + 4 * c4 + 2, s.z);
    atomicAdd(acc + 4 * c4 + 3, s.w);
  }
}


// ------------------------------------------------------------------ prompt assembly + positional embedding
// out[c, t, :] = {prefix | ctx | suffix}[c, t, :] + pos[t, :]     (coop.py:74-90, text_encoder.py:15)
__global__ __launch_bounds__(256) void prompt_embed_kernel(const float* __restrict__ prefix, const float* __restrict__ ctxv,
                                                           const float* __restrict__ suffix, const float* __restrict__ pos,
                                                           float* __restrict__ out, int64_t total4, int n_ctx, int Lc, int W,
                                                           int shared_ctx, int Lout) {
  // Lout <= Lc positions are written per class (the causal text tower is evaluated up to the last EOT only)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int W4 = W / 4;
  const int w4 = (int)(i % W4);
  const int64_t tokg = i / W4;
  const int tkn = (int)(tokg % Lout);
  const int64_t c = tokg / Lout;
  const float* src;
  if (tkn == 0) src = prefix + c * W;
  else if (tkn <= n_ctx) src = ctxv + ((shared_ctx ? 0 : c * n_ctx) + (tkn - 1)) * (int64_t)W;
  else src = suffix + (c * (Lc - 1 - n_ctx) + (tkn - 1 - n_ctx)) * (int64_t)W;
  const float4 a = *reinterpret_cast<const float4*>(src + 4 * w4);
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pos) p = *reinterpret_cast<const float4*>(pos + (int64_t)tkn * W + 4 * w4);
  *reinterpret_cast<float4*>(out + tokg * W + 4 * w4) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
}

// out[i, :] = x[idx[i], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t total4, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int W4 = W / 4;
  const int64_t r = i / W4;
  const int w4 = (int)(i - r * W4);
  *reinterpret_cast<float4*>(out + r * W + 4 * w4) = *reinterpret_cast<const float4*>(x + idx[r] * W + 4 * w4);
}


// out[r, :] = x[r, :] + p[:]  with r over n rows of width LW (positional embedding add, text_encoder.py:15)
__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                        float* __restrict__ out, int64_t total4, int64_t LW4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float4 a = reinterpret_cast<const float4*>(x)[i];
  const float4 b = reinterpret_cast<const float4*>(p)[i % LW4];
  reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// temporal-model input with concat_features: out[r, :] = [ logits[r, 0:C1] | x[r, 0:D] - nc | 0-pad ]
// (anomaly_clip.py:143,223-233)
__global__ __launch_bounds__(256) void concat_features_kernel(const float* __restrict__ logits, const float* __restrict__ x,
                                                              const float* __restrict__ nc, float* __restrict__ out,
                                                              int64_t rows, int C1, int D, int Kp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Kp) return;
  const int64_t r = i / Kp;
  const int k = (int)(i - r * Kp);
  float v = 0.f;
  if (k < C1) v = logits[r * C1 + k];
  else if (k < C1 + D) v = x[r * D + (k - C1)] - nc[k - C1];
  out[i] = v;
}

}  // namespace

extern "C" int acx_vit_patches(acx_ctx* ctx, const float* frames, void* patches, int32_t out_dtype, int32_t F,
                               int32_t R, int32_t P, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!frames || !patches) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_patches: null pointer%s");
  if (F <= 0) return ACX_OK;
  if (P % 4 || R % P) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_patches: need P%%4==0 and R%%P==0%s");
  const int g = R / P;
  const int64_t total4 = (int64_t)F * g * g * 3 * P * P / 4;
  const dim3 grid((unsigned)((total4 + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == ACX_BF16X3P || out_dtype == ACX_BF16X2P || out_dtype == ACX_F16X2P) {
    if ((3 * P * P) % 32) return acx_fail(ctx, ACX_E_BADARG, "acx_vit_patches: K-panel planes need 3 P P %% 32 == 0%s");
    if (out_dtype == ACX_F16X2P) hipLaunchKernelGGL((patches_kernel<4>), grid, block, 0, s, frames, patches, total4, R, P, g);
    else if (out_dtype == ACX_BF16X2P) hipLaunchKernelGGL((patches_kernel<3>), grid, block, 0, s, frames, patches, total4, R, P, g);
    else
    hipLaunchKernelGGL((patches_kernel<2>), grid, block, 0, s, frames, patches, total4, R, P, g);
  } else if (out_dtype == ACX_BF16) hipLaunchKernelGGL((patches_kernel<1>), grid, block, 0, s, frames, patches, total4, R, P, g);
  else hipLaunchKernelGGL((patches_kernel<0>), grid, block, 0, s, frames, patches, total4, R, P, g);
  ACX_CHECK_LAUNCH(ctx, "acx_vit_patches");
  return ACX_OK;
}

extern "C" int acx_text_directions(acx_ctx* ctx, const float* text, const float* ncentroid, float* dirs, int32_t C,
                                   int32_t D, int32_t normal_id, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!text || !ncentroid || !dirs) return acx_fail(ctx, ACX_E_BADARG, "acx_text_directions: null pointer%s");
  if (C < 2 || normal_id < 0 || normal_id >= C) return acx_fail(ctx, ACX_E_BADARG, "acx_text_directions: bad C/normal_id%s");
  hipLaunchKernelGGL(text_dirs_kernel, dim3(C - 1), dim3(256), 0, (hipStream_t)stream, text, ncentroid, dirs, D, normal_id);
  ACX_CHECK_LAUNCH(ctx, "acx_text_directions");
  return ACX_OK;
}

static inline int bn_blocks(int64_t rows) {
  int64_t nb = (rows + 255) / 256;
  return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}
extern "C" size_t acx_bn_workspace_bytes(int64_t rows, int32_t C1) {
  (void)rows;
  return (size_t)1024 * 2 * 64 * sizeof(double) * (C1 > 0 ? 1 : 1);     // [<= 1024 blocks][2][<= 64 columns] f64
}

// shared launcher: projection (+ optional fused batch statistics); returns false when the shape does not fit the MFMA kernel
static bool launch_selector_mfma(const float* x, const float* nc, const float* dirs, float* raw, int64_t rows, int D, int C1,
                                 double* part, int* nblocks, int ncu, hipStream_t s) {
  const int NT = (C1 + 15) / 16;
  const size_t lds = ((size_t)16 * NT * (D + 4) + D) * 4;
  if ((D != 64 && D != 128 && D != 256 && D != 512 && D != 768 && D != 1024) || lds > 160 * 1024) return false;
  // LDS-DMA variant: one 8-wave workgroup per CU streaming full 128-B lines (64 KB of per-wave rings + the directions)
  const size_t lds_dma = 8 * SELD_ST * 4096 + lds;
  if (NT <= 2 && (D == 512 || D == 256 || D == 128 || D == 1024 || D == 768) && lds_dma <= 160 * 1024 && rows >= 256) {
    const int64_t ng = (rows + 15) / 16;
    int64_t nbd = (ng + 7) / 8;
    if (nbd > ncu) nbd = ncu;
    *nblocks = (int)nbd;
    const dim3 dgrid((unsigned)nbd), dblock(512);
#define ACX_SELDMA(DD, N_)                                                                                                \
  do {                                                                                                                    \
    if (part) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)selector_project_dma_kernel<DD, N_, true>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);                                \
      hipLaunchKernelGGL((selector_project_dma_kernel<DD, N_, true>), dgrid, dblock, lds_dma, s, x, nc, dirs, raw, rows, C1, part); \
    } else {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)selector_project_dma_kernel<DD, N_, false>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);                                \
      hipLaunchKernelGGL((selector_project_dma_kernel<DD, N_, false>), dgrid, dblock, lds_dma, s, x, nc, dirs, raw, rows, C1, part); \
    }                                                                                                                     \
  } while (0)
#define ACX_SELDMA_D(DD) do { if (NT == 1) ACX_SELDMA(DD, 1); else ACX_SELDMA(DD, 2); } while (0)
    switch (D) {
      case 128: ACX_SELDMA_D(128); break;
      case 256: ACX_SELDMA_D(256); break;
      case 512: ACX_SELDMA_D(512); break;
      case 768: ACX_SELDMA_D(768); break;
      default: ACX_SELDMA_D(1024); break;
    }
#undef ACX_SELDMA_D
#undef ACX_SELDMA
    return true;
  }
  const int64_t ngroups = (rows + 15) / 16;
  int64_t nb = (ngroups + 3) / 4;
  const int64_t cap = 2 * (int64_t)ncu;                                 // two resident blocks per CU
  if (nb > cap) nb = cap;
  *nblocks = (int)nb;
  const dim3 grid((unsigned)nb), block(256);
#define ACX_SELM(DD, N_)                                                                                                  \
  do {                                                                                                                    \
    if (part) {                                                                                                           \
      (void)hipFuncSetAttribute((const void*)selector_project_mfma_kernel<DD, N_, true>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
      hipLaunchKernelGGL((selector_project_mfma_kernel<DD, N_, true>), grid, block, lds, s, x, nc, dirs, raw, rows, C1, part); \
    } else {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)selector_project_mfma_kernel<DD, N_, false>,                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
      hipLaunchKernelGGL((selector_project_mfma_kernel<DD, N_, false>), grid, block, lds, s, x, nc, dirs, raw, rows, C1, part); \
    }                                                                                                                     \
  } while (0)
#define ACX_SELD(DD)                                                         \
  do {                                                                       \
    switch (NT) {                                                            \
      case 1: ACX_SELM(DD, 1); break;                                        \
      case 2: ACX_SELM(DD, 2); break;                                        \
      case 3: ACX_SELM(DD, 3); break;                                        \
      default: ACX_SELM(DD, 4); break;                                       \
    }                                                                        \
  } while (0)
  switch (D) {
    case 64: ACX_SELD(64); break;
    case 128: ACX_SELD(128); break;
    case 256: ACX_SELD(256); break;
    case 512: ACX_SELD(512); break;
    case 768: ACX_SELD(768); break;
    default: ACX_SELD(1024); break;
  }
#undef ACX_SELD
#undef ACX_SELM
  return true;
}

extern "C" int acx_selector_project(acx_ctx* ctx, const float* x, const float* ncentroid, const float* dirs, float* raw,
                                    int64_t rows, int32_t D, int32_t C1, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!x || !ncentroid || !dirs || !raw) return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (C1 <= 0 || C1 > 64) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_selector_project: need 1 <= C-1 <= 64%s");
  if (D % 64 || D > 1024 || (D / 64 != 1 && D / 64 != 2 && (D / 64) % 4))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_selector_project: D must be 64/128/256/512/768/1024%s");
  if ((((uintptr_t)x | (uintptr_t)ncentroid | (uintptr_t)dirs) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project: x / ncentroid / dirs must be 16-byte aligned%s");
  hipStream_t s = (hipStream_t)stream;
  int nb_unused = 0;
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  if (launch_selector_mfma(x, ncentroid, dirs, raw, rows, D, C1, nullptr, &nb_unused, ncu, s)) {
    ACX_CHECK_LAUNCH(ctx, "acx_selector_project");
    return ACX_OK;
  }
  // (C-1) * D too large for the MFMA kernel's LDS layout: wave-per-row kernel
  const size_t lds = (size_t)C1 * D * 4;
  if (lds > 160 * 1024) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_selector_project: (C-1) * D * 4 bytes exceed the 160 KB of LDS%s");
  int64_t nb = (rows + 3) / 4;
  if (nb > 2048) nb = 2048;
  const dim3 grid((unsigned)nb), block(256);
#define ACX_SEL(V)                                                                                         \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)selector_project_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((selector_project_kernel<V>), grid, block, lds, s, x, ncentroid, dirs, raw, rows, C1); \
  } while (0)
  switch (D / 64) {
    case 1: ACX_SEL(1); break;
    case 2: ACX_SEL(2); break;
    case 4: ACX_SEL(4); break;
    case 8: ACX_SEL(8); break;
    case 12: ACX_SEL(12); break;
    default: ACX_SEL(16); break;
  }
#undef ACX_SEL
  ACX_CHECK_LAUNCH(ctx, "acx_selector_project");
  return ACX_OK;
}

// SyncBN exchange, the combine step (Chan et al.): gathered[r] = (mean[C], biased_var[C] * rows_r, rows_r) of rank r ->
// mean, biased / unbiased variance over all rows and the total row count, in ONE launch (the dozen elementwise launches
// of the torch expression sat on the critical path of every data-parallel forward).  Ranks are summed in order.
namespace {
__global__ __launch_bounds__(64) void bn_combine_kernel(const float* __restrict__ g, int R, int C, float* __restrict__ mean,
                                                        float* __restrict__ var_b, float* __restrict__ var_u,
                                                        float* __restrict__ total) {
  const int c = threadIdx.x;
  float n, m, vb, vu;
  acx_bn_combine_col(g, R, C, c < C ? c : 0, n, m, vb, vu);
  if (c == 0) total[0] = n;
  if (c >= C) return;
  mean[c] = m;
  var_b[c] = vb;
  var_u[c] = vu;
}
}  // namespace

extern "C" int acx_bn_combine(acx_ctx* ctx, const float* gathered, int32_t ranks, int32_t C1, float* mean, float* var_biased,
                              float* var_unbiased, float* total_rows, void* stream) {
  if (!gathered || !mean || !var_biased || !var_unbiased || !total_rows) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_combine: null pointer%s");
  if (ranks < 1 || C1 < 1 || C1 > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_combine: need ranks >= 1 and 1 <= C1 <= 64%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, ranks, C1, mean, var_biased,
                     var_unbiased, total_rows);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_combine");
  return ACX_OK;
}

extern "C" int acx_bn_stats(acx_ctx* ctx, const float* raw, int64_t rows, int32_t C1, float* mean, float* var_biased,
                            float* var_unbiased, void* workspace, size_t workspace_bytes, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!raw || !mean || !var_biased || !var_unbiased || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_stats: null pointer%s");
  if (rows <= 0 || C1 <= 0 || C1 > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_stats: need rows > 0 and 1 <= C1 <= 64%s");
  if (workspace_bytes < acx_bn_workspace_bytes(rows, C1) || ((uintptr_t)workspace & 7))
    return acx_fail(ctx, ACX_E_BADARG, "acx_bn_stats: workspace too small (acx_bn_workspace_bytes) or misaligned%s");
  hipStream_t s = (hipStream_t)stream;
  const int nb = bn_blocks(rows);
  const int64_t rpb = (rows + nb - 1) / nb;
  hipLaunchKernelGGL((bn_partial_kernel<0>), dim3(nb), dim3(256), 0, s, raw, (const float*)nullptr, rows, C1, 64, rpb, (double*)workspace);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, 64, C1, rows, mean, var_biased,
                     var_unbiased);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_stats");
  return ACX_OK;
}

// sums[0:C1] = sum dl, sums[C1:2C1] = sum dl * xhat  (BatchNorm1d training backward, part A)
__global__ __launch_bounds__(256) void bn_sums_finalize_kernel(const double* __restrict__ part, int nblocks, int CP, int C1,
                                                               float* __restrict__ sums) {
  __shared__ double red[256];
  const double v = bn_pair_total(part, nblocks, CP, C1, red);
  if ((int)threadIdx.x < 2 * C1) sums[threadIdx.x] = (float)v;         // pair index == k * C1 + c
}
extern "C" int acx_bn_bwd_stats(acx_ctx* ctx, const float* logits, const float* dlogits, float* sums /* [2*C1] */,
                                int64_t rows, int32_t C1, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logits || !dlogits || !sums || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_stats: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (C1 <= 0 || C1 > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_stats: need 1 <= C1 <= 64%s");
  if (workspace_bytes < acx_bn_workspace_bytes(rows, C1) || ((uintptr_t)workspace & 7))
    return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_stats: workspace too small (acx_bn_workspace_bytes) or misaligned%s");
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  const int nb = bn_blocks(rows);
  const int64_t rpb = (rows + nb - 1) / nb;
  hipLaunchKernelGGL((bn_partial_kernel<1>), dim3(nb), dim3(256), 0, s, logits, dlogits, rows, C1, 64, rpb, (double*)workspace);
  hipLaunchKernelGGL(bn_sums_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, 64, C1, sums);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_bwd_stats");
  return ACX_OK;
}

extern "C" int acx_selector_project_stats(acx_ctx* ctx, const float* x, const float* ncentroid, const float* dirs, float* raw,
                                          int64_t rows, int32_t D, int32_t C1, float* mean, float* var_biased,
                                          float* var_unbiased, void* workspace, size_t workspace_bytes, void* stream) {
  if (!mean || !var_biased || !var_unbiased || !workspace)
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project_stats: null pointer%s");
  if (rows <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project_stats: empty%s");
  if (C1 <= 0 || C1 > 64) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_selector_project_stats: need 1 <= C-1 <= 64%s");
  if (workspace_bytes < acx_bn_workspace_bytes(rows, C1) || ((uintptr_t)workspace & 7))
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project_stats: workspace too small or misaligned%s");
  if (!x || !ncentroid || !dirs || !raw || (((uintptr_t)x | (uintptr_t)ncentroid | (uintptr_t)dirs) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_project_stats: null or misaligned pointer%s");
  hipStream_t s = (hipStream_t)stream;
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  int nb = 0;
  {
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    if (!launch_selector_mfma(x, ncentroid, dirs, raw, rows, D, C1, (double*)workspace, &nb, ncu, s)) {
      // shape outside the MFMA kernel: projection, then the stand-alone statistics
      int rc = acx_selector_project(ctx, x, ncentroid, dirs, raw, rows, D, C1, stream);
      if (rc != ACX_OK) return rc;
      return acx_bn_stats(ctx, raw, rows, C1, mean, var_biased, var_unbiased, workspace, workspace_bytes, stream);
    }
  }
  AcxProfScope prof2__(ctx, ACX_K_OTHER, s);
  const int CP = 16 * ((C1 + 15) / 16);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, CP, C1, rows, mean, var_biased,
                     var_unbiased);
  ACX_CHECK_LAUNCH(ctx, "acx_selector_project_stats");
  return ACX_OK;
}

extern "C" int acx_selector_bn(acx_ctx* ctx, const float* raw, const float* mean, const float* var, float* logits,
                               int64_t ldl, int64_t rows, int32_t C1, float eps, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!raw || !mean || !var || !logits) return acx_fail(ctx, ACX_E_BADARG, "acx_selector_bn: null pointer%s");
  if (rows <= 0) return ACX_OK;
  const int64_t total = rows * C1;
  hipLaunchKernelGGL(selector_bn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw,
                     mean, var, logits, ldl, total, C1, eps);
  ACX_CHECK_LAUNCH(ctx, "acx_selector_bn");
  return ACX_OK;
}

// Axial attention forward on the f32 MFMA: one wave per (line, head) group, q / k / v in a wave-private LDS slice.
// S^T = K Q^T puts a lane's QUERY in the MFMA column, so the softmax is in-lane sums + two cross-quarter shuffles and
// the P^T tiles are the B operand of O^T = V^T P^T as they stand (B[k = j][n = i]: lane & 15 = i, MFMA step r <-> key
// j = 4 kq + r); a lane ends with four contiguous output channels of one query (one 16-byte store).  The thread-per-query
// kernel above (51 / 41 us per launch at 64 videos) keeps the shapes this one does not cover.
template <int T, int E>
__global__ __launch_bounds__(256) void axial_attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int gn,
                                                                  int gl, int heads, int axis, float scale, int64_t ngroups) {
  constexpr int ES = E + 4;
  constexpr int TT = T / 16, ET = E / 16, KS = E / 16;
  constexpr int WF = 3 * T * ES;
  extern __shared__ __attribute__((aligned(16))) char smem_af[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  float* sQ = reinterpret_cast<float*>(smem_af) + wave * WF;
  float* sK = sQ + T * ES;
  float* sV = sK + T * ES;
  const int He = heads * E, ld = 3 * He;
  const int other = axis == 0 ? gl : gn;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t line = grp / heads;
    const int h = (int)(grp - line * heads);
    const int64_t tile = line / other;
    const int o = (int)(line - tile * other);
    const int64_t row0 = axis == 0 ? tile * gn * gl + o : (tile * gn + o) * gl;
    const int64_t rstep = axis == 0 ? gl : 1;
#pragma unroll
    for (int idx = lane; idx < T * (E / 4); idx += 64) {
      const int t = idx / (E / 4), c4 = idx - t * (E / 4);
      const float* p = qkv + (row0 + t * rstep) * ld + h * E + 4 * c4;
      *reinterpret_cast<float4*>(sQ + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(sK + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p + He);
      *reinterpret_cast<float4*>(sV + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p + 2 * He);
    }
#pragma unroll
    for (int it = 0; it < TT; ++it) {
      f32x4 st[TT];                                // [jt]: S^T tiles of query tile it (rows j, column i)
#pragma unroll
      for (int jt = 0; jt < TT; ++jt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 k4 = *reinterpret_cast<const float4*>(sK + (16 * jt + li) * ES + 16 * ks + 4 * kq);
          const float4 q4 = *reinterpret_cast<const float4*>(sQ + (16 * it + li) * ES + 16 * ks + 4 * kq);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.x, q4.x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.y, q4.y, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.z, q4.z, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.w, q4.w, a, 0, 0, 0);
        }
        st[jt] = a;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[jt][r] *= scale; mx = fmaxf(mx, st[jt][r]); }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[jt][r] = __expf(st[jt][r] - mx); sum += st[jt][r]; }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      // O^T[e][i] = sum_j V[j][e] P^T[j][i]: A = V^T (lane & 15 = e, step r <-> j = 4 kq + r), B = the P^T tiles
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < TT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sV[(16 * jt + 4 * kq + r) * ES + 16 * et + li], st[jt][r], acc, 0, 0, 0);
        // lane: query i = 16 it + li, channels e = 16 et + 4 kq + (0..3)
        *reinterpret_cast<float4*>(out + (row0 + (16 * it + li) * rstep) * He + h * E + 16 * et + 4 * kq) =
            make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      }
    }
  }
}

extern "C" int acx_axial_attention(acx_ctx* ctx, const float* qkv, float* out, int32_t tiles, int32_t gn, int32_t gl,
                                   int32_t heads, int32_t e, int32_t axis, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!qkv || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_axial_attention: null pointer%s");
  if (tiles <= 0) return ACX_OK;
  const int T = axis == 0 ? gn : gl;
  if ((T == 16 || T == 32) && (e == 16 || e == 32) && heads > 0 && !(((uintptr_t)qkv | (uintptr_t)out) & 15)) {
    const int64_t ngroups = (int64_t)tiles * (axis == 0 ? gl : gn) * heads;
    const size_t ldsm = (size_t)4 * 3 * T * (e + 4) * sizeof(float);
    int64_t nbm = (ngroups + 3) / 4;
    const int64_t capm = 2 * (int64_t)(ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256);
    if (nbm > capm) nbm = capm;
    const float scale = 1.f / sqrtf((float)e);
    hipStream_t sm = (hipStream_t)stream;
#define ACX_AXF(TT_, EE_)                                                                                  \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)axial_attn_fwd_mfma_kernel<TT_, EE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm); \
    hipLaunchKernelGGL((axial_attn_fwd_mfma_kernel<TT_, EE_>), dim3((unsigned)nbm), dim3(256), ldsm, sm, qkv, out, gn, gl, heads, axis, \
                       scale, ngroups);                                                                    \
  } while (0)
    if (T == 32) { if (e == 32) ACX_AXF(32, 32); else ACX_AXF(32, 16); }
    else { if (e == 32) ACX_AXF(16, 32); else ACX_AXF(16, 16); }
#undef ACX_AXF
    ACX_CHECK_LAUNCH(ctx, "acx_axial_attention(mfma)");
    return ACX_OK;
  }
  if ((T != 16 && T != 32) || (e != 16 && e != 32) || heads <= 0 || 256 % (T * heads))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_axial_attention: need axis length in {16,32}, e in {16,32}, T*heads | 256%s");
  const int lpb = 256 / (T * heads);
  const int64_t nlines = (int64_t)tiles * (axis == 0 ? gl : gn);
  const size_t lds = (size_t)lpb * T * heads * e * 2 * 4;
  const dim3 grid((unsigned)((nlines + lpb - 1) / lpb)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define ACX_AX(TT, EE)                                                                                     \
  do {                                                                                                     \
    (void)hipFuncSetAttribute((const void*)axial_attn_kernel<TT, EE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((axial_attn_kernel<TT, EE>), grid, block, lds, s, qkv, out, tiles, gn, gl, heads, axis, nlines); \
  } while (0)
  if (T == 32 && e == 32) ACX_AX(32, 32);
  else if (T == 32) ACX_AX(32, 16);
  else if (e == 32) ACX_AX(16, 32);
  else ACX_AX(16, 16);
#undef ACX_AX
  ACX_CHECK_LAUNCH(ctx, "acx_axial_attention");
  return ACX_OK;
}

extern "C" int acx_class_probs(acx_ctx* ctx, const float* sim, const float* scores, float* probs, int64_t rows,
                               int32_t C1, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!sim || !scores || !probs) return acx_fail(ctx, ACX_E_BADARG, "acx_class_probs: null pointer%s");
  if (rows <= 0) return ACX_OK;
  hipLaunchKernelGGL(class_probs_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sim,
                     scores, probs, rows, C1);
  ACX_CHECK_LAUNCH(ctx, "acx_class_probs");
  return ACX_OK;
}

extern "C" int acx_cast_bf16(acx_ctx* ctx, const float* src, void* dst, int64_t n, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!src || !dst) return acx_fail(ctx, ACX_E_BADARG, "acx_cast_bf16: null pointer%s");
  if (n <= 0) return ACX_OK;
  const int64_t nt = (n + 3) / 4;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (u16*)dst, n);
  ACX_CHECK_LAUNCH(ctx, "acx_cast_bf16");
  return ACX_OK;
}

static int split_bf16x3_impl(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows,
                             int64_t cols, void* stream, int panel) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!src || !dst) return acx_fail(ctx, ACX_E_BADARG, "acx_split_bf16x3: null pointer%s");
  if (rows <= 0 || cols <= 0) return ACX_OK;
  if (cols % 4 || ld % 4 || ld < cols || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7) || (plane_stride_bytes & 7) ||
      plane_stride_bytes < rows * cols * 2 || (panel && cols % 32))
    return acx_fail(ctx, ACX_E_BADARG, "acx_split_bf16x3: cols / ld multiples of 4 (panel layout: cols of 32), aligned pointers, planes of >= rows * cols bf16%s");
  const int64_t n4 = rows * (cols / 4);
  if (n4 > ((int64_t)1 << 38)) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_split_bf16x3: too many elements%s");
  if (panel)
    hipLaunchKernelGGL(split_bf16x3_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, (u16*)dst,
                       plane_stride_bytes / 2, rows, (int)(cols / 4));
  else
    hipLaunchKernelGGL(split_bf16x3_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, (u16*)dst,
                       plane_stride_bytes / 2, rows, (int)(cols / 4));
  ACX_CHECK_LAUNCH(ctx, "acx_split_bf16x3");
  return ACX_OK;
}
// up to 16 dense matrices (ld == cols) in ONE launch: the head's convolution weights (forward and dX layouts of two layers =
// eight tensors) are re-split after every optimizer step -- eight 6-us launches at the head of a data-parallel rank's 2.3 ms step
struct SplitSegs { const float* src[16]; u16* dst[16]; long long n4[16]; long long blk0[17]; int n; };
__global__ __launch_bounds__(256) void split_bf16x3_multi_kernel(const SplitSegs S) {
  int k = 0;
  while (k + 1 < S.n && (long long)blockIdx.x >= S.blk0[k + 1]) ++k;
  const long long i = ((long long)blockIdx.x - S.blk0[k]) * 256 + threadIdx.x;
  if (i >= S.n4[k]) return;
  const float4 v = reinterpret_cast<const float4*>(S.src[k])[i];
  const float x[4] = {v.x, v.y, v.z, v.w};
  u16 h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = f2bf(x[e]);
    if ((h[e] & 0x7fffu) == 0x7f80u && (__float_as_uint(x[e]) & 0x7fffffffu) < 0x7f800000u) h[e] = (u16)(__float_as_uint(x[e]) >> 16);
    const float r1 = x[e] - bf2f(h[e]);
    m[e] = f2bf(r1);
    l[e] = f2bf(r1 - bf2f(m[e]));
  }
  u16* d0 = S.dst[k] + 4 * i;
  const long long plane = 4 * S.n4[k];
  *reinterpret_cast<uint2*>(d0) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
  *reinterpret_cast<uint2*>(d0 + plane) = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
  *reinterpret_cast<uint2*>(d0 + 2 * plane) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}
extern "C" int acx_split_bf16x3_multi(acx_ctx* ctx, int32_t n, const float* const* src, void* const* dst, const int64_t* numel, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (n < 0 || (n > 0 && (!src || !dst || !numel))) return acx_fail(ctx, ACX_E_BADARG, "acx_split_bf16x3_multi: null pointer%s");
  for (int base = 0; base < n; base += 16) {
    SplitSegs S;
    memset(&S, 0, sizeof(S));
    long long blocks = 0;
    int k = 0;
    for (; k < 16 && base + k < n; ++k) {
      const int64_t ne = numel[base + k];
      if (!src[base + k] || !dst[base + k] || ne <= 0 || ne % 4 || (((uintptr_t)src[base + k] | (uintptr_t)dst[base + k]) & 15) || ((ne * 2) & 15))
        return acx_fail(ctx, ACX_E_BADARG, "acx_split_bf16x3_multi: dense 16-byte aligned tensors with numel %% 8 == 0%s");
      S.src[k] = src[base + k]; S.dst[k] = (u16*)dst[base + k]; S.n4[k] = ne / 4; S.blk0[k] = blocks;
      blocks += (ne / 4 + 255) / 256;
    }
    S.blk0[k] = blocks; S.n = k;
    if (blocks > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_split_bf16x3_multi: too many elements%s");
    hipLaunchKernelGGL(split_bf16x3_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_split_bf16x3_multi");
  return ACX_OK;
}

extern "C" int acx_split_bf16x3(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows,
                                int64_t cols, void* stream) {
  return split_bf16x3_impl(ctx, src, ld, dst, plane_stride_bytes, rows, cols, stream, 0);
}
extern "C" int acx_split_bf16x3_panel(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows,
                                      int64_t cols, void* stream) {
  return split_bf16x3_impl(ctx, src, ld, dst, plane_stride_bytes, rows, cols, stream, 1);
}

// TWO fp16 planes hi | lo of scale * src (scale: a power of two; dst: 2 planes of rows * cols fp16, plane stride in bytes), row-major
// or in K-panel layout: the operands of the ACX_PREC_F16X3 products (acx_gemm_desc.a_dtype = ACX_F16)
extern "C" int acx_split_f16x2(acx_ctx* ctx, const float* src, int64_t ld, void* dst, int64_t plane_stride_bytes, int64_t rows,
                               int64_t cols, float scale, int32_t panel, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!src || !dst) return acx_fail(ctx, ACX_E_BADARG, "acx_split_f16x2: null pointer%s");
  if (rows <= 0 || cols <= 0) return ACX_OK;
  if (cols % 4 || ld % 4 || (((uintptr_t)src | (uintptr_t)dst) & 7) || (plane_stride_bytes & 7) || plane_stride_bytes < rows * cols * 2 ||
      (panel && cols % 32) || !(scale > 0.f))
    return acx_fail(ctx, ACX_E_BADARG, "acx_split_f16x2: cols / ld multiples of 4 (panel layout: cols of 32), aligned pointers, planes of >= rows * cols fp16, scale > 0%s");
  const int64_t n4 = rows * (cols / 4);
  if (panel)
    hipLaunchKernelGGL((split_bf16x3_kernel<1, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, (u16*)dst,
                       plane_stride_bytes / 2, rows, (int)(cols / 4), scale);
  else
    hipLaunchKernelGGL((split_bf16x3_kernel<0, 1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, (u16*)dst,
                       plane_stride_bytes / 2, rows, (int)(cols / 4), scale);
  ACX_CHECK_LAUNCH(ctx, "acx_split_f16x2");
  return ACX_OK;
}

extern "C" int acx_colsum(acx_ctx* ctx, const float* x, float* acc, int64_t rows, int32_t D, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!x || !acc) return acx_fail(ctx, ACX_E_BADARG, "acx_colsum: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (D % 4 || D / 4 > 256 || D <= 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_colsum: need D%%4==0 and D<=1024%s");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x, acc,
                     rows, D, rpb);
  ACX_CHECK_LAUNCH(ctx, "acx_colsum");
  return ACX_OK;
}

extern "C" int acx_prompt_embed(acx_ctx* ctx, const float* prefix, const float* ctxv, const float* suffix, const float* pos,
                                float* out, int32_t C, int32_t n_ctx, int32_t Lc, int32_t W, int32_t shared_ctx,
                                int32_t Lout, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!prefix || !ctxv || !suffix || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_prompt_embed: null pointer%s");
  if (C <= 0) return ACX_OK;
  if (Lout <= 0) Lout = Lc;
  if (W % 4 || n_ctx < 0 || n_ctx + 1 >= Lc || Lout > Lc) return acx_fail(ctx, ACX_E_BADARG, "acx_prompt_embed: bad geometry%s");
  const int64_t total4 = (int64_t)C * Lout * W / 4;
  hipLaunchKernelGGL(prompt_embed_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prefix,
                     ctxv, suffix, pos, out, total4, n_ctx, Lc, W, shared_ctx, Lout);
  ACX_CHECK_LAUNCH(ctx, "acx_prompt_embed");
  return ACX_OK;
}

extern "C" int acx_gather_rows(acx_ctx* ctx, const float* x, const int64_t* idx, float* out, int64_t n, int32_t W,
                               void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!x || !idx || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_gather_rows: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (W % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_gather_rows: W%%4%s");
  const int64_t total4 = n * W / 4;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, idx,
                     out, total4, W);
  ACX_CHECK_LAUNCH(ctx, "acx_gather_rows");
  return ACX_OK;
}

extern "C" int acx_add_bcast(acx_ctx* ctx, const float* x, const float* p, float* out, int64_t n, int64_t LW,
                             void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!x || !p || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_add_bcast: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (LW % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_add_bcast: LW%%4%s");
  const int64_t total4 = n * LW / 4;
  hipLaunchKernelGGL(add_bcast_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, p, out,
                     total4, LW / 4);
  ACX_CHECK_LAUNCH(ctx, "acx_add_bcast");
  return ACX_OK;
}

extern "C" int acx_concat_features(acx_ctx* ctx, const float* logits, const float* x, const float* ncentroid, float* out,
                                   int64_t rows, int32_t C1, int32_t D, int32_t Kp, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!logits || !x || !ncentroid || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_concat_features: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (Kp < C1 + D) return acx_fail(ctx, ACX_E_BADARG, "acx_concat_features: Kp < C1 + D%s");
  const int64_t total = rows * Kp;
  hipLaunchKernelGGL(concat_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, x, ncentroid, out, rows, C1, D, Kp);
  ACX_CHECK_LAUNCH(ctx, "acx_concat_features");
  return ACX_OK;
}

// =====================================================================================================
// Frame preprocessing: torchvision Resize(224, BICUBIC) on a PIL image ==
// PIL.Image.resize (two separable passes over 8-bit data with 22-bit fixed-point coefficients and an 8-bit
// intermediate image), CenterCrop(224), ToTensor (/255), Normalize(mean, std)
// (reference src/utils/augmentations.py:21-34, gtransforms.py:89-102,35-40,373-381,479-486).
// The coefficient tables are computed on the host exactly like PIL's precompute_coeffs/normalize_coeffs_8bpc
// (anomalyclip_amd/preprocess.py); the kernels reproduce ImagingResampleHorizontal_8bpc / Vertical_8bpc bit for bit,
// restricted to the rows/columns the centre crop keeps.  HBM-bound: one pass over the uint8 frames.
namespace {

__device__ __forceinline__ unsigned char clip8_22(int v) {
  v >>= 22;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[f][y][xo][c] = clip8(0.5 + sum_x in[f][y][xmin+x][c] * k[xo][x])   for the `ocols` kept output columns
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ tmp,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                         int64_t total, int H, int W, int ocols) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % 3);
  const int xo = (int)((i / 3) % ocols);
  const int64_t fy = i / (3 * (int64_t)ocols);          // f*H + y
  const int xmin = bounds[2 * xo], xn = bounds[2 * xo + 1];
  const unsigned char* row = in + (fy * W + xmin) * 3 + c;
  const int* k = kk + (size_t)xo * ksize;
  int ss = 1 << 21;
  for (int x = 0; x < xn; ++x) ss += (int)row[3 * x] * k[x];
  tmp[i] = clip8_22(ss);
}

// out[f][c][yo][xo] = (clip8(0.5 + sum_y tmp[f][ymin+y][xo][c] * k[yo][y]) / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, float* __restrict__ out,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk,
                                                              int ksize, int64_t total, int H, int ocols, int orows,
                                                              float m0, float m1, float m2, float s0, float s1, float s2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % ocols);
  const int yo = (int)((i / ocols) % orows);
  const int c = (int)((i / ((int64_t)ocols * orows)) % 3);
  const int64_t f = i / ((int64_t)ocols * orows * 3);
  const int ymin = bounds[2 * yo], yn = bounds[2 * yo + 1];
  const unsigned char* col = tmp + ((f * H + ymin) * ocols + xo) * 3 + c;
  const int* k = kk + (size_t)yo * ksize;
  int ss = 1 << 21;
  for (int y = 0; y < yn; ++y) ss += (int)col[(size_t)y * ocols * 3] * k[y];
  const float v = (float)clip8_22(ss) / 255.f;           // ToTensor: uint8 -> float32 / 255
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  out[i] = (v - mean) / sd;
}

// ---- fused variant: one workgroup per (frame, band of R output rows).  Phase 1 resamples the input rows the band needs
// horizontally into an 8-bit image in LDS (thread = output column: its KS coefficients live in registers; the input rows
// pass through LDS in chunks), phase 2 resamples vertically out of LDS (thread = four pixels x three
// channels = three dwords per tap) and writes the normalised f32 planes with 16-byte stores.  The uint8 intermediate
// never reaches memory; arithmetic and rounding are those of the two kernels above (bit-identical output).
// ALLIN: the band's input rows span at most FOUR chunks and every chunk's 16-byte blocks are requested up front (4 x 4
// registers per thread): the workgroup waits for HBM ONCE instead of once per chunk -- the chunk loop's arithmetic is ~0.3 us
// per chunk against ~3 us of memory latency, so with sequential chunks a workgroup spent its life waiting (240 x 320 frames:
// 168 us per 512 frames, 0.42 of the copy rate); the vertical pass's bounds / coefficients come out of LDS as well.
template <int KS, bool ALLIN>
__global__ __launch_bounds__(256) void preprocess_fused_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                               const int* __restrict__ hb, const int* __restrict__ hk,
                                                               const int* __restrict__ vb, const int* __restrict__ vk, int vks,
                                                               int H, int W, int orows, int ocols, int R, int maxrows, int cr,
                                                               float m0, float m1, float m2, float s0, float s1, float s2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ptmp[];
  __shared__ float lut[3][256];                  // (v / 255 - mean[c]) / std[c] for v = 0..255: the float tail, evaluated once
  const int f = blockIdx.y, yo0 = blockIdx.x * R, yo1 = min(orows, yo0 + R);
  const int y0 = vb[2 * yo0];
  const int nrows = min(vb[2 * (yo1 - 1)] + vb[2 * (yo1 - 1) + 1] - y0, maxrows);
  const int rowb = ocols * 3;
  const int t = threadIdx.x;
  {
    const float v = (float)t / 255.f;                                     // ToTensor: uint8 -> float32 / 255
    lut[0][t] = (v - m0) / s0; lut[1][t] = (v - m1) / s1; lut[2][t] = (v - m2) / s2;
  }
  // ---- phase 1: tmp[yl][xo][c] = clip8(0.5 + sum_x in[f][y0 + yl][xmin + x][c] * k[xo][x])
  // (24-bit multiplies: |k| < 2^23 -- a normalised bicubic weight lies in (-0.2, 1.2) x 2^22 -- and pixels are 8 bits).
  // The input rows go through LDS in chunks of `cr` whole rows: 16-byte coalesced loads of the contiguous byte range
  // (aligned down / up to 16 bytes: an aligned 16-byte block that holds a valid byte never leaves its page), then every
  // tap is three ds_read_u8 -- per-lane windows 3.2 bytes apart cost the vector-memory path one access per lane, LDS
  // serves them as broadcasts.
  unsigned char* raw = ptmp + (size_t)maxrows * rowb;                     // [cr rows] + 32 bytes of alignment slack
  const int xo = t;                                                       // ocols <= 256 (dispatch)
  int k[KS], px[KS];
  if (xo < ocols) {
    const int xmin = hb[2 * xo];
#pragma unroll
    for (int x = 0; x < KS; ++x) { k[x] = hk[(size_t)xo * KS + x]; px[x] = 3 * min(xmin + x, W - 1); }
  }
  const int roww = W * 3;
  __shared__ int svb[2 * 32], svk[32 * 15];      // the band's vertical bounds / coefficients (R <= 32 rows, <= 15 taps)
  // one chunk of input rows: `fill` puts its 16-byte blocks into `raw`, then every thread resamples its output column
  auto chunk = [&](int c0, auto&& fill) {
    const int nr = min(cr, nrows - c0);
    const size_t g0 = ((size_t)f * H + y0 + c0) * (size_t)roww;
    const size_t ga = g0 & ~(size_t)15;
    const int phase = (int)(g0 - ga);
    const int nvec = (phase + nr * roww + 15) >> 4;
    fill(nvec, ga);
    __syncthreads();
    if (xo < ocols) {
#pragma unroll 2
      for (int r = 0; r < nr; ++r) {
        const unsigned char* rp = raw + phase + r * roww;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
#pragma unroll
        for (int x = 0; x < KS; ++x) {
          a0 += __mul24((int)rp[px[x]], k[x]);
          a1 += __mul24((int)rp[px[x] + 1], k[x]);
          a2 += __mul24((int)rp[px[x] + 2], k[x]);
        }
        unsigned char* q = ptmp + (c0 + r) * rowb + 3 * xo;
        q[0] = clip8_22(a0); q[1] = clip8_22(a1); q[2] = clip8_22(a2);
      }
    }
    __syncthreads();
  };
  auto stage_v = [&]() {
    if (t < 2 * (yo1 - yo0)) svb[t] = vb[2 * yo0 + t];
    for (int i = t; i < (yo1 - yo0) * vks; i += 256) svk[i] = vk[(size_t)yo0 * vks + i];
  };
  if constexpr (ALLIN) {
    // sixteen NAMED registers (arrays handed to lambdas by reference stay in scratch memory: 272 bytes per lane measured)
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    uint4 p00 = z4, p01 = z4, p02 = z4, p03 = z4, p10 = z4, p11 = z4, p12 = z4, p13 = z4;
    uint4 p20 = z4, p21 = z4, p22 = z4, p23 = z4, p30 = z4, p31 = z4, p32 = z4, p33 = z4;
#define PPF_FETCH(c, a, b, cc_, dd)                                                                \
    if ((c) * cr < nrows) {                                                                        \
      const int nr_ = min(cr, nrows - (c) * cr);                                                   \
      const size_t g0_ = ((size_t)f * H + y0 + (c) * cr) * (size_t)roww;                           \
      const size_t ga_ = g0_ & ~(size_t)15;                                                        \
      const int nvec_ = ((int)(g0_ - ga_) + nr_ * roww + 15) >> 4;                                 \
      const uint4* src_ = reinterpret_cast<const uint4*>(in + ga_);                                \
      if (t < nvec_) a = src_[t];                                                                  \
      if (t + 256 < nvec_) b = src_[t + 256];                                                      \
      if (t + 512 < nvec_) cc_ = src_[t + 512];                                                    \
      if (t + 768 < nvec_) dd = src_[t + 768];                                                     \
    }
    PPF_FETCH(0, p00, p01, p02, p03) PPF_FETCH(1, p10, p11, p12, p13) PPF_FETCH(2, p20, p21, p22, p23) PPF_FETCH(3, p30, p31, p32, p33)
#undef PPF_FETCH
    stage_v();
#define PPF_PUT(a, b, cc_, dd)                                                                     \
    [&](int nvec, size_t) {                                                                        \
      uint4* rw = reinterpret_cast<uint4*>(raw);                                                   \
      if (t < nvec) rw[t] = a;                                                                     \
      if (t + 256 < nvec) rw[t + 256] = b;                                                         \
      if (t + 512 < nvec) rw[t + 512] = cc_;                                                       \
      if (t + 768 < nvec) rw[t + 768] = dd;                                                        \
    }
    chunk(0, PPF_PUT(p00, p01, p02, p03));
    if (cr < nrows) chunk(cr, PPF_PUT(p10, p11, p12, p13));
    if (2 * cr < nrows) chunk(2 * cr, PPF_PUT(p20, p21, p22, p23));
    if (3 * cr < nrows) chunk(3 * cr, PPF_PUT(p30, p31, p32, p33));
#undef PPF_PUT
  } else {
    stage_v();
    for (int c0 = 0; c0 < nrows; c0 += cr)
      chunk(c0, [&](int nvec, size_t ga) {
        const uint4* src = reinterpret_cast<const uint4*>(in + ga);
        for (int i = t; i < nvec; i += 256) reinterpret_cast<uint4*>(raw)[i] = src[i];
      });
  }
  // ---- phase 2: out[f][c][yo][4 g ..] = lut[c][clip8(0.5 + sum_y tmp[ymin - y0 + y][..] * k[yo][y])]
  const int ng = ocols >> 2;
  for (int id = t; id < (yo1 - yo0) * ng; id += 256) {
    const int yl = id / ng, g = id - yl * ng, yo = yo0 + yl;
    const int ymin = svb[2 * yl] - y0, yn = svb[2 * yl + 1];
    const int* kv = svk + yl * vks;
    int acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 1 << 21;
    for (int y = 0; y < yn; ++y) {
      const int row = min(ymin + y, maxrows - 1);
      const unsigned* src = reinterpret_cast<const unsigned*>(ptmp + row * rowb + 12 * g);
      const unsigned d0 = src[0], d1 = src[1], d2 = src[2];
      const int kk_ = kv[y];
      acc[0] += __mul24((int)(d0 & 255u), kk_);         acc[1] += __mul24((int)((d0 >> 8) & 255u), kk_);
      acc[2] += __mul24((int)((d0 >> 16) & 255u), kk_); acc[3] += __mul24((int)(d0 >> 24), kk_);
      acc[4] += __mul24((int)(d1 & 255u), kk_);         acc[5] += __mul24((int)((d1 >> 8) & 255u), kk_);
      acc[6] += __mul24((int)((d1 >> 16) & 255u), kk_); acc[7] += __mul24((int)(d1 >> 24), kk_);
      acc[8] += __mul24((int)(d2 & 255u), kk_);         acc[9] += __mul24((int)((d2 >> 8) & 255u), kk_);
      acc[10] += __mul24((int)((d2 >> 16) & 255u), kk_); acc[11] += __mul24((int)(d2 >> 24), kk_);
    }
    float v[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) v[e] = lut[e % 3][clip8_22(acc[e])];
    const size_t plane = (size_t)orows * ocols;
    float* o = out + (size_t)f * 3 * plane + (size_t)yo * ocols + 4 * g;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[3], v[6], v[9]);
    *reinterpret_cast<float4*>(o + plane) = make_float4(v[1], v[4], v[7], v[10]);
    *reinterpret_cast<float4*>(o + 2 * plane) = make_float4(v[2], v[5], v[8], v[11]);
  }
}

}  // namespace

extern "C" int acx_preprocess_frames(acx_ctx* ctx, const unsigned char* frames, float* out, unsigned char* tmp,
                                     const int32_t* hbounds, const int32_t* hcoef, int32_t hksize, const int32_t* vbounds,
                                     const int32_t* vcoef, int32_t vksize, int32_t F, int32_t H, int32_t W, int32_t orows,
                                     int32_t ocols, const float* mean3, const float* std3, void* stream) {
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  if (!frames || !out || !tmp || !hbounds || !hcoef || !vbounds || !vcoef || !mean3 || !std3)
    return acx_fail(ctx, ACX_E_BADARG, "acx_preprocess_frames: null pointer%s");
  if (F <= 0) return ACX_OK;
  hipStream_t s = (hipStream_t)stream;
  // fused path: odd tap counts up to 15 (downscales up to 3.5x), whole float4 output rows, the band's rows in <= 96 KB of LDS.
  // A band of R output rows needs at most (R - 1) * scale + vksize + 1 input rows, scale <= (vksize - 1) / 4.
  if (ACX_DBG_SWITCH("PREPROCESS_FUSED", true) && hksize >= 5 && hksize <= 15 && (hksize & 1) && ocols % 4 == 0 &&
      !((uintptr_t)out & 15) && F <= 65535) {
    const int R = vksize <= 9 ? 32 : 16;
    const int maxrows = (int)((R - 1) * ((vksize - 1) / 4.0) + vksize + 2);
    const int cr = std::max(1, std::min(16, 12288 / (W * 3)));              // input rows per LDS chunk (<= 12 KB)
    const size_t lds = (size_t)maxrows * ocols * 3 + (size_t)cr * W * 3 + 32;
    if (lds <= 96 * 1024 && ocols <= 256 && (maxrows * ocols * 3) % 16 == 0 && R <= 32 && vksize <= 15) {
      const dim3 grid((unsigned)((orows + R - 1) / R), (unsigned)F);
      // every chunk of a band requested up front: at most four chunks of at most 1024 16-byte blocks
      const bool allin = (maxrows + cr - 1) / cr <= 4 && (size_t)cr * W * 3 + 32 <= 4 * 256 * 16;
#define ACX_PPF(KS)                                                                                 \
  do {                                                                                              \
    static bool attr_dev_[64] = {}; int dv_ = 0; (void)hipGetDevice(&dv_); bool& done_ = attr_dev_[dv_ & 63]; \
    if (!done_) {                                                                                   \
      (void)hipFuncSetAttribute((const void*)preprocess_fused_kernel<KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
      (void)hipFuncSetAttribute((const void*)preprocess_fused_kernel<KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
      done_ = true;                                                                                 \
    }                                                                                               \
    if (allin)                                                                                      \
      hipLaunchKernelGGL((preprocess_fused_kernel<KS, true>), grid, dim3(256), lds, s, frames, out, hbounds, hcoef, vbounds, vcoef, vksize, \
                         H, W, orows, ocols, R, maxrows, cr, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);            \
    else                                                                                            \
      hipLaunchKernelGGL((preprocess_fused_kernel<KS, false>), grid, dim3(256), lds, s, frames, out, hbounds, hcoef, vbounds, vcoef, vksize, \
                         H, W, orows, ocols, R, maxrows, cr, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);            \
  } while (0)
      switch (hksize) {
        case 5: ACX_PPF(5); break;   case 7: ACX_PPF(7); break;   case 9: ACX_PPF(9); break;
        case 11: ACX_PPF(11); break; case 13: ACX_PPF(13); break; default: ACX_PPF(15); break;
      }
#undef ACX_PPF
      ACX_CHECK_LAUNCH(ctx, "acx_preprocess_frames");
      return ACX_OK;
    }
  }
  const int64_t t1 = (int64_t)F * H * ocols * 3;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, frames, tmp, hbounds, hcoef, hksize, t1,
                     H, W, ocols);
  const int64_t t2 = (int64_t)F * 3 * orows * ocols;
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, (const unsigned char*)tmp, out,
                     vbounds, vcoef, vksize, t2, H, ocols, orows, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  ACX_CHECK_LAUNCH(ctx, "acx_preprocess_frames");
  return ACX_OK;
}
