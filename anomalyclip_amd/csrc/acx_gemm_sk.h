// acx_gemm -- "skinny" f32 kernel for problems with FEW ROWS (the text tower of a data-parallel rank: 77 rows per class,
// 154 at two classes per GPU; clip/model.py:188-230 forward and the dX chain of its backward).
// (included by acx_gemm.hip inside its anonymous namespace; shares Args / helpers defined there)
// =====================================================================================================
// At M = 154 the tile kernels are bound by LATENCY, not by flops or bytes: a 128x128 tile walks 16-64 K-steps of
// {global load -> LDS -> barrier -> MFMA} on 8-24 CUs, then a second launch reduces its split-K partials
// (profiles/r02_dp8_rank_share_kernel_stats.csv: 15 + 5.5 us per GEMM, ~100 GEMMs per step).  Here
//   * a workgroup owns a 32x32 output tile -- 5 x N/32 = 80 ... 320 workgroups at M = 154, one per CU -- and its four
//     waves split K: wave w owns k-blocks {2w, 2w+1} of every 256-wide K chunk, ISSUES THE LDS-DMA FOR EXACTLY THAT
//     DATA ITSELF (global_load_lds_dwordx4, 16 KB per wave per chunk, bank swizzle on the source address) and waits on its
//     own vmcnt only: no workgroup barrier anywhere in the K loop;
//   * two 64 KB stages: K = 512 is resident in ONE burst (every load of the tile in flight at once, ~one HBM latency),
//     longer K streams chunk c+2 into the stage chunk c just left;
//   * the four partial 32x32 accumulators meet once, in LDS, in wave order (fixed summation order), and 256 threads
//     finish with one 16-byte store each: bias, QuickGELU, residual, or the QuickGELU derivative of a saved
//     pre-activation (dX chain: d_pre = (d_x @ W) * gelu'(pre), clip/model.py:183-185 backward);
//   * optional A prologue: QuickGELU applied to the A fragments as they leave LDS (x_next = gelu(pre) @ W^T: the
//     activation is never materialised).
constexpr int SK_CH = 256;                         // K chunk (floats): 8 k-blocks of 32
constexpr int SK_OP_B = 32 * SK_CH * 4;            // one operand image of a chunk: [8 k-blocks][32 rows][128 B] = 32 KB
constexpr int SK_STAGE_B = 2 * SK_OP_B;            // A | W
constexpr int SK_RED_F = 32 * 36;                  // per-wave accumulator image: 32 rows x (32 + 4 pad) floats
constexpr int SK_LDS_B = 2 * SK_STAGE_B + 4 * SK_RED_F * 4;

enum { SK_EPI_PLAIN = 0, SK_EPI_QUICKGELU = 1, SK_EPI_RES = 2, SK_EPI_GELUGRAD = 3 };

__device__ __forceinline__ float sk_quickgelu(float v) { return v * (1.f / (1.f + __expf(-1.702f * v))); }

template <int EPI, int A_GELU>
__global__ __launch_bounds__(256) void gemm_f32_sk_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int tiles_n = (d.N + 31) / 32;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;     // n fastest: neighbours share the A rows in L2
  const int m0 = tm * 32, n0 = tn * 32;
  const int nch = d.K / SK_CH;

  // ---- DMA: this wave's 16 KB of a chunk = k-blocks 2w, 2w+1 of A and of W, 4 row groups of 8 rows each.
  // wave-instruction (kb, rg): lane l -> row 8 rg + l/8, LDS position l%8 holds source chunk (l%8) ^ ((row >> 1) & 7)
  const unsigned lds0 = (unsigned)(uintptr_t)(p2_lds_t*)smem;
  const int drow = lane >> 3;
  unsigned aoff[4], woff[4];                                                 // element offsets of the lane's 4 row groups
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int row = 8 * rg + drow;
    const int sc = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    aoff[rg] = (unsigned)min(m0 + row, d.M - 1) * (unsigned)d.lda + sc;
    woff[rg] = (unsigned)min(n0 + row, d.N - 1) * (unsigned)d.ldw + sc;
  }
#define SK_DMA1(gptr, ldsaddr)                                                                     \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
#define SK_ISSUE(c)                                                                                \
  do {                                                                                             \
    const unsigned s_ = lds0 + ((c) & 1) * SK_STAGE_B + (2 * wave) * 4096;                         \
    const int k0_ = (c) * SK_CH + 64 * wave;                                                       \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                             \
      _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                           \
        SK_DMA1((const float*)d.A + (size_t)(aoff[rg] + k0_ + 32 * kb), s_ + kb * 4096 + rg * 1024);            \
        SK_DMA1((const float*)d.W + (size_t)(woff[rg] + k0_ + 32 * kb), s_ + SK_OP_B + kb * 4096 + rg * 1024);  \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  SK_ISSUE(0);
  if (nch > 1) SK_ISSUE(1);
  const int sw = (li >> 1) & 7;
  const int fr = (2 * wave) * 4096 + li * 128;                               // this lane's row in the wave's first k-block
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // chunk c landed; chunk c+1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* sA = smem + (c & 1) * SK_STAGE_B + fr;
    const char* sW = sA + SK_OP_B;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int o = kb * 4096 + (((2 * cc + hh) ^ sw) * 16);
        float4 xa = *reinterpret_cast<const float4*>(sA + o);
        const float4 xb = *reinterpret_cast<const float4*>(sW + o);
        if constexpr (A_GELU != 0) { xa.x = sk_quickgelu(xa.x); xa.y = sk_quickgelu(xa.y); xa.z = sk_quickgelu(xa.z); xa.w = sk_quickgelu(xa.w); }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, xb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, xb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, xb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, xb.w, acc, 0, 0, 0);
      }
    }
    if (c + 2 < nch) {
      // this wave's fragment reads of stage c&1 have all returned (their values fed the MFMAs above)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SK_ISSUE(c + 2);
    }
  }
#undef SK_ISSUE
#undef SK_DMA1
  // ---- the four K-partials meet in LDS (own region: nothing else lives there), summed in wave order
  float* red = reinterpret_cast<float*>(smem + 2 * SK_STAGE_B);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * SK_RED_F + (4 * hh + (r & 3) + 8 * (r >> 2)) * 36 + li] = acc[r];
  __syncthreads();
  const int row = t >> 3, c4 = (t & 7) * 4;
  float4 v = *reinterpret_cast<const float4*>(red + row * 36 + c4);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 p = *reinterpret_cast<const float4*>(red + w * SK_RED_F + row * 36 + c4);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  const int gm = m0 + row, gn = n0 + c4;
  if (gm >= d.M || gn >= d.N) return;
  float o[4] = {v.x, v.y, v.z, v.w};
  const bool full = gn + 3 < d.N;                                            // N % 4 == 0 is required, so always true; kept for safety
  if (d.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (full || gn + j < d.N) o[j] += d.bias[gn + j];
  }
  if constexpr (EPI == SK_EPI_QUICKGELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = sk_quickgelu(o[j]);
  }
  if constexpr (EPI == SK_EPI_RES) {
    const float4 r4 = *reinterpret_cast<const float4*>(d.residual + (size_t)gm * d.ldr + gn);
    o[0] += r4.x; o[1] += r4.y; o[2] += r4.z; o[3] += r4.w;
  }
  if constexpr (EPI == SK_EPI_GELUGRAD) {
    // y = p * sigmoid(1.702 p):  dy/dp = s * (1 + 1.702 p (1 - s))
    const float4 p4 = *reinterpret_cast<const float4*>(d.gelu_grad_of + (size_t)gm * d.ldg + gn);
    const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = 1.f / (1.f + __expf(-1.702f * pp[j]));
      o[j] *= s * (1.f + 1.702f * pp[j] * (1.f - s));
    }
  }
  *reinterpret_cast<float4*>((float*)d.C + (size_t)gm * d.ldc + gn) = make_float4(o[0], o[1], o[2], o[3]);
}
