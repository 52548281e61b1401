// acx_gemm -- "skinny" f32 kernel for problems with FEW ROWS (the text tower of a data-parallel rank: 77 rows per class,
// 154 at two classes per GPU; clip/model.py:188-230 forward and the dX chain of its backward).
// (included by acx_gemm.hip inside its anonymous namespace; shares Args / helpers defined there)
// =====================================================================================================
// At M = 154 the tile kernels are bound by LATENCY, not by flops or bytes: a 128x128 tile walks 16-64 K-steps of
// {global load -> LDS -> barrier -> MFMA} on 8-24 CUs, then a second launch reduces its split-K partials
// (profiles/r02_dp8_rank_share_kernel_stats.csv: 15 + 5.5 us per GEMM, ~100 GEMMs per step).  Here
//   * a workgroup owns a 32x32 output tile -- 5 x N/32 = 80 ... 320 workgroups at M = 154, one per CU -- and its EIGHT
//     waves split K: wave w owns k-block w (32 floats) of every 256-wide K chunk, ISSUES THE LDS-DMA FOR EXACTLY THAT
//     DATA ITSELF (global_load_lds_dwordx4, 8 KB per wave per chunk, bank swizzle on the source address) and waits on its
//     own vmcnt only: no workgroup barrier anywhere in the K loop;
//   * two 64 KB stages: K = 512 is resident in ONE burst (every load of the tile in flight at once, ~one HBM latency),
//     longer K streams chunk c+2 into the stage chunk c just left;
//   * the eight partial 32x32 accumulators meet once, in LDS, in wave order (fixed summation order), and 512 threads
//     finish with one 8-byte store each: bias, QuickGELU, residual, or the QuickGELU derivative of a saved
//     pre-activation (dX chain: d_pre = (d_x @ W) * gelu'(pre), clip/model.py:183-185 backward);
//   * optional A prologue: QuickGELU applied to the A fragments as they leave LDS (x_next = gelu(pre) @ W^T: the
//     activation is never materialised).
constexpr int SK_CH = 256;                         // K chunk (floats): 8 k-blocks of 32, one per wave
constexpr int SK_OP_B = 32 * SK_CH * 4;            // one operand image of a chunk: [8 k-blocks][32 rows][128 B] = 32 KB
constexpr int SK_STAGE_B = 2 * SK_OP_B;            // A | W
constexpr int SK_LDS_B = 2 * SK_STAGE_B;           // 128 KB: two stages; the K-partials reuse them after the loop

enum { SK_EPI_PLAIN = 0, SK_EPI_QUICKGELU = 1, SK_EPI_RES = 2, SK_EPI_GELUGRAD = 3 };

// QuickGELU x * sigmoid(1.702 x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the prologue variant runs
// it on every A fragment of every column tile
__device__ __forceinline__ float sk_sigmoid1702(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v)); }
__device__ __forceinline__ float sk_quickgelu(float v) { return v * sk_sigmoid1702(v); }

// A_NORM: LayerNorm over K applied to the A rows as they leave LDS (K == 512 = the whole row resident in the two stages):
// y = LN(x) W^T + b without a LayerNorm launch and without materialising LN(x) (clip/model.py:214-216: ln_1 -> attention in-proj,
// ln_2 -> c_fc).  Two-pass statistics (mean, then centred sum of squares) exchanged between the eight waves through LDS.
template <int EPI, int A_GELU, int A_NORM = 0>
__global__ __launch_bounds__(512) void gemm_f32_sk_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);                   // 0..7: owns k-block `wave` of every chunk
  const int li = lane & 31, hh = lane >> 5;
  const int tiles_n = (d.N + 31) / 32;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;     // n fastest: neighbours share the A rows in L2
  const int m0 = tm * 32, n0 = tn * 32;
  // cross-workgroup K split (gridDim.y pieces of `kchunk` chunks): long-K problems on few tiles -- the text tower's K = 2048
  // GEMMs at a rank's 32-224 rows are 16-112 workgroups streaming 4 MB of weights, 16-17 us each -- see the tail of the kernel
  const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
  const int c_lo = ksplit > 1 ? (int)blockIdx.y * g.kchunk : 0;
  const int nch = ksplit > 1 ? min(g.kchunk, d.K / SK_CH - c_lo) : d.K / SK_CH;

  // ---- DMA: this wave's 8 KB of a chunk = k-block `wave` of A and of W, 4 row groups of 8 rows each.
  // wave-instruction rg: lane l -> row 8 rg + l/8, LDS position l%8 holds source chunk (l%8) ^ ((row >> 1) & 7)
  const unsigned lds0 = (unsigned)(uintptr_t)(p2_lds_t*)smem;
  const int drow = lane >> 3;
  unsigned aoff[4], woff[4];                                                 // element offsets of the lane's 4 row groups
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int row = 8 * rg + drow;
    const int sc = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    aoff[rg] = (unsigned)min(m0 + row, d.M - 1) * (unsigned)d.lda + sc + 32 * wave;
    woff[rg] = (unsigned)min(n0 + row, d.N - 1) * (unsigned)d.ldw + sc + 32 * wave;
  }
#define SK_DMA1(gptr, ldsaddr)                                                                     \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
#define SK_ISSUE(c)                                                                                \
  do {                                                                                             \
    const unsigned s_ = lds0 + ((c) & 1) * SK_STAGE_B + wave * 4096;                               \
    const int k0_ = (c_lo + (c)) * SK_CH;                                                          \
    _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                             \
      SK_DMA1((const float*)d.A + (size_t)(aoff[rg] + k0_), s_ + rg * 1024);                       \
      SK_DMA1((const float*)d.W + (size_t)(woff[rg] + k0_), s_ + SK_OP_B + rg * 1024);             \
    }                                                                                              \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  SK_ISSUE(0);
  if (nch > 1) SK_ISSUE(1);
  const int sw = (li >> 1) & 7;
  const int fr = wave * 4096 + li * 128;                                     // this lane's row in the wave's k-block
  if constexpr (A_NORM != 0) {
    // K == 512, no K split: both chunks are in flight; wait for all of it, normalise the A fragments, then the MFMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 xa[2][4], xb[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int o = ((2 * cc + hh) ^ sw) * 16;
        xa[c][cc] = *reinterpret_cast<const float4*>(smem + c * SK_STAGE_B + fr + o);
        xb[c][cc] = *reinterpret_cast<const float4*>(smem + c * SK_STAGE_B + SK_OP_B + fr + o);
      }
    __shared__ float nst[2][8][32];                                          // [pass][wave][row]
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) s1 += (xa[c][cc].x + xa[c][cc].y) + (xa[c][cc].z + xa[c][cc].w);
    s1 += __shfl_xor(s1, 32, 64);                                            // the two half-rows of the wave's k-blocks
    if (hh == 0) nst[0][wave][li] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += nst[0][w][li];
    mean *= (1.f / 512.f);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        xa[c][cc].x -= mean; xa[c][cc].y -= mean; xa[c][cc].z -= mean; xa[c][cc].w -= mean;
        s2 += (xa[c][cc].x * xa[c][cc].x + xa[c][cc].y * xa[c][cc].y) + (xa[c][cc].z * xa[c][cc].z + xa[c][cc].w * xa[c][cc].w);
      }
    s2 += __shfl_xor(s2, 32, 64);
    if (hh == 0) nst[1][wave][li] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += nst[1][w][li];
    const float rstd = 1.f / sqrtf(var * (1.f / 512.f) + d.a_norm_eps);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        // the fragment at position (2 cc + hh) holds k = 256 c + 32 wave + 4 (2 cc + hh) .. + 3 of this lane's row
        const int k0 = 256 * c + 32 * wave + 4 * (2 * cc + hh);
        const float4 gw = *reinterpret_cast<const float4*>(d.a_norm_w + k0), gb = *reinterpret_cast<const float4*>(d.a_norm_b + k0);
        xa[c][cc].x = xa[c][cc].x * rstd * gw.x + gb.x; xa[c][cc].y = xa[c][cc].y * rstd * gw.y + gb.y;
        xa[c][cc].z = xa[c][cc].z * rstd * gw.z + gb.z; xa[c][cc].w = xa[c][cc].w * rstd * gw.w + gb.w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][cc].x, xb[c][cc].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][cc].y, xb[c][cc].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][cc].z, xb[c][cc].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][cc].w, xb[c][cc].w, acc, 0, 0, 0);
      }
    __syncthreads();                                                         // every wave is done reading the stages
  } else
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // chunk c landed; chunk c+1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* sA = smem + (c & 1) * SK_STAGE_B + fr;
    const char* sW = sA + SK_OP_B;
    float4 xa[4], xb[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int o = ((2 * cc + hh) ^ sw) * 16;
      xa[cc] = *reinterpret_cast<const float4*>(sA + o);
      xb[cc] = *reinterpret_cast<const float4*>(sW + o);
    }
    if (c + 2 < nch) {
      // every fragment of stage c&1 is in registers: the stage can be refilled while the MFMAs below run
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SK_ISSUE(c + 2);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if constexpr (A_GELU != 0) {
        xa[cc].x = sk_quickgelu(xa[cc].x); xa[cc].y = sk_quickgelu(xa[cc].y);
        xa[cc].z = sk_quickgelu(xa[cc].z); xa[cc].w = sk_quickgelu(xa[cc].w);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cc].x, xb[cc].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cc].y, xb[cc].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cc].z, xb[cc].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cc].w, xb[cc].w, acc, 0, 0, 0);
    }
  }
#undef SK_ISSUE
#undef SK_DMA1
  // ---- the eight K-partials meet in LDS, each wave writing into ITS OWN stage-0 region (nobody else ever touches it, and
  // its last DMA has landed), then summed in wave order
  float* red = reinterpret_cast<float*>(smem + wave * 4096);                  // [32 rows][32 cols]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(4 * hh + (r & 3) + 8 * (r >> 2)) * 32 + li] = acc[r];
  __syncthreads();
  const int row = t >> 4, c2 = (t & 15) * 2;
  const float* r0 = reinterpret_cast<const float*>(smem) + row * 32 + c2;
  float2 v = *reinterpret_cast<const float2*>(r0);
#pragma unroll
  for (int w = 1; w < 8; ++w) {
    const float2 p = *reinterpret_cast<const float2*>(r0 + w * 1024);
    v.x += p.x; v.y += p.y;
  }
  if (ksplit > 1) {
    // The pieces' partial tiles meet through memory WITHOUT a release fence (a device-scope fence writes back the whole L2 of
    // the XCD): agent-scope write-through stores, s_waitcnt vmcnt(0), a relaxed arrival counter per tile; the last piece to
    // arrive adds the pieces in piece order -- its own from registers -- and runs the epilogue.  The counter returns to zero.
    __shared__ bool last;
    const int tile = (int)blockIdx.x, ntile = (int)gridDim.x;
    float* mine = g.partial + ((size_t)blockIdx.y * ntile + tile) * 1024 + 2 * t;
    __hip_atomic_store(mine, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      ACX_HANDOFF_RELEASE();
      last = __hip_atomic_fetch_add(g.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ksplit - 1;
    }
    __syncthreads();
    if (!last) return;
    ACX_HANDOFF_ACQUIRE();
    float2 tot = make_float2(0.f, 0.f);
    for (int q = 0; q < ksplit; ++q) {
      if (q == (int)blockIdx.y) { tot.x += v.x; tot.y += v.y; continue; }
      const float* pq = g.partial + ((size_t)q * ntile + tile) * 1024 + 2 * t;
      tot.x += __hip_atomic_load(pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tot.y += __hip_atomic_load(pq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    v = tot;
    if (t == 0) __hip_atomic_store(g.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int gm = m0 + row, gn = n0 + c2;
  if (gm >= d.M || gn >= d.N) return;                                        // N % 4 == 0: a pair is in range or not
  float o[2] = {v.x, v.y};
  if (d.bias) { o[0] += d.bias[gn]; o[1] += d.bias[gn + 1]; }
  if constexpr (EPI == SK_EPI_QUICKGELU) { o[0] = sk_quickgelu(o[0]); o[1] = sk_quickgelu(o[1]); }
  if constexpr (EPI == SK_EPI_RES) {
    const float2 r2 = *reinterpret_cast<const float2*>(d.residual + (size_t)gm * d.ldr + gn);
    o[0] += r2.x; o[1] += r2.y;
  }
  if constexpr (EPI == SK_EPI_GELUGRAD) {
    // y = p * sigmoid(1.702 p):  dy/dp = s * (1 + 1.702 p (1 - s))
    const float2 p2 = *reinterpret_cast<const float2*>(d.gelu_grad_of + (size_t)gm * d.ldg + gn);
    const float pp[2] = {p2.x, p2.y};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float sg = sk_sigmoid1702(pp[j]);
      o[j] *= sg * (1.f + 1.702f * pp[j] * (1.f - sg));
    }
  }
  *reinterpret_cast<float2*>((float*)d.C + (size_t)gm * d.ldc + gn) = make_float2(o[0], o[1]);
}
