// acx_gemm -- persistent f32 "strip stream" kernel: 256-column tiles of up to 256 rows, one workgroup per CU, LDS-DMA
// (included by acx_gemm.hip inside its anonymous namespace; shares Args / helpers defined there)
// =====================================================================================================
// profiles/r02_gemm_ablation.txt: the 128x128 kernels lose ~10 % to the per-CU operand-load path (two tiles per CU pull
// 64 KB per K-step) and ~3 % to their epilogue.  Here ONE workgroup per CU owns a 256x256 tile -- the same 64 KB per
// K-step for twice the flops of two 128x128 tiles -- and the launch is a persistent stream:
//   * 16 waves (4 x 4, 64 x 64 each: 2 x 2 MFMA 32x32x2 f32 accumulators), four per SIMD, <= 128 VGPRs;
//   * operands global -> LDS by global_load_lds_dwordx4 (inline asm: see acx_attn.hip for why), K-step = 128 B per row,
//     row-linear LDS image with the bank swizzle on the SOURCE chunk (p ^ ((row >> 1) & 7)), two 64 KB stages; the
//     K-steps of consecutive tiles form one DMA stream, so a tile starts with its first K-step already in LDS;
//   * work units are 64-row x 256-column strips; a tile is 1-4 units of one column tile.  A short tile (1-3 units)
//     switches off the waves of the missing 64-row groups (each SIMD hosts the four row groups of one column group), so
//     its time shrinks with its height: the balance granularity is 64 rows, not a 256x256 tile (N = 768: 4.62 rounds of
//     256x256 tiles would run at 92 %, 18.47 units per CU run at 97 %) and no separate tail launch exists.  The tile
//     order is row-major inside an XCD (see "this CU's tiles" below);
//   * epilogue: each wave transposes its accumulator tiles through a private 4 KB XOR-swizzled slice of the stage that
//     was just consumed and goes to memory with 16-byte accesses (8 rows x 128 B per instruction); the next tile's
//     first K-step is already resident, its second is issued right after the epilogue.
constexpr int P2_BN = 256;
constexpr int P2_OP_B = 256 * 128;               // one operand image: 256 rows x 128 B = 32 KB
constexpr int P2_STAGE_B = 2 * P2_OP_B;          // A | W

typedef __attribute__((address_space(3))) void p2_lds_t;

struct P2Cursor {                                // position in this CU's tile list
  int p, end, step;                              // tile index in the XCD's row-major list (column fastest), list length, CUs
  int TN, u0, base, rem;                         // column tiles; the XCD's first row unit; block heights base + (b < rem)
  int col, ru, nun;                              // current tile: column tile, first row unit, units (1..4)
  __device__ __forceinline__ bool valid() const { return p < end; }
  __device__ __forceinline__ void load() {
    if (p < end) {
      const int b = p / TN;
      col = p - b * TN;
      ru = u0 + b * base + min(b, rem);
      nun = base + (b < rem ? 1 : 0);
    }
  }
  __device__ __forceinline__ void next() { p += step; load(); }
};

// RES != 0 is kept compilable (float4 residual per accumulator tile) but NOT dispatched: acx_gemm_takes_strip_stream()
// sends problems with a residual to gemm_f32_w8_kernel.  Round 3 also tried the residual INSIDE the K loop (four rows per
// K-step LDS-DMA'd into a per-wave strip, added to the accumulators by packed FMAs behind the K-step's barrier): 124.7 /
// 131.9 TFLOP/s on out-proj / proj against 126.4 / 132.0 for gemm_f32_w8_kernel and 133.2 / 135.8 without a residual --
// the extra 310 MB cost the same 60-100 us per launch wherever they are read, so the simpler kernel keeps those shapes.
// CV != 0: A is the implicit-GEMM operand of a 3x3 convolution over the (gn, gl) token grid (power-of-two grid, cin % 32
// == 0): K-step kt reads channels [32 (kt % (cin/32)), +32) of the row shifted by tap kt / (cin/32); taps outside the grid
// read the caller's zero page (g.zeros) -- a per-lane DMA source address, nothing else changes.
template <int ACT, int RES, int CV = 0>
__global__ __launch_bounds__(1024, 4) void gemm_f32_p256_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, hh = lane >> 5;

  // ---- this CU's tiles.  Each XCD (bid % 8, its own L2) owns a contiguous range of 64-row units, cut into row blocks of
  // 3-4 units; its tiles are listed row-major (column tile fastest) and dealt round-robin to its CUs, so the TN column
  // tiles of one row block run at the same time on TN CUs of one L2: the A rows come from HBM once, not TN times, and
  // the W panels are shared by the CUs a row block apart.  S = tiles per CU; the block count B is the largest that fits
  // S tiles per CU, the taller blocks come first (CU loads differ by at most one unit per step).
  const int TN = (d.N + P2_BN - 1) / P2_BN, RU = (d.M + 63) / 64;
  const int bid = blockIdx.x, nb = gridDim.x;
  const int xcd = bid & 7, qb = nb >> 3, rb = nb & 7;
  const int nc = qb + (xcd < rb ? 1 : 0);                                 // CUs of this XCD in the launch
  const int cb = xcd * qb + min(xcd, rb);                                 // CUs of the XCDs before it
  const int u0 = (int)((long)RU * cb / nb), un = (int)((long)RU * (cb + nc) / nb) - u0;
  if (un <= 0) return;
  int S = (int)(((long)un * TN + 4L * nc - 1) / (4L * nc));
  int B = min((int)((long)S * nc / TN), un);
  while (B < (un + 3) / 4) { ++S; B = min((int)((long)S * nc / TN), un); }
  P2Cursor cc, dc;                               // compute cursor, DMA cursor
  cc.p = bid >> 3; cc.end = B * TN; cc.step = nc; cc.TN = TN; cc.u0 = u0; cc.base = un / B; cc.rem = un % B;
  cc.col = cc.ru = 0; cc.nun = 0;
  if (!cc.valid()) return;
  cc.load();
  dc = cc;
  const int nk = d.K / 32;

  // ---- DMA: per K-step 64 wave-instructions of 1 KB (8 rows x 128 B); wave w moves rows 16 w .. 16 w + 15 of A and of W
  const unsigned lds0 = (unsigned)(uintptr_t)(p2_lds_t*)smem;
  const int drow = 16 * wave + (lane >> 3);
  const int dchunk0 = ((lane & 7) ^ ((drow >> 1) & 7)) * 4;               // floats
  const int dchunk1 = ((lane & 7) ^ (((drow + 8) >> 1) & 7)) * 4;
  int da0 = 0, da1 = 0, dw0 = 0, dw1 = 0, dma_kt = 0;
  bool dma_a = true;                                                      // this wave's A rows exist in the DMA cursor's tile
  // conv: grid coordinates of this lane's two A rows (tile base row, n, l)
  int cvr0 = 0, cvr1 = 0;                                                 // the two A rows themselves; decomposed per issue
  const int cv_kpt = CV ? d.cin / 32 : 1;                                 // K-steps per tap
  const int cv_shl = CV ? __builtin_ctz((unsigned)d.gl) : 0;
#define P2_DMA_SRC()                                                                               \
  do {                                                                                             \
    const int m0_ = dc.ru * 64, n0_ = dc.col * P2_BN;                                              \
    dma_a = 16 * wave < dc.nun * 64;                                                               \
    da0 = min(m0_ + drow, d.M - 1) * d.lda + dchunk0;                                              \
    da1 = min(m0_ + drow + 8, d.M - 1) * d.lda + dchunk1;                                          \
    if constexpr (CV != 0) {                                                                       \
      cvr0 = min(m0_ + drow, d.M - 1); cvr1 = min(m0_ + drow + 8, d.M - 1);                        \
    }                                                                                              \
    dw0 = min(n0_ + drow, d.N - 1) * d.ldw + dchunk0;                                              \
    dw1 = min(n0_ + drow + 8, d.N - 1) * d.ldw + dchunk1;                                          \
  } while (0)
#define P2_DMA1(gptr, ldsaddr)                                                                     \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
  // issues the K-step the DMA cursor points at into `stage`, then advances the cursor (across tile boundaries)
#define P2_ISSUE(stage)                                                                            \
  do {                                                                                             \
    if (dc.valid()) {                                                                              \
      const unsigned s_ = lds0 + (stage) * P2_STAGE_B + wave * 2048;                               \
      const int ko_ = dma_kt * 32;                                                                 \
      if (dma_a) {                                                                                 \
        if constexpr (CV != 0) {                                                                   \
          const int tap_ = dma_kt / cv_kpt, ci_ = (dma_kt - tap_ * cv_kpt) * 32;                   \
          const int t3_ = tap_ / 3, dn_ = t3_ - 1, dl_ = tap_ - 3 * t3_ - 1;                       \
          const int gm_ = d.gn * d.gl - 1;                                                         \
          const int cvb0 = cvr0 & ~gm_, cvb1 = cvr1 & ~gm_;                                        \
          const int nA_ = ((cvr0 & gm_) >> cv_shl) + dn_, lA_ = (cvr0 & (d.gl - 1)) + dl_;         \
          const int nB_ = ((cvr1 & gm_) >> cv_shl) + dn_, lB_ = (cvr1 & (d.gl - 1)) + dl_;         \
          const bool okA_ = (unsigned)nA_ < (unsigned)d.gn && (unsigned)lA_ < (unsigned)d.gl;      \
          const bool okB_ = (unsigned)nB_ < (unsigned)d.gn && (unsigned)lB_ < (unsigned)d.gl;      \
          const float* pA_ = okA_ ? (const float*)d.A + (size_t)(cvb0 + (nA_ << cv_shl) + lA_) * d.lda + ci_ + dchunk0 \
                                  : g.zeros + dchunk0;                                             \
          const float* pB_ = okB_ ? (const float*)d.A + (size_t)(cvb1 + (nB_ << cv_shl) + lB_) * d.lda + ci_ + dchunk1 \
                                  : g.zeros + dchunk1;                                             \
          P2_DMA1(pA_, s_);                                                                        \
          P2_DMA1(pB_, s_ + 1024);                                                                 \
        } else {                                                                                   \
          P2_DMA1((const float*)d.A + (size_t)(unsigned)(da0 + ko_), s_);                          \
          P2_DMA1((const float*)d.A + (size_t)(unsigned)(da1 + ko_), s_ + 1024);                   \
        }                                                                                          \
      }                                                                                            \
      P2_DMA1((const float*)d.W + (size_t)(unsigned)(dw0 + ko_), s_ + P2_OP_B);                    \
      P2_DMA1((const float*)d.W + (size_t)(unsigned)(dw1 + ko_), s_ + P2_OP_B + 1024);             \
      if (++dma_kt == nk) {                                                                        \
        dma_kt = 0;                                                                                \
        dc.next();                                                                                 \
        if (dc.valid()) P2_DMA_SRC();                                                              \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  // ---- fragment addresses: row (wm*64 + li [+32]) / (wn*64 + li [+32]), chunk (2 q + hh) at position chunk ^ ((li >> 1) & 7)
  const int sw = (li >> 1) & 7;
  const int a_base = (wm * 64 + li) * 128;
  const int w_base = P2_OP_B + (wn * 64 + li) * 128;
  const int o0 = ((0 + hh) ^ sw) * 16, o1 = ((2 + hh) ^ sw) * 16, o2 = ((4 + hh) ^ sw) * 16, o3 = ((6 + hh) ^ sw) * 16;
  float4 xa0, xa1, xb0, xb1;
#define P2_RD(stage, o)                                                                            \
  do {                                                                                             \
    const char* s_ = smem + (stage) * P2_STAGE_B;                                                  \
    xa0 = *reinterpret_cast<const float4*>(s_ + a_base + (o));                                     \
    xa1 = *reinterpret_cast<const float4*>(s_ + a_base + 32 * 128 + (o));                          \
    xb0 = *reinterpret_cast<const float4*>(s_ + w_base + (o));                                     \
    xb1 = *reinterpret_cast<const float4*>(s_ + w_base + 32 * 128 + (o));                          \
  } while (0)
#define P2_MM()                                                                                    \
  do {                                                                                             \
    const float a0_[4] = {xa0.x, xa0.y, xa0.z, xa0.w};                                             \
    const float a1_[4] = {xa1.x, xa1.y, xa1.z, xa1.w};                                             \
    const float b0_[4] = {xb0.x, xb0.y, xb0.z, xb0.w};                                             \
    const float b1_[4] = {xb1.x, xb1.y, xb1.z, xb1.w};                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[j], b0_[j], acc00, 0, 0, 0);                \
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[j], b1_[j], acc01, 0, 0, 0);                \
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[j], b0_[j], acc10, 0, 0, 0);                \
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[j], b1_[j], acc11, 0, 0, 0);                \
    }                                                                                              \
  } while (0)

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }

  // ---- prologue: K-steps 0 and 1 of the stream
  P2_DMA_SRC();
  P2_ISSUE(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  P2_ISSUE(1);

  // epilogue geometry: lane -> (row tr + 8 i, 4 columns at tc) of a 32 x 32 tile; wave-private 4 KB transposer, element
  // (row, col) at row * 32 + (col ^ 4 (row & 7)) floats (conflict-free ds_write_b32 and ds_read_b128)
  const bool vec = d.N % 4 == 0 && d.ldc % 4 == 0 && !((uintptr_t)d.C & 15) && !((uintptr_t)d.bias & 15) &&
                   (RES == 0 || (d.ldr % 4 == 0 && !((uintptr_t)d.residual & 15)));

  int cur = 0;
  while (cc.valid()) {
    const bool active = wm < cc.nun;                                      // this wave's 64-row group exists in the tile
    for (int kt = 0; kt < nk; ++kt) {
      if (active) {
        // one fragment set: four waves per SIMD hide the LDS round trips, a second set does not fit 128 VGPRs
        P2_RD(cur, o0); P2_MM();
        P2_RD(cur, o1); P2_MM();
        P2_RD(cur, o2); P2_MM();
        P2_RD(cur, o3); P2_MM();
      }
      // K-step s + 1 has landed for this wave; after the barrier for every wave, and stage `cur` is free
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) P2_ISSUE(cur);                                     // (at a tile end: after the epilogue below)
      cur ^= 1;
    }
    // ---- epilogue of the tile; stage cur ^ 1 (the one just consumed) serves as transposer space
    const int m0 = cc.ru * 64, n0 = cc.col * P2_BN;
    if (active) {
      // (per-lane epilogue constants re-derived per tile from an opaque copy of the lane id: hoisted out of the tile loop they
      // stay live across the K loop at the 128-register budget -- the convolution instantiations spilled one)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int tr = le >> 3, tc = 4 * (le & 7), li = le & 31, hh = le >> 5;
      float* sT = reinterpret_cast<float*>(smem + (cur ^ 1) * P2_STAGE_B) + wave * 1024;
      float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;          // loaded before the first store of the tile
      if (RES == 0 && d.bias && vec) {        // RES: loaded per accumulator tile next to the residual (register budget)
        const int c0_ = n0 + wn * 64 + tc;
        if (c0_ < d.N) bias0 = ld4(d.bias + c0_);
        if (c0_ + 32 < d.N) bias1 = ld4(d.bias + c0_ + 32);
      }
#define P2_EPI(ACC, mi, ni)                                                                        \
  do {                                                                                             \
    const int gcol = n0 + wn * 64 + (ni) * 32 + tc;                                                \
    const int grow0 = m0 + wm * 64 + (mi) * 32 + tr;                                               \
    if (vec) {                                                                                     \
      const bool gok = gcol < d.N;                                                                 \
      float4 rs[4];                                                                                \
      if constexpr (RES != 0) {                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
          const int row = min(grow0 + 8 * i, d.M - 1);                                             \
          rs[i] = ld4(d.residual + (size_t)row * d.ldr + (gok ? gcol : 0));                        \
        }                                                                                          \
      }                                                                                            \
      float4 b4 = (ni) ? bias1 : bias0;                                                            \
      if constexpr (RES != 0) { if (d.bias && gok) b4 = ld4(d.bias + gcol); }                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
        const int row_ = 4 * hh + (r & 3) + 8 * (r >> 2);                                          \
        sT[row_ * 32 + (li ^ (4 * (row_ & 7)))] = ACC[r];                                          \
        ACC[r] = 0.f;                                                                              \
      }                                                                                            \
      /* every load consumed before the first store (a store behind a pending load makes hipcc wait per store); */ \
      /* the stores themselves are never waited for here: they drain under the next tile's first K-step       */ \
      if constexpr (RES != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
        const int row_ = tr + 8 * i;                                                               \
        const float4 a4 = *reinterpret_cast<const float4*>(sT + row_ * 32 + (tc ^ (4 * (row_ & 7)))); \
        float v[4] = {a4.x + b4.x, a4.y + b4.y, a4.z + b4.z, a4.w + b4.w};                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
          if constexpr (ACT == ACX_ACT_QUICKGELU) v[e] = acx_quickgelu(v[e]);                      \
          if constexpr (ACT == ACX_ACT_LEAKYRELU) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];          \
        }                                                                                          \
        float4 ov = make_float4(v[0], v[1], v[2], v[3]);                                           \
        if constexpr (RES != 0) ov = make_float4(rs[i].x + v[0], rs[i].y + v[1], rs[i].z + v[2], rs[i].w + v[3]); \
        const int row = grow0 + 8 * i;                                                             \
        if (gok && row < d.M) *reinterpret_cast<float4*>((float*)d.C + (size_t)row * d.ldc + gcol) = ov; \
      }                                                                                            \
    } else {                                                                                       \
      const int col = n0 + wn * 64 + (ni) * 32 + li;                                               \
      const bool cok = col < d.N;                                                                  \
      const int colc = cok ? col : d.N - 1;                                                        \
      const float bias = d.bias ? d.bias[colc] : 0.f;                                              \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
        const int row = m0 + wm * 64 + (mi) * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);                \
        float v = ACC[r] + bias;                                                                   \
        if constexpr (ACT == ACX_ACT_QUICKGELU) v = acx_quickgelu(v);                              \
        if constexpr (ACT == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;                       \
        if constexpr (RES != 0) v += d.residual[(size_t)min(row, d.M - 1) * d.ldr + colc];         \
        if (cok && row < d.M) ((float*)d.C)[(size_t)row * d.ldc + col] = v;                        \
        ACC[r] = 0.f;                                                                              \
      }                                                                                            \
    }                                                                                              \
  } while (0)
      P2_EPI(acc00, 0, 0); P2_EPI(acc01, 0, 1); P2_EPI(acc10, 1, 0); P2_EPI(acc11, 1, 1);
#undef P2_EPI
    }
    cc.next();
    if (cc.valid()) {
      // the transposer slices overlap other waves' DMA destinations: everybody is done before K-step 1 of the next tile
      // is issued into this stage (its K-step 0 is already resident in the other one)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      P2_ISSUE(cur ^ 1);
    }
  }
#undef P2_MM
#undef P2_RD
#undef P2_ISSUE
#undef P2_DMA1
#undef P2_DMA_SRC
}
