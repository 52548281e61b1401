// acx_gemm -- bf16 kernels with LDS-DMA staging (128x128 two-blocks-per-CU and persistent 256x256)
// (included by acx_gemm.hip inside its anonymous namespace; shares Args / tile constants / helpers defined there)
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

// CV = 1: implicit-GEMM 3x3 convolution over the (gn, gl) token grid (K = 9 cin, W laid out [N][tap][cin], cin % 64 == 0 so a
// K-step lies inside one tap): the DMA source of an A row becomes the row SHIFTED by the K-step's tap, or the caller's zero
// page when the tap falls outside the grid -- no im2col buffer, no LDS-side predicate.  (The XD-Violence head's bf16
// convolutions ran on the register-staged generic kernel at 435-520 TFLOP/s.)
template <int C_BF16, int ACT, int RES, int CV = 0>
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
  // conv: grid coordinates of this lane's four A rows (tiles hold whole rows of the token grid: M % 128 == 0)
  int cn0 = 0, cn1 = 0, cn2 = 0, cn3 = 0, cl0 = 0, cl1 = 0, cl2 = 0, cl3 = 0;
  const char* zsrc = nullptr;
  int kt_per_tap = 1;
  if constexpr (CV != 0) {
    const int grid_sz = d.gn * d.gl;
#define DMA_CRD(j)                                                                                 \
  do {                                                                                             \
    const int rem_ = (m0 + (4 * wave + (j)) * 8 + (lane >> 3)) % grid_sz;                          \
    cn##j = rem_ / d.gl; cl##j = rem_ - cn##j * d.gl;                                              \
  } while (0)
    DMA_CRD(0); DMA_CRD(1); DMA_CRD(2); DMA_CRD(3);
#undef DMA_CRD
    zsrc = (const char*)g.zeros + (lane & 7) * 16;
    kt_per_tap = d.cin / 64;
  }
  const int dma_off = 4 * wave * 1024;             // this wave's 4 KB slice of each operand image
  // A source of instruction j at K-step kt_: identity -> row + kt_ * 128 B; conv -> the row shifted by the tap, channel block kc_
#define DMA_ASRC(j, kt_, tap_, kc_, dn_, dl_)                                                      \
  (CV == 0 ? ga##j + (size_t)(kt_) * 128                                                           \
           : (((unsigned)(cn##j + (dn_)) < (unsigned)d.gn && (unsigned)(cl##j + (dl_)) < (unsigned)d.gl)                \
                  ? ga##j + ((ptrdiff_t)((dn_) * d.gl + (dl_)) * d.lda) * 2 + (size_t)(kc_) * 128 : zsrc))
#define DMA_ISSUE(stage, kt_)                                                                      \
  do {                                                                                             \
    char* sA_ = smem + (stage) * DMA_STAGE_B + dma_off;                                            \
    char* sW_ = sA_ + DMA_OP_B;                                                                    \
    const size_t ko_ = (size_t)(kt_) * 128;                                                        \
    const int tap_ = CV ? (kt_) / kt_per_tap : 0, kc_ = CV ? (kt_) - tap_ * kt_per_tap : 0;        \
    const int dn_ = CV ? tap_ / 3 - 1 : 0, dl_ = CV ? tap_ - (tap_ / 3) * 3 - 1 : 0;               \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(DMA_ASRC(0, kt_, tap_, kc_, dn_, dl_)), (lds_void_t*)(sA_), 16, 0, 0);       \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(DMA_ASRC(1, kt_, tap_, kc_, dn_, dl_)), (lds_void_t*)(sA_ + 1024), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(DMA_ASRC(2, kt_, tap_, kc_, dn_, dl_)), (lds_void_t*)(sA_ + 2048), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(DMA_ASRC(3, kt_, tap_, kc_, dn_, dl_)), (lds_void_t*)(sA_ + 3072), 16, 0, 0); \
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
#undef DMA_ASRC

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
        if constexpr (ACT == ACX_ACT_QUICKGELU) v = acx_quickgelu(v);
        if constexpr (ACT == ACX_ACT_LEAKYRELU) v = v > 0.f ? v : 0.01f * v;
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
              v.x = acx_quickgelu(v.x); v.y = acx_quickgelu(v.y);
              v.z = acx_quickgelu(v.z); v.w = acx_quickgelu(v.w);
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

