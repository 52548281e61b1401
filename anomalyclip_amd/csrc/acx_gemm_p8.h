// acx_gemm -- bf16 persistent 256x256 kernel, phase-interleaved ("ping-pong") schedule
// (included by acx_gemm.hip inside its anonymous namespace after acx_gemm_bf16.h; shares Args / helpers / typedefs)
// =====================================================================================================
// gemm_bf16_ring_kernel runs its eight waves in lock step: after the K-step barrier every wave issues its DMA, then its
// first fragment reads, and the MFMA pipe idles until LDS has answered (a fifth of the K-step); the K-step ends in
// vmcnt(0).  Here (guide "256^2 8-phase template", rebuilt for this library's persistent tile stream):
//   * the wave grid is 2 (M) x 4 (N), 128 x 64 outputs per wave, v_mfma_f32_32x32x16_bf16 (acc[4][2], operands swapped:
//     the accumulator holds C^T and the epilogue stores 16-byte row pieces);
//   * the two wave GROUPS (rows 0..127 / 128..255 of the tile = waves 0..3 / 4..7 = one wave of each per SIMD) run half a
//     phase apart: a K-tile (64 k) is four phases, each { fragment ds_reads + ONE 16 KB half-tile of DMA -> counted
//     vmcnt -> barrier X -> lgkmcnt(0) -> 8 MFMAs of one 64 x 32 quadrant -> barrier Y }, group 1 enters the loop one
//     barrier late, so on every SIMD one wave issues MFMAs while the other reads LDS and issues DMA;
//   * LDS = two K-tile slots of four half-tiles (AH0 | AH1 | BH0 | BH1, 16 KB each = 128 rows x 128 B, source-side
//     bank swizzle as in the ring kernel).  The halves are cut ACROSS the wave grid -- AH0 = rows {0..63, 128..191},
//     AH1 = {64..127, 192..255}; BH0 = columns {64 q + 0..31}, BH1 = {64 q + 32..63} -- so that each half is read in ONE
//     phase (AH0 + BH0 in phase 1, BH1 in 2, AH1 in 3, none in 4) and can be restaged two phases later:
//         phase 1 of K-tile s issues BH1(s+1), phase 2 AH1(s+1), phase 3 AH0(s+2), phase 4 BH0(s+2)
//     -- every half-tile is in flight for five to six phases, and the wait of every phase is the same vmcnt(8)
//     (everything but the four newest half-tiles has landed; loads return in order), never 0 inside a tile;
//   * ordering (guide: read a staged buffer one phase AFTER the wait that retires it; restage >= 2 phases after its
//     last read): a half-tile read in phase p was retired by every wave's wait of phase p - 1 at the latest, and those
//     waits precede a barrier both groups have passed; a buffer restaged in phase p was last read in phase <= p - 2;
//   * the K-tile stream runs across output tiles (the next tile's first K-tiles land during this tile's epilogue).
//     At a tile end the groups re-align (one extra barrier for group 0), drain the DMA queue (vmcnt(0): the epilogue's
//     stores must not sit in front of counted waits), run the epilogue together and split again; the next tile's first
//     two K-tiles are complete in LDS before the epilogue (the halves phases 1 / 2 would issue go out at the tile end),
//     so the first counted wait after the stores is eight phases later.
// Measured (profiles/r02_gemm_bf16_p8.txt; M = 100864, N = 2304, f32 C): a K-tile costs 1.63 us here against 1.83 us in
// the ring kernel, i.e. 1317 TFLOP/s = 0.53 of the 2.5 PFLOP/s roof in the limit of long K (the guide's template:
// 1320-1340); what the in-model shapes (K = 768: 12 K-tiles per tile) see is the fixed cost per tile -- 14.7 us, of
// which 7.2 us are the C stores: vmcnt counts stores and DMA loads in one in-order queue, so the first counted DMA
// wait after an epilogue also waits for the 256 KB a CU just stored.  Rejected with numbers: de-synchronising the CUs
// or the XCDs by start delays (no gain: the drain is not a chip-wide burst), write-back instead of streaming stores
// (-3 %).
#ifndef ACX_P8_ABL
#define ACX_P8_ABL 0     // timing ablations (wrong results), bit mask: 1 no C stores, 2 no epilogue at all, 4 no DMA drain at the tile end
#endif
constexpr int P8_HALF_B = 128 * 128;            // half-tile: 128 rows x 64 bf16
constexpr int P8_SLOT_B = 4 * P8_HALF_B;        // K-tile slot: AH0 | AH1 | BH0 | BH1
constexpr int P8_LDS_B = 2 * P8_SLOT_B + 8 * 4096;   // + wave-private epilogue transposers

struct P8Src {                                  // DMA source of one K-tile: wave-uniform (the per-lane part is constant)
  int m0, n0;                                   // origin of its output tile
  int kt, j;                                    // K-tile inside the tile, tile ordinal of this workgroup
  int k1, pr;                                   // pairs mode: K-tile inside the operand plane pair `pr` (plain: k1 == kt, pr == 0)
  const char *ab, *wb;                          // base of the pair's A / W plane (recomputed when the pair changes, not per DMA)
};
// acx_gemm_desc.pairs == 6 (f32-accurate product from three bf16 planes per operand): a tile's K loop walks the six plane
// pairs one after the other, smallest cross term first -- pair p reads plane (P8_APLANES >> 4 p) & 15 of A and
// (P8_WPLANES >> 4 p) & 15 of W: (hi,lo) (mid,mid) (lo,hi) (hi,mid) (mid,hi) (hi,hi).  Plain problems have one pair, planes 0.
constexpr unsigned P8_APLANES = 0x010210u, P8_WPLANES = 0x001012u;

// PAIRS: compiled-in support for acx_gemm_desc.pairs = 6 (plane-pair bookkeeping of the K-tile stream); the plain instantiations
// keep the original cursor (with the bookkeeping compiled into them the f32-C variants lost 4-7 %)
template <int C_BF16, int ACT, int RES, int PAIRS = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_p8_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;      // wm = wave group
  const int li = lane & 31, hh = lane >> 5;
  const int tiles_n = (d.N + 255) / 256, tiles_m = (d.M + 255) / 256;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, qq = G >> 3, rr = G & 7;
  const int b0 = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + ((int)blockIdx.x >> 3);
  const int my_tiles = b0 < ntiles ? (ntiles - b0 + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int nk1 = d.K / 64;                     // K-tiles per plane pair
  const int npairs = PAIRS ? (d.pairs > 1 ? d.pairs : 1) : 1;
  const int nk = nk1 * npairs;                  // even (dispatch)

  // ---- DMA: a half-tile is 16 wave-instructions of 1 KB (8 rows x 128 B); wave w issues instructions 2 w, 2 w + 1
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const int hr0 = (2 * wave) * 8 + (lane >> 3), hr1 = hr0 + 8;            // half-tile rows of this lane's two chunks
  const int ch0 = ((lane & 7) ^ ((hr0 >> 1) & 7)) * 16, ch1 = ((lane & 7) ^ ((hr1 >> 1) & 7)) * 16;
  // per-lane tile rows of this lane's two 16-byte pieces in each half-tile (constants), chunk byte offsets ch0 / ch1
  const int ra00 = (hr0 >> 6) * 128 + (hr0 & 63), ra01 = (hr1 >> 6) * 128 + (hr1 & 63);              // A half 0 (+ 64: half 1)
  const int rb00 = (hr0 >> 5) * 64 + (hr0 & 31), rb01 = (hr1 >> 5) * 64 + (hr1 & 31);                // B half 0 (+ 32: half 1)
#define P8_SET_SRC(S, jj)                                                                          \
  do {                                                                                             \
    const int L_ = b0 + min((jj), my_tiles - 1) * G;   /* past the end: re-read the last tile (never consumed) */ \
    const int tm_ = L_ / tiles_n, tn_ = L_ - tm_ * tiles_n;                                        \
    S.m0 = tm_ * 256; S.n0 = tn_ * 256;                                                            \
  } while (0)
#define P8_PLANES(S)                                                                               \
  do {                                                                                             \
    S.ab = (const char*)d.A + (size_t)((P8_APLANES >> (4 * S.pr)) & 15u) * (size_t)d.a_plane_stride; \
    S.wb = (const char*)d.W + (size_t)((P8_WPLANES >> (4 * S.pr)) & 15u) * (size_t)d.w_plane_stride; \
  } while (0)
#define P8_ADVANCE(S)                                                                              \
  do {                                                                                             \
    if constexpr (PAIRS != 0) {                                                                    \
      ++S.k1;                                                                                      \
      if (++S.kt == nk) { S.kt = 0; S.k1 = 0; S.pr = 0; ++S.j; P8_SET_SRC(S, S.j); P8_PLANES(S); } \
      else if (S.k1 == nk1) { S.k1 = 0; ++S.pr; P8_PLANES(S); }                                    \
    } else {                                                                                       \
      if (++S.kt == nk) { S.kt = 0; ++S.j; P8_SET_SRC(S, S.j); }                                   \
    }                                                                                              \
  } while (0)
#define P8_GLDS(gptr, ldsaddr)                                                                     \
  do {                                                                                             \
    unsigned keep_;                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(gptr), "s"(ldsaddr) : "memory");                             \
  } while (0)
  // half-tile `half` (0: AH0, 1: AH1, 2: BH0, 3: BH1) of K-tile S into slot `slot`
#define P8_DMA(S, half, slot)                                                                      \
  do {                                                                                             \
    const unsigned l_ = lds0 + (slot) * P8_SLOT_B + (half) * P8_HALF_B + (2 * wave) * 1024;        \
    if constexpr ((half) < 2) {                                                                    \
      const int r0_ = min(S.m0 + ra00 + ((half) & 1) * 64, d.M - 1), r1_ = min(S.m0 + ra01 + ((half) & 1) * 64, d.M - 1); \
      const char* b_ = PAIRS ? S.ab + (size_t)S.k1 * 128 : (const char*)d.A + (size_t)S.kt * 128;  \
      P8_GLDS(b_ + (size_t)((unsigned)r0_ * (unsigned)d.lda * 2u + (unsigned)ch0), l_);            \
      P8_GLDS(b_ + (size_t)((unsigned)r1_ * (unsigned)d.lda * 2u + (unsigned)ch1), l_ + 1024);     \
    } else {                                                                                       \
      const int r0_ = min(S.n0 + rb00 + ((half) & 1) * 32, d.N - 1), r1_ = min(S.n0 + rb01 + ((half) & 1) * 32, d.N - 1); \
      const char* b_ = PAIRS ? S.wb + (size_t)S.k1 * 128 : (const char*)d.W + (size_t)S.kt * 128;  \
      P8_GLDS(b_ + (size_t)((unsigned)r0_ * (unsigned)d.ldw * 2u + (unsigned)ch0), l_);            \
      P8_GLDS(b_ + (size_t)((unsigned)r1_ * (unsigned)d.ldw * 2u + (unsigned)ch1), l_ + 1024);     \
    }                                                                                              \
  } while (0)

  P8Src c1, c2;                                 // K-tiles s + 1 and s + 2 of the stream
  c1.kt = 0; c1.j = 0; c1.k1 = 0; c1.pr = 0; c1.ab = c1.wb = nullptr; P8_SET_SRC(c1, 0);
  if constexpr (PAIRS != 0) P8_PLANES(c1);
  // ---- prologue: K-tile 0 entirely, AH0 / BH0 of K-tile 1
  P8_DMA(c1, 0, 0); P8_DMA(c1, 2, 0); P8_DMA(c1, 3, 0); P8_DMA(c1, 1, 0);
  P8_ADVANCE(c1);                               // c1 = K-tile 1
  P8_DMA(c1, 0, 1); P8_DMA(c1, 2, 1);
  c2 = c1; P8_ADVANCE(c2);                      // c2 = K-tile 2

  // ---- fragment addresses inside a slot: row, chunk 2 kk + hh at position chunk ^ ((li >> 1) & 7)
  const int sw = (li >> 1) & 7;
  const int fa = (wm * 64 + li) * 128;                          // + mi2 * 32 * 128 ; AH1: + P8_HALF_B
  const int fw = 2 * P8_HALF_B + (wn * 32 + li) * 128;          // BH1: + P8_HALF_B
  const int oK0 = ((0 + hh) ^ sw) * 16, oK1 = ((2 + hh) ^ sw) * 16, oK2 = ((4 + hh) ^ sw) * 16, oK3 = ((6 + hh) ^ sw) * 16;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  bf16x8 fA[2][4], fB0[4], fB1[4];              // A fragments of the current row half, B fragments of both column halves

#define P8_FRAG(slot, off) (*reinterpret_cast<const bf16x8*>(smem + (slot) * P8_SLOT_B + (off)))
#define P8_RD_A(slot, h)                                                                           \
  do {                                                                                             \
    const int b_ = fa + (h) * P8_HALF_B;                                                           \
    fA[0][0] = P8_FRAG(slot, b_ + oK0); fA[0][1] = P8_FRAG(slot, b_ + oK1);                        \
    fA[0][2] = P8_FRAG(slot, b_ + oK2); fA[0][3] = P8_FRAG(slot, b_ + oK3);                        \
    fA[1][0] = P8_FRAG(slot, b_ + 32 * 128 + oK0); fA[1][1] = P8_FRAG(slot, b_ + 32 * 128 + oK1);  \
    fA[1][2] = P8_FRAG(slot, b_ + 32 * 128 + oK2); fA[1][3] = P8_FRAG(slot, b_ + 32 * 128 + oK3);  \
  } while (0)
#define P8_RD_B(F, slot, h)                                                                        \
  do {                                                                                             \
    const int b_ = fw + (h) * P8_HALF_B;                                                           \
    F[0] = P8_FRAG(slot, b_ + oK0); F[1] = P8_FRAG(slot, b_ + oK1);                                \
    F[2] = P8_FRAG(slot, b_ + oK2); F[3] = P8_FRAG(slot, b_ + oK3);                                \
  } while (0)
  // one quadrant: rows 2 mp, 2 mp + 1 x column half ni over the K-tile (two alternating accumulators)
#define P8_MM(mp, ni, FB)                                                                          \
  do {                                                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                             \
      acc[2 * (mp)][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[kk], fA[0][kk], acc[2 * (mp)][ni], 0, 0, 0);         \
      acc[2 * (mp) + 1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[kk], fA[1][kk], acc[2 * (mp) + 1][ni], 0, 0, 0); \
    }                                                                                              \
  } while (0)
  // the tail of every phase's read section, then the MFMA section between the two barriers
#define P8_ENTER(W)                                                                                \
  do {                                                                                             \
    if (W) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
    asm volatile("" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                 \
  } while (0)
#define P8_LEAVE()                                                                                 \
  do {                                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    asm volatile("" ::: "memory");                                                                 \
  } while (0)
  // I12: phases 1 / 2 issue their half-tiles (not in the first K-tile after an epilogue: issued before it);
  // W123 / W4: the DMA wait of phases 1..3 / of phase 4
#define P8_KTILE(slot, I12, W123, W4)                                                              \
  do {                                                                                             \
    P8_RD_A(slot, 0); P8_RD_B(fB0, slot, 0); if (I12) P8_DMA(c1, 3, (slot) ^ 1);                   \
    P8_ENTER(W123); P8_MM(0, 0, fB0); P8_LEAVE();                                                  \
    P8_RD_B(fB1, slot, 1); if (I12) P8_DMA(c1, 1, (slot) ^ 1);                                     \
    P8_ENTER(W123); P8_MM(0, 1, fB1); P8_LEAVE();                                                  \
    P8_RD_A(slot, 1); P8_DMA(c2, 0, slot);                                                         \
    P8_ENTER(W123); P8_MM(1, 1, fB1); P8_LEAVE();                                                  \
    P8_DMA(c2, 2, slot);                                                                           \
    P8_ENTER(W4); P8_MM(1, 0, fB0); P8_LEAVE();                                                    \
    c1 = c2; P8_ADVANCE(c2);                                                                       \
  } while (0)

  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // AH0(0), BH0(0) landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs one barrier behind
  for (int j = 0; j < my_tiles; ++j) {
    for (int kt = 0; kt < nk; kt += 2) {
      // the first two K-tiles after an epilogue read only what landed before it (their missing halves are issued at
      // the tile end below): the first counted wait -- which also covers the epilogue's stores -- is 8 phases away
      const bool fresh = j > 0 && kt == 0;
      P8_KTILE(0, !fresh, !fresh, !fresh);
      P8_KTILE(1, true, !fresh, true);
    }
    // ---- tile end: re-align the groups, complete the K-tile after next, drain the DMA queue, epilogue, split again
    if (wm == 0) __builtin_amdgcn_s_barrier();
    P8_DMA(c1, 3, 1); P8_DMA(c1, 1, 1);                          // BH1, AH1 of the next tile's second K-tile (slot 1 is free)
    if (!(ACX_P8_ABL & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((ACX_P8_ABL & 2) && g.ksplit != 12345) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
    } else {
      const int L = b0 + j * G;
      const int tm = L / tiles_n, tn = L - tm * tiles_n;
      const int m0 = tm * 256, n0 = tn * 256;
      // each 32x32 accumulator tile goes through this wave's private 4 KB of LDS (XOR-swizzled 128-B rows) and comes
      // back row-major: lane l owns 4 consecutive columns (l & 7) of row (l >> 3) + 8 pass (see gemm_bf16_ring_kernel)
      // (the epilogue's per-lane constants are re-derived HERE, once per tile, from an opaque copy of the lane id: hoisted out of
      // the tile loop they stay live across the K loop, which runs at the full 256-register budget -- 4 of them were spilled)
      int le = lane;
      asm volatile("" : "+v"(le));
      const int li = le & 31, hh = le >> 5;
      char* scr = smem + 2 * P8_SLOT_B + wave * 4096;
      const int rl = le >> 3, cj = le & 7;
      const int col0 = n0 + wn * 64 + 4 * cj, col1 = col0 + 32;
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
              if (cok && row < d.M && (!(ACX_P8_ABL & 1) || g.ksplit == 12345)) {
                if constexpr (C_BF16 == 2) {
                  // three bf16 planes hi | mid | lo of the f32 value (ACX_BF16X3): the next pairs = 6 product's A operand
                  const float ov[4] = {v.x, v.y, v.z, v.w};
                  float r1[4], r2[4];
                  uint2 ph, pm, pl;
                  ph.x = f2bf2(ov[0], ov[1]); ph.y = f2bf2(ov[2], ov[3]);
                  r1[0] = ov[0] - __uint_as_float(ph.x << 16); r1[1] = ov[1] - __uint_as_float(ph.x & 0xffff0000u);
                  r1[2] = ov[2] - __uint_as_float(ph.y << 16); r1[3] = ov[3] - __uint_as_float(ph.y & 0xffff0000u);
                  pm.x = f2bf2(r1[0], r1[1]); pm.y = f2bf2(r1[2], r1[3]);
                  r2[0] = r1[0] - __uint_as_float(pm.x << 16); r2[1] = r1[1] - __uint_as_float(pm.x & 0xffff0000u);
                  r2[2] = r1[2] - __uint_as_float(pm.y << 16); r2[3] = r1[3] - __uint_as_float(pm.y & 0xffff0000u);
                  pl.x = f2bf2(r2[0], r2[1]); pl.y = f2bf2(r2[2], r2[3]);
                  const size_t pe = (size_t)d.M * d.ldc;
                  u16* dst = (u16*)d.C + (size_t)row * d.ldc + col;
                  __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&ph), reinterpret_cast<u32x2*>(dst));
                  __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pm), reinterpret_cast<u32x2*>(dst + pe));
                  __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pl), reinterpret_cast<u32x2*>(dst + 2 * pe));
                } else if constexpr (C_BF16 == 1) {
                  uint2 pk;
                  pk.x = f2bf2(v.x, v.y);
                  pk.y = f2bf2(v.z, v.w);
                  u32x2* dst = reinterpret_cast<u32x2*>((u16*)d.C + (size_t)row * d.ldc + col);
                  if constexpr (RES != 0) *dst = *reinterpret_cast<const u32x2*>(&pk);
                  else __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pk), dst);
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
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wm == 1) __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();                    // pairs group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no DMA may still be writing this workgroup's LDS at exit
#undef P8_KTILE
#undef P8_LEAVE
#undef P8_ENTER
#undef P8_MM
#undef P8_RD_B
#undef P8_RD_A
#undef P8_FRAG
#undef P8_DMA
#undef P8_GLDS
#undef P8_ADVANCE
#undef P8_PLANES
#undef P8_SET_SRC
}
