// acx_gemm -- f32-ACCURATE product from three bf16 planes per operand (acx_gemm_desc.pairs = 6): persistent 256x256 kernel,
// ONE wave per SIMD, plane-reuse schedule
// (included by acx_gemm.hip inside its anonymous namespace after acx_gemm_p8.h; shares Args / helpers / typedefs)
// =====================================================================================================
// gemm_bf16_p8_kernel runs a pairs = 6 product as six plain bf16 products one after the other: every plane pair streams
// its A and W K-tiles through LDS again (12 operand tiles per 64 k for 6 products) and every MFMA is fed by 1.5 fragment
// reads.  Measured there: 1.63 us per K-tile, mfma_pipe_busy 0.58, 3.1 x the algorithmic HBM traffic.  Here the SIX products
// of one k range are formed while its SIX plane tiles are resident -- each plane tile is staged ONCE and each fragment read
// feeds three MFMAs:
//   * 4 waves (2 x 2), 128 x 128 outputs per wave = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16 (256 registers), one
//     wave per SIMD with the 512-register budget: two fragment sets of 16 fragments (128 registers), no spills;
//   * a K-step is 32 k.  LDS holds eight 16 KB "units" (one plane tile: 256 rows x 64 B, source-side bank swizzle
//     chunk ^ ((row >> 2) & 3)) + 4 KB of epilogue transposer per wave = 144 KB.  A K-step is two HALF-STEPS:
//         X:  A.hi  x  W.lo, W.mid, W.hi                      (units AH WL WM WH)
//         Y:  A.lo  x  W.hi ;  A.mid  x  W.mid, W.hi           (units AL AM WM WH)
//     -- 96 MFMAs per wave per half-step (two k16 substeps x three products x 16 accumulators) behind ONE barrier, fed by
//     32 ds_read_b128 and 12 LDS-DMA instructions per wave, all of them interleaved one by one between the MFMAs of a
//     fragment set they do not touch (hand-placed: every group is fenced by sched_barrier(0));
//   * unit slots: W.hi / W.mid alternate between two slots by K-step parity (they live for a whole K-step and are
//     restaged a K-step ahead); A.hi, W.lo, A.mid, A.lo own one slot each and are restaged right after the barrier that
//     ends their half-step, one half-step before they are read again:
//         after the barrier of X(t):  DMA A.mid(t), A.lo(t), W.hi(t+1)      wait before Y(t):   vmcnt(4)
//         after the barrier of Y(t):  DMA W.mid(t+1), A.hi(t+1), W.lo(t+1)  wait before X(t+1): vmcnt(0)
//     (a unit is 4 instructions per wave; loads return in order);
//   * the K-step stream runs across the workgroup's work items (persistent; XCD-aware item order as in the other
//     persistent kernels); an item is an output tile, or one K range of a tile when the launch splits K (few tiles: the
//     head's convolutions at a data-parallel rank's rows) -- split items store raw f32 partial tiles, the epilogue is
//     splitk_reduce_kernel's;
//   * CONV: implicit-GEMM 3x3 convolution over the (gn, gl) token grid (power-of-two grid, cin % 32 == 0): the DMA source
//     of an A row is the row shifted by the K-step's tap, or the caller's zero page outside the grid.
// Accumulation order differs from the p8 kernel's (all six products of a k range before the next range, smallest cross
// terms first inside a range); the result is an f32 dot product's either way (tests hold both to the same bounds).
#ifndef ACX_X6_ABL
#define ACX_X6_ABL 0     // timing ablations (wrong results), bit mask: 1 no DMA in the K loop, 2 no vmcnt waits, 4 no epilogue stores, 8 every ds_read from one address
#endif
constexpr int X6_UNIT_B = 256 * 64;              // one plane tile: 256 rows x 32 bf16
constexpr int X6_LDS_B = 8 * X6_UNIT_B + 4 * 4096;
enum { X6_WH0 = 0, X6_WM0 = 1, X6_AH = 2, X6_WL = 3, X6_AM = 4, X6_AL = 5, X6_WH1 = 6, X6_WM1 = 7 };

struct X6Src {                                   // one K-step of the stream (wave-uniform)
  int m0, n0;                                    // origin of its output tile
  int kk;                                        // K-step (32 k) inside the operand rows
  int tk, j;                                     // step inside the item, item ordinal of this workgroup
  int steps;                                     // K-steps of the item (the last K range of a tile may be shorter)
  int dn, dl, c0;                                // TN: tap shift of the tile's B rows (conv) and its first input channel
};

typedef short x6_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) x6_s16x4 x6_lds_s16x4;
// TN fragment: the LDS image is [32 k][256 channels] (k-major), the MFMA wants 8 consecutive k of ONE channel per lane:
// two hardware transpose reads (ds_read_b64_tr_b16: every 16 lanes read a [4 k][16 channel] block, lane c of the 16 gets
// its channel's 4 k values) -- k = 8 hh .. + 3 and + 4 .. + 7.  tools/probes/tr16_probe.hip checks this address map.
__device__ __forceinline__ bf16x8 x6_tr_frag(const char* smem, int off_lo, int off_hi) {
  const x6_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x6_lds_s16x4*)(smem + off_lo));
  const x6_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x6_lds_s16x4*)(smem + off_hi));
  typedef short s16x8_ __attribute__((ext_vector_type(8)));
  const s16x8_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// TN != 0: C[n1, n2] = sum_m A[m, n1] B[m, n2] (a weight gradient: the reduction runs over ROWS).  d.M / d.N = channel counts
// N1 / N2 (output rows / columns, multiples of 256), d.K = rows, d.lda / d.ldw = row strides of the A / B planes; CONV: B's row
// for reduction index m is the token row shifted by the tap of the tile's column block (d.cin % 256 == 0: a 256-column tile
// lies inside one tap), zero page outside the grid and behind the last row.  A unit is then 32 rows x 256 channels (k-major,
// 512 B per row, its eight 64-byte chunks XOR-swizzled by row & 7); the product structure, the slots and the waits are those
// of the NT kernel.  Output: raw f32 (ksplit > 1: partial tiles for tn_reduce_kernel).
// PLAIN != 0: an ordinary bf16 product (one plane per operand: acx_gemm with prec = ACX_PREC_BF16, bf16 A, pairs <= 1) on the same
// one-wave-per-SIMD frame -- work items, LDS-DMA units, fragments and the epilogue are shared; the K loop walks SUPER-STEPS of 64 k:
// four units (A and W of two 32-wide K-steps) per super-step, 64 MFMAs per wave behind one barrier in four blocks of 16, the
// fragments of block b + 1 read between the MFMAs of block b, the next super-step's four units requested between the MFMAs of the
// first block (two unit parities in the eight LDS slots).  K % 64 == 0.
// NI (4 / 2 / 1; not PLAIN): column blocks of 32 per wave -- the workgroup's tile is 256 rows x 64 NI columns (the wave grid stays
// 2 x 2: 128 x 32 NI per wave); the launch's tiles are numbered row-major over that geometry, g.tile0 / g.ntiles select a range
// of them.  Uses: (a) the column STRIPS of a partly filled last round of 256 x 256 tiles (acx_gemm): every accumulator sees exactly
// the MFMA sequence it sees inside a full tile (same K order, same product order), so a row's result does not depend on which
// instantiation produced it -- bit-identical -- while the last round's cost falls with the strip width; (b) narrow outputs
// (N = 128: the XD-Violence head's E = 128 convolutions) without a half-empty tile; (c) TN: a 128-column tile lies inside one
// tap at cin = 128.  NT: W units shrink with the tile (NI DMA instructions per wave and unit: waits vmcnt(NI) before Y); TN units
// stay [32 k][256 channels], the channels behind the tile come from the zero page.
// W14 (TN only): wave grid 1 x 4 -- the tile is 128 rows x 128 NI columns (N1 = 128: the weight gradient of a convolution with
// 128 output channels), the upper half of the A units from the zero page.
// one MFMA of the plane products: bf16 planes (F16 = 0) or fp16 planes (F16 = 1: the two-plane fp16 split of X3 = 2)
template <int F16>
__device__ __forceinline__ f32x16 x6_mfma(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (F16 != 0) {
    typedef _Float16 x6_h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x6_h8, a), __builtin_bit_cast(x6_h8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
}

// X3 = 2: the same three-product schedule on TWO fp16 planes per operand (x = hi + lo, hi = fp16(x), lo = fp16(x - hi): 11 + 11
// significant bits + the sign trick of round-to-nearest = x to 2^-24 while lo is a normal number): (lo, hi) (hi, lo) (hi, hi) on
// v_mfma_f32_32x32x16_f16, every fp16 x fp16 product exact in f32, the dropped (lo, lo) term <= 2^-22 of the leading one -- an error of
// ~2^-22 of sum |a||w|, the f32 MFMA kernel's level, for operands inside fp16's range (|x| < 65504; elements below ~2^-3 carry an
// ABSOLUTE error of <= 2^-25: rows that are tiny as a whole lose relative accuracy).  Plane outputs are fp16 pairs as well.
// X3 (acx_gemm_desc.pairs = 3; identity rows, NT): the THREE leading cross products only -- (A.mid, W.hi) (A.hi, W.mid) (A.hi, W.hi):
// every bf16 x bf16 product exact, the dropped terms <= 2^-16 of the leading one (an error of ~1e-5 of sum |a||w|: sixteen significant
// bits per operand, between TF32's ten and f32's twenty-four) -- on the same frame: a K-step is ONE half-step with the Y half-step's
// fragment / MFMA pattern (its "A.mid" is A.hi, its "A.lo" is A.mid), all four units (A.hi, A.mid, W.hi, W.mid) double-buffered by
// K-step parity in the eight slots (A.hi: AH / AM, A.mid: WL / AL) and restaged a K-step ahead: 96 MFMAs per wave and K-step behind
// one barrier, 16 LDS-DMA instructions between them.  The lo planes are never read.
template <int C_MODE, int ACT, int RES, int CONV, int TN = 0, int PLAIN = 0, int NI = 4, int W14 = 0, int X3 = 0>
__global__ __launch_bounds__(256, 1) void gemm_x6_p4_kernel(const Args g) {
  static_assert(X3 == 0 || (CONV == 0 && TN == 0 && PLAIN == 0 && W14 == 0), "three-product mode: identity rows, NT");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const acx_gemm_desc& d = g.d;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = W14 ? 0 : wave >> 1, wn = W14 ? wave : wave & 1;
  const int li = lane & 31, hh = lane >> 5;
  static_assert(NI == 4 || ((NI == 2 || NI == 1) && PLAIN == 0), "narrow tiles: not for the PLAIN schedule");
  static_assert(W14 == 0 || (TN != 0 && NI == 2), "1 x 4 wave grid: TN products, 128 x 256 tiles");
  constexpr int TH = W14 ? 128 : 256, TW = W14 ? 128 * NI : 64 * NI;      // tile height / width
  const int tiles_n = (d.N + TW - 1) / TW, tiles_m = (d.M + TH - 1) / TH;
  const int ksplit = g.ksplit > 1 ? g.ksplit : 1;
  const int ntiles = g.ntiles > 0 ? g.ntiles : tiles_m * tiles_n;   // tiles of this launch: g.tile0 .. g.tile0 + ntiles (row-major over TH x TW tiles)
  const int nitems = ntiles * ksplit;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, qq = G >> 3, rr = G & 7;
  const int b0 = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + ((int)blockIdx.x >> 3);
  const int my_items = b0 < nitems ? (nitems - b0 + G - 1) / G : 0;
  if (my_items == 0) return;
  const int nks = PLAIN ? d.K / 64 : (d.K + 31) / 32;   // K-steps of a tile (TN: the last one may be ragged: zero page; PLAIN: super-steps)
  const int spi = (nks + ksplit - 1) / ksplit;   // K-steps per item; the last K range of a tile takes what is left (dispatch: > 0)

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  // ---- DMA: a unit is 16 wave-instructions of 1 KB (16 rows x 64 B); wave w issues instructions 4 w .. 4 w + 3.
  // lane -> row (lane >> 2) of the instruction's 16, LDS chunk position lane & 3 = global chunk ^ ((row >> 2) & 3)
  const int dr = 64 * wave + (lane >> 2);                        // + 16 i: this lane's unit row of instruction i
  const int dwr = 16 * NI * wave + (lane >> 2);                  // the same for a W unit (64 NI rows: NI instructions per wave)
  const int dc = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;          // byte offset of its 16-byte piece inside the row's 64 B
  const char* zsrc = (CONV || TN) ? (const char*)g.zeros + (lane & 3) * 16 : nullptr;
  const int sh_gl = CONV ? __builtin_ctz((unsigned)d.gl) : 0;
  const int steps_per_tap = (CONV && !TN) ? d.cin / 32 : 1;
  // K-panel layout of the planes (acx_gemm_desc.panels; identity rows): plane = [K / 32][rows][32], a unit is one contiguous
  // 16 KB block -- the byte distance between K-steps is a whole panel (plane bytes / panels), rows are 64 B apart
  const bool apanel = !CONV && !TN && (d.panels & 1), wpanel = !TN && (d.panels & 2);
  const size_t a_kstride = apanel ? (size_t)d.a_plane_stride / (size_t)(d.K / 32) : (size_t)64;
  const size_t w_kstride = wpanel ? (size_t)d.w_plane_stride / (size_t)(d.K / 32) : (size_t)64;
  const bool cpanel = C_MODE == 2 && (d.c_dtype == ACX_BF16X3P || d.c_dtype == ACX_F16X2P);
  // TN: byte offset of this lane's 16-byte piece of instruction i inside the row's 512-byte tile slice (source-side swizzle)
#define X6_TCH(i) ((((((lane & 31) >> 2) ^ ((2 * (i) + (lane >> 5)) & 7)) << 2) | (lane & 3)) * 16)
#define X6_SET_ITEM(S, jj)                                                                         \
  do {                                                                                             \
    const int L_ = b0 + min((jj), my_items - 1) * G;   /* past the end: re-read the last item (never consumed) */ \
    const int ti_ = L_ / ksplit, ks_ = L_ - ti_ * ksplit;                                          \
    const int tile_ = g.tile0 + ti_;                                                               \
    const int tm_ = tile_ / tiles_n, tn_ = tile_ - tm_ * tiles_n;                                  \
    S.m0 = tm_ * TH; S.n0 = tn_ * TW;                                                              \
    S.kk = ks_ * spi; S.steps = min(spi, nks - ks_ * spi);                                         \
    S.dn = S.dl = 0; S.c0 = S.n0;                                                                  \
    if constexpr (TN != 0 && CONV != 0) {                                                          \
      const int tap_ = S.n0 / d.cin, t3_ = tap_ / 3;                                               \
      S.c0 = S.n0 - tap_ * d.cin; S.dn = t3_ - 1; S.dl = tap_ - 3 * t3_ - 1;                       \
    }                                                                                              \
  } while (0)
  // per-lane byte offsets of this lane's four A rows / W rows of item S (the same for every plane and K-step), and -- CONV --
  // the taps that stay inside the token grid for each of the A rows (bit tap of VM[i])
#define X6_ROWS(S, RA, RW, VM)                                                                     \
  do {                                                                                             \
    if constexpr (TN == 0)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
      const int m_ = S.m0 + dr + 16 * i;                                                           \
      RA[i] = (unsigned)(CONV ? m_ : min(m_, d.M - 1)) * (apanel ? 64u : (unsigned)d.lda * 2u) + (unsigned)dc; \
      RW[i] = (unsigned)min(S.n0 + dwr + 16 * (i < NI ? i : 0), d.N - 1) * (wpanel ? 64u : (unsigned)d.ldw * 2u) + (unsigned)dc; \
      if constexpr (CONV != 0) {                                                                   \
        const int n_ = (m_ >> sh_gl) & (d.gn - 1), l_c = m_ & (d.gl - 1);                          \
        const unsigned rn_ = (n_ > 0 ? 1u : 0u) | 2u | (n_ < d.gn - 1 ? 4u : 0u);   /* dn = -1, 0, +1 */ \
        const unsigned rl_ = (l_c > 0 ? 1u : 0u) | 2u | (l_c < d.gl - 1 ? 4u : 0u); /* dl = -1, 0, +1 */ \
        VM[i] = ((rn_ & 1u) ? rl_ : 0u) | ((rn_ & 2u) ? rl_ << 3 : 0u) | ((rn_ & 4u) ? rl_ << 6 : 0u); \
      }                                                                                            \
    }                                                                                              \
  } while (0)
#define X6_ADVANCE(S, RA, RW, VM)                                                                  \
  do {                                                                                             \
    ++S.kk;                                                                                        \
    if (++S.tk == S.steps) { S.tk = 0; ++S.j; X6_SET_ITEM(S, S.j); X6_ROWS(S, RA, RW, VM); }       \
  } while (0)
  // One LDS-DMA instruction: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to LDS [m0 ..+1 KB).  M0 is not
  // restored: nothing else in this kernel reads it (ds_read / ds_write do not use M0 on gfx9+).
#define X6_GLDS_S(base, voff, ldsaddr)                                                             \
  do {                                                                                             \
    if ((ACX_X6_ABL & 1) && g.ksplit != 12345) break;                                              \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"                   \
                 : : "v"(voff), "s"(ldsaddr), "s"(base) : "memory");                               \
  } while (0)
#define X6_GLDS_V(gptr, ldsaddr)                                                                   \
  do {                                                                                             \
    if ((ACX_X6_ABL & 1) && g.ksplit != 12345) break;                                              \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                  \
                 : : "v"(gptr), "s"(ldsaddr) : "memory");                                          \
  } while (0)
  // instruction i (0..3) of this wave's share of a unit: plane `pl` of A at K-step S (row offsets RA, tap masks VM) -> slot
#define X6_DMA_A_K(S, KK, RA, VM, pl, slot, i)                                                           \
  do {                                                                                             \
    const unsigned l_ = lds0 + (slot) * X6_UNIT_B + (4 * wave + (i)) * 1024;                       \
    const char* b_ = (const char*)d.A + (size_t)(pl) * (size_t)d.a_plane_stride;                   \
    if constexpr (TN != 0) {                                                                       \
      const int m_ = (KK) * 32 + 8 * wave + 2 * (i) + (lane >> 5);                                 \
      const char* p_ = b_ + ((size_t)(unsigned)m_ * (size_t)(d.lda * 2) + (size_t)(S.m0 * 2 + X6_TCH(i))); \
      X6_GLDS_V((m_ < d.K && (TH == 256 || X6_TCH(i) < TH * 2)) ? p_ : zsrc, l_);                  \
    } else if constexpr (CONV != 0) {                                                              \
      const int tap_ = (KK) / steps_per_tap, kc_ = (KK) - tap_ * steps_per_tap;                    \
      const int t3_ = tap_ / 3;                                                                    \
      const int dn_ = t3_ - 1, dl_ = tap_ - 3 * t3_ - 1;                                           \
      const char* bt_ = b_ + ((ptrdiff_t)(dn_ * d.gl + dl_) * d.lda * 2 + kc_ * 64);   /* uniform: tap shift + channel block */ \
      const bool ok_ = (VM[i] >> tap_) & 1u;                                                       \
      X6_GLDS_V(ok_ ? bt_ + RA[i] : zsrc, l_);                                                     \
    } else {                                                                                       \
      X6_GLDS_S(b_ + (size_t)(KK) * a_kstride, RA[i], l_);                                         \
    }                                                                                              \
  } while (0)
#define X6_DMA_W_K(S, KK, RW, pl, slot, i)                                                               \
  do {                                                                                             \
    if (TN == 0 && (i) >= NI) break;               /* (constant after unrolling) */                 \
    const unsigned l_ = lds0 + (slot) * X6_UNIT_B + ((TN ? 4 : NI) * wave + (i)) * 1024;           \
    if constexpr (TN != 0) {                                                                       \
      const char* b_ = (const char*)d.W + (size_t)(pl) * (size_t)d.w_plane_stride;                 \
      const int m_ = (KK) * 32 + 8 * wave + 2 * (i) + (lane >> 5);                                 \
      bool ok_ = m_ < d.K && (TW >= 256 || X6_TCH(i) < TW * 2);   /* channels behind a narrow tile: zero page */ \
      int row_ = m_;                                                                               \
      if constexpr (CONV != 0) {                                                                   \
        const int n_ = (m_ >> sh_gl) & (d.gn - 1), l_c = m_ & (d.gl - 1);                          \
        ok_ = ok_ && (unsigned)(n_ + S.dn) < (unsigned)d.gn && (unsigned)(l_c + S.dl) < (unsigned)d.gl; \
        row_ = m_ + S.dn * d.gl + S.dl;                                                            \
      }                                                                                            \
      const char* p_ = b_ + ((size_t)(unsigned)(ok_ ? row_ : 0) * (size_t)(d.ldw * 2) + (size_t)(S.c0 * 2 + X6_TCH(i))); \
      X6_GLDS_V(ok_ ? p_ : zsrc, l_);                                                              \
    } else {                                                                                       \
      const char* b_ = (const char*)d.W + (size_t)(pl) * (size_t)d.w_plane_stride + (size_t)(KK) * w_kstride; \
      X6_GLDS_S(b_, RW[i], l_);                                                                    \
    }                                                                                              \
  } while (0)

#define X6_DMA_A(S, RA, VM, pl, slot, i) X6_DMA_A_K(S, S.kk, RA, VM, pl, slot, i)
#define X6_DMA_W(S, RW, pl, slot, i) X6_DMA_W_K(S, S.kk, RW, pl, slot, i)

  X6Src c0, c1;                                  // K-steps gs and gs + 1 of the stream
  unsigned ra0[4], rw0[4], vm0[4] = {0u, 0u, 0u, 0u}, ra1[4], rw1[4], vm1[4] = {0u, 0u, 0u, 0u};
  c0.tk = 0; c0.j = 0; X6_SET_ITEM(c0, 0); X6_ROWS(c0, ra0, rw0, vm0);
  if constexpr (PLAIN != 0) {
    // ---- prologue (PLAIN): the four units of super-step 0 -> slots 0..3 (A of k-step 0, A of k-step 1, W, W)
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_A_K(c0, 2 * c0.kk, ra0, vm0, 0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_A_K(c0, 2 * c0.kk + 1, ra0, vm0, 0, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_W_K(c0, 2 * c0.kk, rw0, 0, 2, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_W_K(c0, 2 * c0.kk + 1, rw0, 0, 3, i);
  } else if constexpr (X3 != 0) {
    // ---- prologue (X3): W.hi, W.mid, A.hi, A.mid of K-step 0 -> the parity-0 slots WH0, WM0, AH, WL
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_W(c0, rw0, 0, X6_WH0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_W(c0, rw0, 1, X6_WM0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_A(c0, ra0, vm0, 0, X6_AH, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) X6_DMA_A(c0, ra0, vm0, 1, X6_WL, i);
  } else {
  // ---- prologue: W.hi, W.mid, A.hi, W.lo of K-step 0 (parity 0 slots)
#pragma unroll
  for (int i = 0; i < 4; ++i) X6_DMA_W(c0, rw0, 0, X6_WH0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) X6_DMA_W(c0, rw0, 1, X6_WM0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) X6_DMA_A(c0, ra0, vm0, 0, X6_AH, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) X6_DMA_W(c0, rw0, 2, X6_WL, i);
  }
  c1 = c0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ra1[i] = ra0[i]; rw1[i] = rw0[i]; vm1[i] = vm0[i]; }
  X6_ADVANCE(c1, ra1, rw1, vm1);

  // ---- fragment addresses inside a unit: row, chunk 2 s + hh at position chunk ^ ((li >> 2) & 3); + 2048 per 32-row block
  const int sw = (li >> 2) & 3;
  const int fa0 = (wm * 128 + li) * 64 + ((0 + hh) ^ sw) * 16, fa1 = (wm * 128 + li) * 64 + ((2 + hh) ^ sw) * 16;
  const int fw0 = (wn * 32 * NI + li) * 64 + ((0 + hh) ^ sw) * 16, fw1 = (wn * 32 * NI + li) * 64 + ((2 + hh) ^ sw) * 16;

  // TN: transpose-read addresses of row block / column block blk, k half r (see x6_tr_frag); + 8192 for substep 1
  int trA[4][2], trW[4][2];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i16 = lane & 15, x_ = 4 * r + (i16 >> 2);
      const int base_ = (8 * hh + x_) * 512 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;
      trA[blk][r] = TN ? base_ + (((wm * 4 + blk) ^ x_) * 64) : 0;
      trW[blk][r] = TN ? base_ + (((wn * NI + (blk < NI ? blk : 0)) ^ x_) * 64) : 0;
    }

  f32x16 acc[4][NI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  // fragment sets: [0..3] A.hi (X) / A.mid (Y) row blocks, [4..7] W.hi, [8..11] W.mid column blocks, [12..15] W.lo column
  // blocks (X) / A.lo row blocks (Y)
  bf16x8 F0[16], F1[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) F1[q][e] = (__bf16)0.f;

#define X6_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (((ACX_X6_ABL & 8) && g.ksplit != 12345) ? 0 : (off))))
  // fragment q (0..15) of substep s of an X / Y half-step; wpar = byte offset of this K-step's W.hi / W.mid slots
#define X6_LOADF(dst, slotoff, TRX, fx0, fx1, blk, s)                                              \
  do {                                                                                             \
    if constexpr (TN != 0) dst = x6_tr_frag(smem, (slotoff) + (s) * 8192 + TRX[blk][0], (slotoff) + (s) * 8192 + TRX[blk][1]); \
    else dst = X6_FRAG((slotoff) + ((s) ? fx1 : fx0) + (blk) * 2048);                              \
  } while (0)
#define X6_RD_X(F, s, q)                                                                           \
  do {                                                                                             \
    if ((q) >= 4 && ((q) & 3) >= NI) break;        /* W column blocks the strip does not have */    \
    if ((q) < 4) X6_LOADF(F[q], X6_AH * X6_UNIT_B, trA, fa0, fa1, (q) & 3, s);                     \
    else if ((q) < 8) X6_LOADF(F[q], wpar + X6_WH0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);        \
    else if ((q) < 12) X6_LOADF(F[q], wpar + X6_WM0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);       \
    else X6_LOADF(F[q], X6_WL * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);                             \
  } while (0)
#define X6_RD_Y(F, s, q)                                                                           \
  do {                                                                                             \
    if ((q) >= 4 && (q) < 12 && ((q) & 3) >= NI) break;                                            \
    if ((q) < 4) X6_LOADF(F[q], X6_AM * X6_UNIT_B, trA, fa0, fa1, (q) & 3, s);                     \
    else if ((q) < 8) X6_LOADF(F[q], wpar + X6_WH0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);        \
    else if ((q) < 12) X6_LOADF(F[q], wpar + X6_WM0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);       \
    else X6_LOADF(F[q], X6_AL * X6_UNIT_B, trA, fa0, fa1, (q) & 3, s);                             \
  } while (0)
  // MFMA q (0..47) of a fragment set: product q / 16, row block (q % 16) / 4, column block q % 4.  Operands swapped: the
  // accumulator holds C^T (lane = output row, registers = 4-column groups), as in the p8 kernel.
  //   X: (A.hi, W.lo) (A.hi, W.mid) (A.hi, W.hi)        Y: (A.lo, W.hi) (A.mid, W.mid) (A.mid, W.hi)
#define X6_MM_X(F, q)                                                                              \
  do {                                                                                             \
    constexpr int p_ = (q) / 16, mi_ = ((q) % 16) / 4, ni_ = (q) % 4;                              \
    if constexpr (ni_ < NI)                                                                        \
    acc[mi_][ni_] = x6_mfma<0>(F[(p_ == 0 ? 12 : p_ == 1 ? 8 : 4) + ni_], F[mi_], acc[mi_][ni_]); \
  } while (0)
#define X6_MM_Y(F, q)                                                                              \
  do {                                                                                             \
    constexpr int p_ = (q) / 16, mi_ = ((q) % 16) / 4, ni_ = (q) % 4;                              \
    if constexpr (ni_ < NI)                                                                        \
    acc[mi_][ni_] = x6_mfma<(X3 == 2)>(F[(p_ == 1 ? 8 : 4) + ni_], F[(p_ == 0 ? 12 : 0) + mi_], acc[mi_][ni_]); \
  } while (0)
#define X6_FENCE() __builtin_amdgcn_sched_barrier(0)
#define X6_REP48(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) M(32) M(33) M(34) M(35) M(36) M(37) M(38) M(39) M(40) M(41) M(42) M(43) M(44) M(45) M(46) M(47)

  int wpar = 0;                                  // 0 / 6 units: the W.hi / W.mid slots of the current K-step
  for (int j = 0; j < my_items; ++j) {
    const int steps_j = c0.steps;                // (c0 is at the item's first K-step here)
    for (int tk = 0; tk < steps_j; ++tk) {
      if constexpr (PLAIN != 0) {
        // ======================================================================= one super-step (64 k) of a plain product
        const int par = wpar ? 4 * X6_UNIT_B : 0, nxt = wpar ? 0 : 4;       // this super-step's slots (byte offset), the next one's (slot)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        X6_FENCE();
        __builtin_amdgcn_s_barrier();
        X6_FENCE();
        // fragment q (0..3: row blocks of A, 4..7: column blocks of W) of substep s (0..3: 16 k each; 0, 1 in the first K-step's units)
#define X6_RD_P(F, s, q)                                                                           \
  do {                                                                                             \
    if ((q) < 4) F[q] = X6_FRAG(par + ((s) >> 1) * X6_UNIT_B + (((s) & 1) ? fa1 : fa0) + (q) * 2048); \
    else F[q] = X6_FRAG(par + (2 + ((s) >> 1)) * X6_UNIT_B + (((s) & 1) ? fw1 : fw0) + ((q) - 4) * 2048); \
  } while (0)
#define X6_MM_P(F, q) acc[(q) >> 2][(q) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[4 + ((q) & 3)], F[(q) >> 2], acc[(q) >> 2][(q) & 3], 0, 0, 0)
        // block A: the previous super-step's last substep (F1; zero at an item's first super-step) + reads of substep 0 -> F0 + the
        // 16 DMA instructions of the NEXT super-step's units (their slots were read last before the barrier above)
#define X6_BA(q)                                                                                   \
  do {                                                                                             \
    if constexpr ((q) < 8) X6_RD_P(F0, 0, (q) & 7);                                                \
    { constexpr int u_ = ((q) >> 2) & 3, i_ = (q) & 3;                                             \
      if (u_ == 0) X6_DMA_A_K(c1, 2 * c1.kk, ra1, vm1, 0, nxt + 0, i_);                            \
      else if (u_ == 1) X6_DMA_A_K(c1, 2 * c1.kk + 1, ra1, vm1, 0, nxt + 1, i_);                   \
      else if (u_ == 2) X6_DMA_W_K(c1, 2 * c1.kk, rw1, 0, nxt + 2, i_);                            \
      else X6_DMA_W_K(c1, 2 * c1.kk + 1, rw1, 0, nxt + 3, i_); }                                   \
    X6_MM_P(F1, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
#define X6_BX(q, FC, FN, sn)                                                                       \
  do {                                                                                             \
    if constexpr ((q) < 8) X6_RD_P(FN, sn, (q) & 7);                                               \
    X6_MM_P(FC, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
#define X6_REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
        X6_REP16(X6_BA)
#define X6_BB(q) X6_BX(q, F0, F1, 1)
#define X6_BC(q) X6_BX(q, F1, F0, 2)
#define X6_BD(q) X6_BX(q, F0, F1, 3)
        X6_REP16(X6_BB)
        X6_REP16(X6_BC)
        X6_REP16(X6_BD)
#undef X6_BD
#undef X6_BC
#undef X6_BB
#undef X6_BX
#undef X6_BA
        wpar = wpar ? 0 : 1;
        c0 = c1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra0[i] = ra1[i]; rw0[i] = rw1[i]; vm0[i] = vm1[i]; }
        X6_ADVANCE(c1, ra1, rw1, vm1);
        continue;
      }
      const int wnext = wpar ? 0 : 6 * X6_UNIT_B;
      const int slot_wh_next = wpar ? X6_WH0 : X6_WH1, slot_wm_next = wpar ? X6_WM0 : X6_WM1;
      if constexpr (X3 != 0) {
        // ======================================================================= one K-step of the three-product mode (half-step Z)
        const int apar = wpar ? 2 * X6_UNIT_B : 0;                            // A.hi: AH / AM, A.mid: WL / AL by K-step parity
        const int sa_next = wpar ? X6_AH : X6_AM, sm_next = wpar ? X6_WL : X6_AL;
        if (ACX_X6_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        X6_FENCE();
        __builtin_amdgcn_s_barrier();
        X6_FENCE();
#define X6_RD_Z(F, s, q)                                                                           \
  do {                                                                                             \
    if ((q) >= 4 && (q) < 12 && ((q) & 3) >= NI) break;                                            \
    if ((q) < 4) X6_LOADF(F[q], apar + X6_AH * X6_UNIT_B, trA, fa0, fa1, (q) & 3, s);              \
    else if ((q) < 8) X6_LOADF(F[q], wpar + X6_WH0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);        \
    else if ((q) < 12) X6_LOADF(F[q], wpar + X6_WM0 * X6_UNIT_B, trW, fw0, fw1, (q) & 3, s);       \
    else X6_LOADF(F[q], apar + X6_WL * X6_UNIT_B, trA, fa0, fa1, (q) & 3, s);                      \
  } while (0)
        // block 1: the previous K-step's second substep (F1; zero at an item's first K-step) with this K-step's first-substep reads
        // (-> F0) and the 16 DMA instructions of the NEXT K-step's four units (other parity) between the MFMAs
#define X6_B1_Z(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_Z(F0, 0, (q) & 15);                                              \
    if constexpr ((q) >= 16 && (q) < 48 && (((q) - 16) % 2) == 0) {                                \
      constexpr int u_ = (((q) - 16) / 8) & 3, i_ = (((q) - 16) / 2) & 3;                          \
      if (u_ == 0) X6_DMA_W(c1, rw1, 0, slot_wh_next, i_);                                         \
      else if (u_ == 1) X6_DMA_W(c1, rw1, 1, slot_wm_next, i_);                                    \
      else if (u_ == 2) X6_DMA_A(c1, ra1, vm1, 0, sa_next, i_);                                    \
      else X6_DMA_A(c1, ra1, vm1, 1, sm_next, i_);                                                 \
    }                                                                                              \
    X6_MM_Y(F1, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
        X6_REP48(X6_B1_Z)
#undef X6_B1_Z
#define X6_B2_Z(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_Z(F1, 1, (q) & 15);                                              \
    X6_MM_Y(F0, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
        X6_REP48(X6_B2_Z)
#undef X6_B2_Z
#undef X6_RD_Z
        wpar = wnext;
        c0 = c1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra0[i] = ra1[i]; rw0[i] = rw1[i]; vm0[i] = vm1[i]; }
        X6_ADVANCE(c1, ra1, rw1, vm1);
        continue;
      }
      // =========================================================================== half-step X
      if (ACX_X6_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      X6_FENCE();
      __builtin_amdgcn_s_barrier();
      X6_FENCE();
      // block 1: the previous half-step's second substep (set F1, Y products; F1 is ZERO at an item's first K-step: no branch
      // around the accumulators, whose register assignment hipcc only keeps in place in straight-line code) with this
      // half-step's first-substep reads (-> F0) and its 12 DMA instructions between the MFMAs
#define X6_B1_X(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_X(F0, 0, (q) & 15);                                                             \
    if constexpr ((q) >= 16 && (q) < 40 && (((q) - 16) % 2) == 0) {                                          \
      constexpr int u_ = (((q) - 16) / 8) & 3, i_ = (((q) - 16) / 2) & 3;                                \
      if (u_ == 0) X6_DMA_A(c0, ra0, vm0, 1, X6_AM, i_);                                                     \
      else if (u_ == 1) X6_DMA_A(c0, ra0, vm0, 2, X6_AL, i_);                                                \
      else X6_DMA_W(c1, rw1, 0, slot_wh_next, i_);                                                      \
    }                                                                                              \
    X6_MM_Y(F1, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
      X6_REP48(X6_B1_X)
#undef X6_B1_X
      // block 2: this half-step's first substep (F0, X products) with its second-substep reads (-> F1)
#define X6_B2_X(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_X(F1, 1, (q) & 15);                                                             \
    X6_MM_X(F0, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
      X6_REP48(X6_B2_X)
#undef X6_B2_X
      // =========================================================================== half-step Y
      if (ACX_X6_ABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(TN ? 4 : NI) : "memory");     // A.mid, A.lo landed (W.hi of the next K-step -- NI instructions -- may be in flight)
      X6_FENCE();
      __builtin_amdgcn_s_barrier();
      X6_FENCE();
#define X6_B1_Y(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_Y(F0, 0, (q) & 15);                                                             \
    if constexpr ((q) >= 16 && (q) < 40 && (((q) - 16) % 2) == 0) {                                          \
      constexpr int u_ = (((q) - 16) / 8) & 3, i_ = (((q) - 16) / 2) & 3;                                \
      if (u_ == 0) X6_DMA_W(c1, rw1, 1, slot_wm_next, i_);                                              \
      else if (u_ == 1) X6_DMA_A(c1, ra1, vm1, 0, X6_AH, i_);                                                \
      else X6_DMA_W(c1, rw1, 2, X6_WL, i_);                                                             \
    }                                                                                              \
    X6_MM_X(F1, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
      X6_REP48(X6_B1_Y)
#undef X6_B1_Y
#define X6_B2_Y(q)                                                                                 \
  do {                                                                                             \
    if constexpr ((q) < 16) X6_RD_Y(F1, 1, (q) & 15);                                                             \
    X6_MM_Y(F0, (q));                                                                              \
    X6_FENCE();                                                                                    \
  } while (0);
      X6_REP48(X6_B2_Y)
#undef X6_B2_Y
      wpar = wnext;
      c0 = c1;
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra0[i] = ra1[i]; rw0[i] = rw1[i]; vm0[i] = vm1[i]; }
      X6_ADVANCE(c1, ra1, rw1, vm1);
    }
    // ---- item end: the last substep (F1, Y products), then the epilogue.  The DMA queue is drained first: the epilogue's
    // stores must not sit in front of a counted wait (one in-order vmcnt for loads and stores)
    if constexpr (PLAIN != 0) {
#define X6_DRAINP(q) X6_MM_P(F1, (q));
      X6_REP16(X6_DRAINP)
#undef X6_DRAINP
    } else {
#define X6_DRAIN(q) X6_MM_Y(F1, (q));
    X6_REP48(X6_DRAIN)
#undef X6_DRAIN
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      const int L = b0 + j * G;
      const int ti = L / ksplit, ks = L - ti * ksplit;
      const int tile = g.tile0 + ti;
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const int m0 = tm * TH, n0 = tn * TW;
      const bool raw = ksplit > 1;               // split items: raw f32 partial tile [ks][M][N], epilogue in the reduce launch
      float* Cf = raw ? g.partial + (size_t)ks * d.M * d.N : (float*)d.C;
      const int ldc = raw ? d.N : d.ldc;
      // each 32x32 accumulator tile goes through this wave's private 4 KB of LDS (XOR-swizzled 128-B rows) and comes back
      // row-major: lane l owns 4 consecutive columns (l & 7) of row (l >> 3) + 8 pass: 4 stores of 8 rows x 128 B per tile.
      char* scr = smem + 8 * X6_UNIT_B + wave * 4096;
      // fp16 planes (X3 = 2): the operands' power-of-two scales leave here; the other instantiations multiply by the CONSTANT 1 (no code)
      const float osc_ = (X3 == 2) ? (d.out_scale != 0.f ? d.out_scale : 1.f) : 1.f;
      const int rl = lane >> 3, cj = lane & 7;
      const int colw = n0 + wn * 32 * NI + 4 * cj;   // + 32 ni
      const int roww = m0 + wm * 128 + rl;       // + 32 mi + 8 ps
      // bias of the four column blocks: consumed HERE, outside every store -- a load still pending when stores are issued
      // makes hipcc's counted waits cover the stores as well (one in-order vmcnt)
      // (plane outputs, C_MODE 2: lane l owns EIGHT consecutive columns (l & 3) of row (l >> 2) + 16 pass -- 16-byte bf16 stores,
      // 6 instead of 12 store instructions per tile: the store path takes ~64 cycles per wave-instruction whatever its width)
      const int colp = n0 + wn * 32 * NI + 8 * (lane & 3);   // + 32 ni
      const int rowp = m0 + wm * 128 + (lane >> 2);      // + 32 mi + 16 pass
      float4 bia[NI], bib[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bia[ni] = bib[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias && !raw) {
          if constexpr (C_MODE == 2) {
            bia[ni] = *reinterpret_cast<const float4*>(d.bias + min(colp + 32 * ni, d.N - 8));
            bib[ni] = *reinterpret_cast<const float4*>(d.bias + min(colp + 32 * ni, d.N - 8) + 4);
          } else {
            bia[ni] = *reinterpret_cast<const float4*>(d.bias + min(colw + 32 * ni, d.N - 4));
          }
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        asm volatile("" : "+v"(bia[ni].x), "+v"(bia[ni].y), "+v"(bia[ni].z), "+v"(bia[ni].w));
        if constexpr (C_MODE == 2) asm volatile("" : "+v"(bib[ni].x), "+v"(bib[ni].y), "+v"(bib[ni].z), "+v"(bib[ni].w));
      }
      // residual of accumulator tile bi = 4 ni + mi into R (rows / columns clamped: an edge tile masks at the store)
#define X6_RES_LOAD(R, bi)                                                                         \
  do {                                                                                             \
    if constexpr (RES != 0) {                                                                      \
      _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                             \
        R[ps] = *reinterpret_cast<const float4*>(                                                  \
            d.residual + (size_t)min(roww + ((bi) & 3) * 32 + 8 * ps, d.M - 1) * d.ldr + min(colw + 32 * ((bi) >> 2), d.N - 4)); \
    }                                                                                              \
  } while (0)
      // transposer + epilogue arithmetic + stores of accumulator tile bi; PRED: mask rows / columns outside the problem
#define X6_EPI_BLOCK(R, bi, PRED)                                                                  \
  do {                                                                                             \
    constexpr int ni = (bi) >> 2, mi = (bi) & 3;                                                   \
    const int col = colw + 32 * ni;                                                                \
    const float4 b4 = bia[ni];                                                                     \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)   /* accumulator (C^T layout): row li, columns 8k + 4hh .. +3 = chunk 2k + hh */ \
      *reinterpret_cast<float4*>(scr + li * 128 + (((2 * k + hh) ^ (li & 7)) * 16)) =             \
          make_float4(acc[mi][ni][4 * k], acc[mi][ni][4 * k + 1], acc[mi][ni][4 * k + 2], acc[mi][ni][4 * k + 3]); \
    if constexpr (C_MODE == 2) {                                                                   \
      _Pragma("unroll") for (int ps = 0; ps < 2; ++ps) {                                           \
        const int rr_ = (lane >> 2) + 16 * ps;                                                     \
        const float4 va_ = *reinterpret_cast<const float4*>(scr + rr_ * 128 + (((2 * (lane & 3)) ^ (rr_ & 7)) * 16));     \
        const float4 vb_ = *reinterpret_cast<const float4*>(scr + rr_ * 128 + (((2 * (lane & 3) + 1) ^ (rr_ & 7)) * 16)); \
        float ov[8] = {va_.x * osc_ + bia[ni].x, va_.y * osc_ + bia[ni].y, va_.z * osc_ + bia[ni].z, va_.w * osc_ + bia[ni].w,  \
                       vb_.x * osc_ + bib[ni].x, vb_.y * osc_ + bib[ni].y, vb_.z * osc_ + bib[ni].z, vb_.w * osc_ + bib[ni].w};  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
          if constexpr (ACT == ACX_ACT_QUICKGELU) ov[e] = acx_quickgelu(ov[e]);                    \
          else if constexpr (ACT == ACX_ACT_LEAKYRELU) ov[e] = ov[e] > 0.f ? ov[e] : 0.01f * ov[e]; \
        }                                                                                          \
        /* three bf16 planes hi | mid | lo of the f32 value (ACX_BF16X3): the next pairs = 6 product's A operand */ \
        uint4 ph, pm, pl;                                                                          \
        float r1[8], r2[8];                                                                        \
        constexpr int PF_ = (X3 == 2);             /* plane format: bf16 triples, or fp16 pairs (X3 = 2) */ \
        ph.x = acx_pk2<PF_>(ov[0], ov[1]); ph.y = acx_pk2<PF_>(ov[2], ov[3]); ph.z = acx_pk2<PF_>(ov[4], ov[5]); ph.w = acx_pk2<PF_>(ov[6], ov[7]); \
        const unsigned hw_[4] = {ph.x, ph.y, ph.z, ph.w};                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
          r1[2 * e] = ov[2 * e] - acx_unpk_lo<PF_>(hw_[e]); r1[2 * e + 1] = ov[2 * e + 1] - acx_unpk_hi<PF_>(hw_[e]); \
        }                                                                                          \
        pm.x = acx_pk2<PF_>(r1[0], r1[1]); pm.y = acx_pk2<PF_>(r1[2], r1[3]); pm.z = acx_pk2<PF_>(r1[4], r1[5]); pm.w = acx_pk2<PF_>(r1[6], r1[7]); \
        const unsigned mw_[4] = {pm.x, pm.y, pm.z, pm.w};                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                            \
          r2[2 * e] = r1[2 * e] - acx_unpk_lo<PF_>(mw_[e]); r2[2 * e + 1] = r1[2 * e + 1] - acx_unpk_hi<PF_>(mw_[e]); \
        }                                                                                          \
        pl.x = f2bf2(r2[0], r2[1]); pl.y = f2bf2(r2[2], r2[3]); pl.z = f2bf2(r2[4], r2[5]); pl.w = f2bf2(r2[6], r2[7]); \
        const int row = rowp + mi * 32 + 16 * ps, colq = colp + 32 * ni;                           \
        if ((!(PRED) || (colq < d.N && row < d.M)) && (!(ACX_X6_ABL & 4) || g.ksplit == 12345)) {  \
          typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));                         \
          const size_t crows_ = d.c_plane_rows ? (size_t)d.c_plane_rows : (size_t)d.M;             \
          const size_t pe = crows_ * d.ldc;                                                        \
          u16* dst = (u16*)d.C + (cpanel ? ((size_t)(colq >> 5) * crows_ + row) * 32 + (colq & 31) : (size_t)row * d.ldc + colq); \
          __builtin_nontemporal_store(*reinterpret_cast<const u32x4_*>(&ph), reinterpret_cast<u32x4_*>(dst)); \
          __builtin_nontemporal_store(*reinterpret_cast<const u32x4_*>(&pm), reinterpret_cast<u32x4_*>(dst + pe)); \
          if constexpr (X3 == 0)   /* (three-product mode: the consumer never reads the lo plane) */   \
          __builtin_nontemporal_store(*reinterpret_cast<const u32x4_*>(&pl), reinterpret_cast<u32x4_*>(dst + 2 * pe)); \
        }                                                                                          \
      }                                                                                            \
    } else                                                                                         \
    _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                             \
      const int rr_ = rl + 8 * ps;                                                                 \
      float4 v = *reinterpret_cast<const float4*>(scr + rr_ * 128 + ((cj ^ (rr_ & 7)) * 16));     \
      if constexpr (X3 == 2) { v.x *= osc_; v.y *= osc_; v.z *= osc_; v.w *= osc_; }               \
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;                                          \
      if constexpr (ACT == ACX_ACT_QUICKGELU) {                                                    \
        v.x = acx_quickgelu(v.x); v.y = acx_quickgelu(v.y);                                        \
        v.z = acx_quickgelu(v.z); v.w = acx_quickgelu(v.w);                                        \
      } else if constexpr (ACT == ACX_ACT_LEAKYRELU) {                                             \
        v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;                  \
        v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;                  \
      }                                                                                            \
      if constexpr (RES != 0) { v.x += R[ps].x; v.y += R[ps].y; v.z += R[ps].z; v.w += R[ps].w; }  \
      const int row = roww + mi * 32 + 8 * ps;                                                     \
      if ((!(PRED) || (col < d.N && row < d.M)) && (!(ACX_X6_ABL & 4) || g.ksplit == 12345)) {     \
        if constexpr (C_MODE == 1) {                                                        \
          uint2 pk;                                                                                \
          pk.x = f2bf2(v.x, v.y);                                                                  \
          pk.y = f2bf2(v.z, v.w);                                                                  \
          __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&pk),                        \
                                      reinterpret_cast<u32x2*>((u16*)d.C + (size_t)row * d.ldc + col)); \
        } else {                                                                                   \
          f32x4* dst = reinterpret_cast<f32x4*>(Cf + (size_t)row * ldc + col);                     \
          if constexpr (RES != 0) *dst = *reinterpret_cast<const f32x4*>(&v);                      \
          else __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&v), dst);              \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;                          \
  } while (0)
      float4 R0[4], R1[4];
      if (m0 + TH <= d.M && n0 + TW <= d.N) {
        // full tile: unpredicated stores (straight-line code: hipcc's counted waits then leave the stores in flight), the
        // residual loaded TWO accumulator tiles ahead of its use -- a wait for it never covers the stores of the last two
#define X6_FULL2(bi) X6_EPI_BLOCK(R0, bi, false); X6_RES_LOAD(R0, (bi) + 2); X6_EPI_BLOCK(R1, (bi) + 1, false); X6_RES_LOAD(R1, (bi) + 3);
        X6_RES_LOAD(R0, 0); X6_RES_LOAD(R1, 1);
        if constexpr (NI == 4) {
          X6_FULL2(0) X6_FULL2(2) X6_FULL2(4) X6_FULL2(6) X6_FULL2(8) X6_FULL2(10) X6_FULL2(12)
          X6_EPI_BLOCK(R0, 14, false); X6_EPI_BLOCK(R1, 15, false);
        } else if constexpr (NI == 2) {
          X6_FULL2(0) X6_FULL2(2) X6_FULL2(4)
          X6_EPI_BLOCK(R0, 6, false); X6_EPI_BLOCK(R1, 7, false);
        } else {
          X6_FULL2(0)
          X6_EPI_BLOCK(R0, 2, false); X6_EPI_BLOCK(R1, 3, false);
        }
#undef X6_FULL2
      } else {
        // edge tile: rows / columns masked at the store (a pending residual load then makes every store block wait: one
        // accumulator tile at a time, edge tiles only)
#define X6_EDGE(bi)                                                                                \
  do {                                                                                             \
    X6_RES_LOAD(R0, bi);                                                                           \
    if constexpr (RES != 0) {                                                                      \
      _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                             \
        asm volatile("" : "+v"(R0[ps].x), "+v"(R0[ps].y), "+v"(R0[ps].z), "+v"(R0[ps].w));         \
    }                                                                                              \
    X6_EPI_BLOCK(R0, bi, true);                                                                    \
  } while (0);
        X6_EDGE(0) X6_EDGE(1) X6_EDGE(2) X6_EDGE(3)
        if constexpr (NI >= 2) { X6_EDGE(4) X6_EDGE(5) X6_EDGE(6) X6_EDGE(7) }
        if constexpr (NI == 4) { X6_EDGE(8) X6_EDGE(9) X6_EDGE(10) X6_EDGE(11) X6_EDGE(12) X6_EDGE(13) X6_EDGE(14) X6_EDGE(15) }
#undef X6_EDGE
      }
#undef X6_EPI_BLOCK
#undef X6_RES_LOAD
    }
    // the next item's first block 1 multiplies F1: zero (after the epilogue: the registers are free for the residual until here)
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) F1[q][e] = (__bf16)0.f;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // no DMA may still be writing this workgroup's LDS at exit
#undef X6_REP48
#undef X6_REP16
#undef X6_MM_P
#undef X6_RD_P
#undef X6_DMA_W_K
#undef X6_DMA_A_K
#undef X6_FENCE
#undef X6_MM_Y
#undef X6_MM_X
#undef X6_RD_Y
#undef X6_RD_X
#undef X6_LOADF
#undef X6_TCH
#undef X6_FRAG
#undef X6_DMA_W
#undef X6_DMA_A
#undef X6_GLDS_V
#undef X6_GLDS_S
#undef X6_ROWS
#undef X6_ADVANCE
#undef X6_SET_ITEM
}
