// Metrics epilogue (reference anomaly_clip_module.py:501-626 + torchmetrics 0.11.0
// curve functions).  HBM-bound integer work: a stable LSD radix sort of (score, label) pairs, one scan pass that
// turns the sorted pairs into the exact AUROC / AP / Youden-optimal threshold, and a counting kernel for
// y_pred / top-1 / top-5 / confusion / F1.  No floating-point atomics anywhere: AUROC and the threshold argmax
// are exact int64 arithmetic, AP is a fixed-order f64 two-stage reduction, the counters are integer atomics.
#include "acx_internal.h"

namespace {

constexpr int MT_THREADS = 256;
constexpr int MT_ITEMS = 8;
constexpr int MT_TILE = MT_THREADS * MT_ITEMS;   // 2048 elements per block

// ------------------------------------------------------------------------------------------------ radix sort
__device__ __forceinline__ uint32_t key_encode(uint32_t u, int desc) {
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;     // IEEE-754 total order as unsigned ascending
  return desc ? ~u : u;
}
__device__ __forceinline__ uint32_t key_decode(uint32_t u, int desc) {
  if (desc) u = ~u;
  return u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
}

// per-block digit histogram, written digit-major ([256][nblocks]) so one linear scan yields the scatter bases
// (blockIdx.y = problem of a batched sort: keys kstride apart, histograms 256 * nblocks apart)
__global__ __launch_bounds__(MT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ kin,
                                                                 uint32_t* __restrict__ hist, int n, int nblocks,
                                                                 int shift, int first, int desc, size_t kstride) {
  __shared__ uint32_t h[256];
  const int tid = threadIdx.x;
  kin += blockIdx.y * kstride;
  hist += (size_t)blockIdx.y * 256 * nblocks;
  h[tid] = 0;
  __syncthreads();
  const int base = blockIdx.x * MT_TILE;
#pragma unroll
  for (int it = 0; it < MT_ITEMS; ++it) {
    const int idx = base + it * MT_THREADS + tid;
    if (idx < n) {
      uint32_t k = kin[idx];
      if (first) k = key_encode(k, desc);
      atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  hist[(size_t)tid * nblocks + blockIdx.x] = h[tid];
}

// Scatter bases, stage 1: block d turns row d of the digit-major histogram ([256][nblocks]) into its exclusive prefix
// over the tiles (in place) and writes the digit's total to tot[d].  256 independent row scans instead of ONE block
// walking all 256 * nblocks entries (that single-block scan was 82 % of the epilogue's kernel time); stage 2 -- the
// exclusive scan of the 256 digit totals -- is done by every scatter block in LDS.
__global__ __launch_bounds__(256) void scan_rows_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ tot, int nblocks) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  hist += (size_t)blockIdx.y * 256 * nblocks;
  tot += blockIdx.y * 256;
  uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 256) {
    const int i = base + tid;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int j = 0; j < w; ++j) woff += wsum[j];
    const uint32_t carry = carry_s;
    if (i < nblocks) row[i] = carry + woff + inc - v;
    __syncthreads();
    if (tid == 255) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) tot[blockIdx.x] = carry_s;
}

// stable scatter: wave w owns elements [w*512, (w+1)*512) of the tile and walks them in array order, ranking
// equal digits with 8 ballots per item; per-wave running counters live in LDS (wave-private, no barrier needed).
__global__ __launch_bounds__(MT_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ kin,
                                                                    const uint32_t* __restrict__ vin,
                                                                    uint32_t* __restrict__ kout,
                                                                    uint32_t* __restrict__ vout,
                                                                    const uint32_t* __restrict__ bases,
                                                                    const uint32_t* __restrict__ tot, int n,
                                                                    int nblocks, int shift, int first, int last,
                                                                    int desc, size_t kstride, size_t vstride, size_t ostride) {
  __shared__ uint32_t cnt[4][256];
  __shared__ uint32_t dbase[256];
  __shared__ uint32_t dws[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  kin += blockIdx.y * kstride; vin += blockIdx.y * vstride;
  kout += blockIdx.y * ostride; vout += blockIdx.y * ostride;
  bases += (size_t)blockIdx.y * 256 * nblocks; tot += blockIdx.y * 256;
  {   // exclusive scan of the 256 digit totals (thread = digit)
    const uint32_t v = tot[tid];
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) dws[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int j = 0; j < wave; ++j) woff += dws[j];
    dbase[tid] = woff + inc - v;
  }
  for (int i = tid; i < 1024; i += MT_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * MT_TILE + wave * (MT_TILE / 4);
  uint32_t k[MT_ITEMS], v[MT_ITEMS], r[MT_ITEMS];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int it = 0; it < MT_ITEMS; ++it) {
    const int idx = base + it * 64 + lane;
    const bool valid = idx < n;
    uint32_t kk = valid ? kin[idx] : 0u;
    if (first) kk = key_encode(kk, desc);
    k[it] = kk;
    v[it] = valid ? vin[idx] : 0u;
    const uint32_t d = (kk >> shift) & 255u;
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t rank_in = __popcll(m & lt);
    const uint32_t prefix = valid ? cnt[wave][d] : 0u;
    r[it] = prefix + rank_in;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank_in == 0) cnt[wave][d] = prefix + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    const uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid];
    const uint32_t gb = dbase[tid] + bases[(size_t)tid * nblocks + blockIdx.x];
    cnt[0][tid] = gb;
    cnt[1][tid] = gb + c0;
    cnt[2][tid] = gb + c0 + c1;
    cnt[3][tid] = gb + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MT_ITEMS; ++it) {
    const int idx = base + it * 64 + lane;
    if (idx < n) {
      const uint32_t pos = cnt[wave][(k[it] >> shift) & 255u] + r[it];
      kout[pos] = last ? key_decode(k[it], desc) : k[it];
      vout[pos] = v[it];
    }
  }
}

// ------------------------------------------------------------------------------------------------ block scans
__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
__device__ __forceinline__ long long wave_incl_max(long long v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    long long t = __shfl_up(v, off, 64);
    if (lane >= off) v = max(v, t);
  }
  return v;
}
// exclusive block scans over 256 threads (4 waves); `sh` needs 4 slots; returns the exclusive prefix and the total
__device__ __forceinline__ int block_excl_sum(int v, int* sh, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int inc = wave_incl_sum(v, lane);
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  int off = 0;
  for (int i = 0; i < w; ++i) off += sh[i];
  total = sh[0] + sh[1] + sh[2] + sh[3];
  return off + inc - v;
}
__device__ __forceinline__ long long block_excl_max(long long v, long long* sh, long long& total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long inc = wave_incl_max(v, lane);
  long long prev = __shfl_up(inc, 1, 64);
  if (lane == 0) prev = -1;
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  long long off = -1;
  for (int i = 0; i < w; ++i) off = max(off, sh[i]);
  total = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
  return max(off, prev);
}

// ------------------------------------------------------------------------------------------------ curve
// Per-block workspace records (one per tile of the sorted array).
struct CurveBlock {
  int tsum;            // positives in the tile
  int bcnt;            // distinct-threshold points (run ends) in the tile
  long long last;      // packed (index << 32 | tps) of the tile's last run end, -1 if none   [after scan: tps is global]
  int toff;            // exclusive prefix of tsum
  int coff;            // exclusive prefix of bcnt
  long long prev;      // packed (index << 32 | global tps) of the last run end BEFORE the tile, -1 if none
  long long auc2;      // sum (fps_i - fps_prev) * (tps_i + tps_prev)
  double ap;           // sum (tps_i - tps_prev) * tps_i / (tps_i + fps_i)
  long long jval;      // max tps*N - fps*P over the tile's points
  long long jidx;      // its sorted index (smallest on ties), -1 if none
};

struct TileScan {
  int tp[MT_ITEMS];       // local inclusive positives count at each item
  unsigned ends;          // bit it: item is a run end (distinct threshold point)
  int tsum;               // thread's positives
};

// per-problem parameters of a batched curve launch (blockIdx.y = problem): class id, "target = label != cls" flag
struct CurveBatch {
  int cls[64];
  unsigned long long negate;
  size_t sstride, lstride;      // elements between the problems' score / label arrays
};

__device__ __forceinline__ bool is_target(uint32_t label, int cls, int negate) {
  return negate ? ((int)label != cls) : ((int)label == cls);
}

// thread t handles the 8 consecutive elements [t*8, t*8+8) of the tile
__device__ __forceinline__ void tile_load(const float* __restrict__ s, const uint32_t* __restrict__ lab, int n,
                                          int cls, int negate, TileScan& ts) {
  const int i0 = blockIdx.x * MT_TILE + threadIdx.x * MT_ITEMS;
  float sc[MT_ITEMS + 1];
#pragma unroll
  for (int it = 0; it <= MT_ITEMS; ++it) sc[it] = (i0 + it < n) ? s[i0 + it] : 0.f;
  int run = 0;
  ts.ends = 0;
#pragma unroll
  for (int it = 0; it < MT_ITEMS; ++it) {
    const int i = i0 + it;
    if (i < n) {
      run += is_target(lab[i], cls, negate) ? 1 : 0;
      if (i == n - 1 || sc[it] != sc[it + 1]) ts.ends |= 1u << it;
    }
    ts.tp[it] = run;
  }
  ts.tsum = run;
}

__global__ __launch_bounds__(MT_THREADS) void curve_stats_kernel(const float* __restrict__ s,
                                                                  const uint32_t* __restrict__ lab, int n,
                                                                  const CurveBatch cb, CurveBlock* __restrict__ blk) {
  __shared__ int shi[4];
  __shared__ long long shl[4];
  const int cls = cb.cls[blockIdx.y], negate = (int)((cb.negate >> blockIdx.y) & 1ull);
  s += blockIdx.y * cb.sstride; lab += blockIdx.y * cb.lstride; blk += (size_t)blockIdx.y * gridDim.x;
  TileScan ts;
  tile_load(s, lab, n, cls, negate, ts);
  int total;
  const int excl = block_excl_sum(ts.tsum, shi, total);
  int cnt_total;
  (void)block_excl_sum(__popc(ts.ends), shi, cnt_total);
  long long mine = -1;
  if (ts.ends) {
    const int it = 31 - __clz(ts.ends);
    const long long idx = (long long)blockIdx.x * MT_TILE + threadIdx.x * MT_ITEMS + it;
    mine = (idx << 32) | (unsigned)(excl + ts.tp[it]);       // tps still tile-local here
  }
  long long last;
  (void)block_excl_max(mine, shl, last);
  if (threadIdx.x == 0) {
    blk[blockIdx.x].tsum = total;
    blk[blockIdx.x].bcnt = cnt_total;
    blk[blockIdx.x].last = last;
  }
}

struct CurveResultDev {       // == struct acx_curve_result
  double auroc, ap;
  long long n_pos, n_neg, n_distinct, opt_index;
  float opt_threshold, pad;
};

// one block: prefixes across tiles (tsum -> toff, bcnt -> coff, last -> prev with GLOBAL tps) and the totals
__global__ __launch_bounds__(1024) void curve_prefix_kernel(CurveBlock* __restrict__ blk, int nb, long long n,
                                                             CurveResultDev* __restrict__ res) {
  __shared__ int ps[1024], pc[1024];
  __shared__ long long pm[1024];
  const int tid = threadIdx.x;
  blk += (size_t)blockIdx.x * nb; res += blockIdx.x;          // blockIdx.x = problem
  const int per = (nb + 1023) / 1024;
  const int lo = min(tid * per, nb), hi = min(lo + per, nb);
  int s = 0, c = 0;
  for (int i = lo; i < hi; ++i) { s += blk[i].tsum; c += blk[i].bcnt; }
  ps[tid] = s; pc[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int a = tid >= off ? ps[tid - off] : 0, b = tid >= off ? pc[tid - off] : 0;
    __syncthreads();
    ps[tid] += a; pc[tid] += b;
    __syncthreads();
  }
  int rs = ps[tid] - s, rc = pc[tid] - c;
  long long m = -1;
  for (int i = lo; i < hi; ++i) {
    blk[i].toff = rs; blk[i].coff = rc;
    long long l = blk[i].last;
    if (l >= 0) { l = (l & ~0xFFFFFFFFll) | (unsigned)((int)(l & 0xFFFFFFFFll) + rs); blk[i].last = l; m = max(m, l); }
    rs += blk[i].tsum; rc += blk[i].bcnt;
  }
  pm[tid] = m;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    long long a = tid >= off ? pm[tid - off] : -1;
    __syncthreads();
    pm[tid] = max(pm[tid], a);
    __syncthreads();
  }
  long long run = tid ? pm[tid - 1] : -1;
  for (int i = lo; i < hi; ++i) {
    blk[i].prev = run;
    run = max(run, blk[i].last);
  }
  if (tid == 1023) {
    res->n_pos = ps[1023];
    res->n_neg = n - ps[1023];
    res->n_distinct = pc[1023];
  }
}

__global__ __launch_bounds__(MT_THREADS) void curve_points_kernel(const float* __restrict__ s,
                                                                   const uint32_t* __restrict__ lab, int n,
                                                                   const CurveBatch cb, CurveBlock* __restrict__ blk,
                                                                   const CurveResultDev* __restrict__ res,
                                                                   int* __restrict__ ctps, int* __restrict__ cfps,
                                                                   float* __restrict__ cthr) {
  __shared__ int shi[4];
  __shared__ long long shl[4];
  __shared__ double shd[4];
  const int cls = cb.cls[blockIdx.y], negate = (int)((cb.negate >> blockIdx.y) & 1ull);
  s += blockIdx.y * cb.sstride; lab += blockIdx.y * cb.lstride; blk += (size_t)blockIdx.y * gridDim.x; res += blockIdx.y;
  if (blockIdx.y != 0) ctps = nullptr;                        // the curve arrays belong to problem 0
  TileScan ts;
  tile_load(s, lab, n, cls, negate, ts);
  const CurveBlock b = blk[blockIdx.x];
  const long long P = res->n_pos, N = res->n_neg;
  int tot;
  const int excl = block_excl_sum(ts.tsum, shi, tot) + b.toff;        // global positives before this thread
  const int cexcl = block_excl_sum(__popc(ts.ends), shi, tot) + b.coff;
  const long long i0 = (long long)blockIdx.x * MT_TILE + threadIdx.x * MT_ITEMS;
  long long mine = -1;
  if (ts.ends) {
    const int it = 31 - __clz(ts.ends);
    mine = ((i0 + it) << 32) | (unsigned)(excl + ts.tp[it]);
  }
  long long dummy;
  long long prev = max(block_excl_max(mine, shl, dummy), b.prev);
  long long auc2 = 0, jval = (long long)0x8000000000000000ll, jidx = -1;
  double ap = 0.0;
  int c = cexcl;
#pragma unroll
  for (int it = 0; it < MT_ITEMS; ++it) {
    if (ts.ends & (1u << it)) {
      const long long i = i0 + it;
      const long long tp = excl + ts.tp[it], fp = i + 1 - tp;
      const long long pi = prev >> 32;                       // -1 -> (tpp, fpp) = (0, 0)
      const long long tpp = prev < 0 ? 0 : (prev & 0xFFFFFFFFll), fpp = prev < 0 ? 0 : pi + 1 - tpp;
      auc2 += (fp - fpp) * (tp + tpp);
      ap += (double)(tp - tpp) * ((double)tp / (double)(tp + fp));
      const long long j = tp * N - fp * P;
      if (j > jval) { jval = j; jidx = i; }
      if (ctps) { ctps[c] = (int)tp; cfps[c] = (int)fp; cthr[c] = s[i]; }
      ++c;
      prev = (i << 32) | (unsigned)tp;
    }
  }
  // block reductions in a fixed order (lane tree, then waves 0..3)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    auc2 += __shfl_down(auc2, off, 64);
    ap += __shfl_down(ap, off, 64);
    const long long oj = __shfl_down(jval, off, 64), oi = __shfl_down(jidx, off, 64);
    if (oj > jval || (oj == jval && oi >= 0 && (jidx < 0 || oi < jidx))) { jval = oj; jidx = oi; }
  }
  __shared__ long long sj[4], si[4];
  __syncthreads();
  if (lane == 0) { shl[w] = auc2; shd[w] = ap; sj[w] = jval; si[w] = jidx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long a = 0; double p = 0.0; long long bj = (long long)0x8000000000000000ll, bi = -1;
    for (int i = 0; i < 4; ++i) {
      a += shl[i]; p += shd[i];
      if (sj[i] > bj || (sj[i] == bj && si[i] >= 0 && (bi < 0 || si[i] < bi))) { bj = sj[i]; bi = si[i]; }
    }
    blk[blockIdx.x].auc2 = a; blk[blockIdx.x].ap = p; blk[blockIdx.x].jval = bj; blk[blockIdx.x].jidx = bi;
  }
}

__global__ __launch_bounds__(256) void curve_final_kernel(const float* __restrict__ s, const CurveBlock* __restrict__ blk,
                                                           int nb, CurveResultDev* __restrict__ res, size_t sstride) {
  __shared__ long long sa[256], sj[256], si[256];
  __shared__ double sp[256];
  const int tid = threadIdx.x;
  s += blockIdx.x * sstride; blk += (size_t)blockIdx.x * nb; res += blockIdx.x;      // blockIdx.x = problem
  long long a = 0, bj = (long long)0x8000000000000000ll, bi = -1;
  double p = 0.0;
  for (int i = tid; i < nb; i += 256) {
    a += blk[i].auc2; p += blk[i].ap;
    const long long oj = blk[i].jval, oi = blk[i].jidx;
    if (oi >= 0 && (oj > bj || (oj == bj && (bi < 0 || oi < bi)))) { bj = oj; bi = oi; }
  }
  sa[tid] = a; sp[tid] = p; sj[tid] = bj; si[tid] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      sa[tid] += sa[tid + off]; sp[tid] += sp[tid + off];
      const long long oj = sj[tid + off], oi = si[tid + off];
      if (oi >= 0 && (oj > sj[tid] || (oj == sj[tid] && (si[tid] < 0 || oi < si[tid])))) { sj[tid] = oj; si[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const long long P = res->n_pos, N = res->n_neg;
    res->auroc = (P > 0 && N > 0) ? 0.5 * (double)sa[0] / ((double)P * (double)N) : 0.0;
    res->ap = P > 0 ? sp[0] / (double)P : __longlong_as_double(0x7ff8000000000000ll);
    // the prepended (0,0) ROC point (value 0, threshold 1.0) wins unless a real point is strictly better
    // (with an absent class torchmetrics zeroes the curve: every point ties at 0 and the first one wins)
    if (si[0] >= 0 && sj[0] > 0 && P > 0 && N > 0) { res->opt_index = si[0]; res->opt_threshold = s[si[0]]; }
    else { res->opt_index = -1; res->opt_threshold = 1.0f; }
    res->pad = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ counts
// counts layout (int64): top1_hit[C] | top5_hit[C] | class_n[C] | confusion[C*C] (row = true class) |
//                        f1_tp[10] | f1_fp[10] | f1_fn[10]
__global__ __launch_bounds__(256) void test_counts_kernel(const float* __restrict__ scores,
                                                           const float* __restrict__ probs,
                                                           const long long* __restrict__ labels, int n, int C,
                                                           int normal_idx, const float* __restrict__ thr_dev,
                                                           int* __restrict__ y_pred,
                                                           unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int lc[];
  const int ncnt = 3 * C + C * C + 30;
  for (int i = threadIdx.x; i < ncnt; i += blockDim.x) lc[i] = 0;
  __syncthreads();
  const float thr = *thr_dev;
  const int Cm = C - 1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float sc = scores[i];
    const int lab = (int)labels[i];
    const float* p = probs + (size_t)i * Cm;
    int am = 0;
    float best = p[0];
    for (int j = 1; j < Cm; ++j) {
      const float v = p[j];
      if (v > best) { best = v; am = j; }
    }
    if (am >= normal_idx) ++am;
    const int yp = sc < thr ? normal_idx : am;                         // :538-547
    y_pred[i] = yp;
    // rank of the true class among the non-normal probabilities (ties: lower index first)
    int rank = Cm;
    if (lab != normal_idx) {
      const int l = lab > normal_idx ? lab - 1 : lab;
      const float pl = p[l];
      rank = 0;
      for (int j = 0; j < Cm; ++j) rank += (p[j] > pl || (p[j] == pl && j < l)) ? 1 : 0;
    }
    const bool pn = yp == normal_idx;
    const bool hit5 = pn ? (lab == normal_idx || rank < 4) : (rank < 5);       // :556-572
    atomicAdd(&lc[lab], yp == lab ? 1u : 0u);
    atomicAdd(&lc[C + lab], hit5 ? 1u : 0u);
    atomicAdd(&lc[2 * C + lab], 1u);
    atomicAdd(&lc[3 * C + lab * C + yp], 1u);
    const bool pos = lab != normal_idx;
    unsigned int* f1 = lc + 3 * C + C * C;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const bool pb = !(sc < (float)((double)(t + 1) / 10.0));         // torch.where(scores < thresh, 0, 1), :624-625
      if (pb && pos) atomicAdd(&f1[t], 1u);
      if (pb && !pos) atomicAdd(&f1[10 + t], 1u);
      if (!pb && pos) atomicAdd(&f1[20 + t], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ncnt; i += blockDim.x)
    if (lc[i]) atomicAdd(&counts[i], (unsigned long long)lc[i]);
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int64_t acx_sort_workspace_bytes(int64_t n) {
  if (n < 0) return -1;
  const int64_t nb = (n + MT_TILE - 1) / MT_TILE;
  return 2 * n * 4 + 256 * (nb > 0 ? nb : 1) * 4 + 256 * 4 + 256;
}

// `batch` independent sorts of n pairs each in ONE launch sequence (12 launches for any batch): keys of problem b at
// keys + b * key_stride, payloads at vals + b * val_stride (val_stride = 0: every problem sorts the SAME payload array, the
// metrics epilogue's one-vs-rest curves all carry the label vector), outputs dense [batch][n].  The workspace is `batch`
// times acx_sort_workspace_bytes(n).
extern "C" int acx_sort_pairs_batched(acx_ctx* ctx, const float* keys, int64_t key_stride, const uint32_t* vals,
                                      int64_t val_stride, float* keys_out, uint32_t* vals_out, int64_t n, int32_t batch,
                                      int32_t descending, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: null context");
  if (n < 0 || n >= (1ll << 31) - MT_TILE) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: %sn=%ld out of range", "", (long)n);
  if (batch < 0 || batch > 65535) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: %sbatch=%ld out of range", "", (long)batch);
  if (n == 0 || batch == 0) return ACX_OK;
  if (!keys || !vals || !keys_out || !vals_out || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: null buffer");
  if (key_stride < n || (val_stride != 0 && val_stride < n)) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: stride < n%s", "");
  if ((const void*)keys == (void*)keys_out || (const void*)vals == (void*)vals_out)
    return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs_batched: in-place sort is not supported");
  if (workspace_bytes < (int64_t)batch * acx_sort_workspace_bytes(n))
    return acx_fail(ctx, ACX_E_WORKSPACE, "acx_sort_pairs_batched: %sworkspace %ld < %ld bytes", "", (long)workspace_bytes,
                    (long)((int64_t)batch * acx_sort_workspace_bytes(n)));
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n + MT_TILE - 1) / MT_TILE);
  uint32_t* tk = (uint32_t*)workspace;                       // [batch][n]
  uint32_t* tv = tk + (size_t)batch * n;                     // [batch][n]
  uint32_t* hist = tv + (size_t)batch * n;                   // [batch][256][nb]
  uint32_t* tot = hist + (size_t)batch * 256 * nb;           // [batch][256]
  // ping-pong: in -> tmp -> out -> tmp -> out
  const uint32_t* sk[4] = {(const uint32_t*)keys, tk, (uint32_t*)keys_out, tk};
  const uint32_t* sv[4] = {vals, tv, vals_out, tv};
  uint32_t* dk[4] = {tk, (uint32_t*)keys_out, tk, (uint32_t*)keys_out};
  uint32_t* dv[4] = {tv, vals_out, tv, vals_out};
  AcxProfScope prof(ctx, ACX_K_OTHER, st);
  for (int p = 0; p < 4; ++p) {
    const size_t ks = p == 0 ? (size_t)key_stride : (size_t)n, vs = p == 0 ? (size_t)val_stride : (size_t)n;
    radix_hist_kernel<<<dim3(nb, batch), MT_THREADS, 0, st>>>(sk[p], hist, (int)n, nb, 8 * p, p == 0, descending, ks);
    scan_rows_kernel<<<dim3(256, batch), 256, 0, st>>>(hist, tot, nb);
    radix_scatter_kernel<<<dim3(nb, batch), MT_THREADS, 0, st>>>(sk[p], sv[p], dk[p], dv[p], hist, tot, (int)n, nb, 8 * p,
                                                                 p == 0, p == 3, descending, ks, vs, (size_t)n);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return acx_fail(ctx, ACX_E_HIP, "acx_sort_pairs_batched: %s", hipGetErrorString(e));
  return ACX_OK;
}

extern "C" int acx_sort_pairs(acx_ctx* ctx, const float* keys, const uint32_t* vals, float* keys_out,
                              uint32_t* vals_out, int64_t n, int32_t descending, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs: null context");
  if (n < 0 || n >= (1ll << 31) - MT_TILE) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs: %sn=%ld out of range", "", (long)n);
  if (n == 0) return ACX_OK;
  if (!keys || !vals || !keys_out || !vals_out || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs: null buffer");
  if ((const void*)keys == (void*)keys_out || (const void*)vals == (void*)vals_out)
    return acx_fail(ctx, ACX_E_BADARG, "acx_sort_pairs: in-place sort is not supported");
  if (workspace_bytes < acx_sort_workspace_bytes(n))
    return acx_fail(ctx, ACX_E_WORKSPACE, "acx_sort_pairs: %sworkspace %ld < %ld bytes", "", (long)workspace_bytes,
                    (long)acx_sort_workspace_bytes(n));
  return acx_sort_pairs_batched(ctx, keys, n, vals, n, keys_out, vals_out, n, 1, descending, workspace, workspace_bytes, stream);
}

extern "C" int64_t acx_clf_curve_workspace_bytes(int64_t n) {
  if (n < 0) return -1;
  const int64_t nb = (n + MT_TILE - 1) / MT_TILE;
  return (nb > 0 ? nb : 1) * (int64_t)sizeof(CurveBlock);
}

// `batch` curves in one launch sequence (4 launches for any batch <= 64): problem b reads sorted_scores + b * score_stride and
// sorted_labels + b * label_stride with target = (label == cls[b]) or, where negate[b] != 0, (label != cls[b]); results[b] is its
// record.  The optional curve arrays (all three or none) receive the points of problem 0 only.
extern "C" int acx_clf_curve_batched(acx_ctx* ctx, const float* sorted_scores, int64_t score_stride,
                                     const uint32_t* sorted_labels, int64_t label_stride, int64_t n, int32_t batch,
                                     const int32_t* cls, const int32_t* negate, acx_curve_result* results, int32_t* curve_tps,
                                     int32_t* curve_fps, float* curve_thresholds, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  static_assert(sizeof(CurveResultDev) == sizeof(acx_curve_result), "acx_curve_result layout");
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve_batched: null context");
  if (n <= 0 || n >= (1ll << 31) - MT_TILE) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve_batched: %sn=%ld out of range", "", (long)n);
  if (batch < 1 || batch > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve_batched: %sbatch=%ld not in 1..64", "", (long)batch);
  if (!sorted_scores || !sorted_labels || !cls || !negate || !results || !workspace)
    return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve_batched: null buffer");
  if ((curve_tps != nullptr) != (curve_fps != nullptr) || (curve_tps != nullptr) != (curve_thresholds != nullptr))
    return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve_batched: pass all three curve arrays or none");
  if (workspace_bytes < (int64_t)batch * acx_clf_curve_workspace_bytes(n))
    return acx_fail(ctx, ACX_E_WORKSPACE, "acx_clf_curve_batched: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n + MT_TILE - 1) / MT_TILE);
  CurveBlock* blk = (CurveBlock*)workspace;                   // [batch][nb]
  CurveResultDev* res = (CurveResultDev*)results;
  CurveBatch cb;
  memset(&cb, 0, sizeof(cb));
  for (int b = 0; b < batch; ++b) { cb.cls[b] = cls[b]; if (negate[b]) cb.negate |= 1ull << b; }
  cb.sstride = (size_t)score_stride; cb.lstride = (size_t)label_stride;
  AcxProfScope prof(ctx, ACX_K_OTHER, st);
  curve_stats_kernel<<<dim3(nb, batch), MT_THREADS, 0, st>>>(sorted_scores, sorted_labels, (int)n, cb, blk);
  curve_prefix_kernel<<<batch, 1024, 0, st>>>(blk, nb, (long long)n, res);
  curve_points_kernel<<<dim3(nb, batch), MT_THREADS, 0, st>>>(sorted_scores, sorted_labels, (int)n, cb, blk, res, curve_tps,
                                                               curve_fps, curve_thresholds);
  curve_final_kernel<<<batch, 256, 0, st>>>(sorted_scores, blk, nb, res, (size_t)score_stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return acx_fail(ctx, ACX_E_HIP, "acx_clf_curve_batched: %s", hipGetErrorString(e));
  return ACX_OK;
}

extern "C" int acx_clf_curve(acx_ctx* ctx, const float* sorted_scores, const uint32_t* sorted_labels, int64_t n,
                             int32_t cls, int32_t negate, acx_curve_result* result, int32_t* curve_tps,
                             int32_t* curve_fps, float* curve_thresholds, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve: null context");
  if (n <= 0 || n >= (1ll << 31) - MT_TILE) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve: %sn=%ld out of range", "", (long)n);
  if (!sorted_scores || !sorted_labels || !result || !workspace) return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve: null buffer");
  if ((curve_tps != nullptr) != (curve_fps != nullptr) || (curve_tps != nullptr) != (curve_thresholds != nullptr))
    return acx_fail(ctx, ACX_E_BADARG, "acx_clf_curve: pass all three curve arrays or none");
  if (workspace_bytes < acx_clf_curve_workspace_bytes(n)) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_clf_curve: workspace too small");
  return acx_clf_curve_batched(ctx, sorted_scores, n, sorted_labels, n, n, 1, &cls, &negate, result, curve_tps, curve_fps,
                               curve_thresholds, workspace, workspace_bytes, stream);
}

extern "C" int acx_test_counts(acx_ctx* ctx, const float* scores, const float* probs, const int64_t* labels,
                               int64_t n, int32_t C, int32_t normal_idx, const float* threshold, int32_t* y_pred,
                               int64_t* counts, void* stream) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_test_counts: null context");
  if (n <= 0 || n >= (1ll << 31)) return acx_fail(ctx, ACX_E_BADARG, "acx_test_counts: %sn=%ld out of range", "", (long)n);
  if (C < 6 || C > 64 || normal_idx < 0 || normal_idx >= C)
    return acx_fail(ctx, ACX_E_BADARG, "acx_test_counts: need 6 <= C <= 64 (top-5 over C-1 classes), %sgot C=%ld", "", (long)C);
  if (!scores || !probs || !labels || !threshold || !y_pred || !counts) return acx_fail(ctx, ACX_E_BADARG, "acx_test_counts: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int ncnt = 3 * C + C * C + 30;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)ncnt * 8, st);
  if (e != hipSuccess) return acx_fail(ctx, ACX_E_HIP, "acx_test_counts: memset: %s", hipGetErrorString(e));
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  AcxProfScope prof(ctx, ACX_K_OTHER, st);
  test_counts_kernel<<<blocks, 256, (size_t)ncnt * 4, st>>>(scores, probs, (const long long*)labels, (int)n, C, normal_idx,
                                                             threshold, y_pred, (unsigned long long*)counts);
  e = hipGetLastError();
  if (e != hipSuccess) return acx_fail(ctx, ACX_E_HIP, "acx_test_counts: %s", hipGetErrorString(e));
  return ACX_OK;
}
