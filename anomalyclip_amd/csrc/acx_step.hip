// Support kernels of the whole-step training graph (anomalyclip_amd/components/step_graph.py): everything a data-parallel
// rank's optimisation step needs besides the forward / backward kernels, so that NO torch kernel runs between the first
// and the last launch of a step and the whole step replays from HIP graphs:
//   acx_prep_multi          every derived weight layout of the temporal model (q|kv concatenation, transposes for the dX
//                           GEMMs, flipped-tap [Cin][tap][Cout] conv weights, padded projection weight, positional tables)
//                           in ONE launch -- was ~20 copy / cat / transpose launches per step
//   acx_adamw_multi_dev     multi-tensor AdamW whose per-tensor scalars (1 - lr wd, lr / bias correction) live in DEVICE
//                           memory: lr schedules and the step count change without re-capturing the graph
//   acx_multi_copy          y_i = x_i for a list of tensors (gradients that cannot be produced in place -> flat buffer)
//   acx_bn_pack / acx_bn_running_update / acx_fill_f32   the selector's BatchNorm bookkeeping without torch elementwise ops
#include "acx_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ strided copy / transpose
constexpr int PREP_MAX_SEG = 72;
struct PrepSegs {
  const float* src[PREP_MAX_SEG];
  float* dst[PREP_MAX_SEG];
  int rows[PREP_MAX_SEG], cols[PREP_MAX_SEG];       // extent of the SOURCE block
  int src_ld[PREP_MAX_SEG], dst_ld[PREP_MAX_SEG];
  int tile0[PREP_MAX_SEG + 1];                      // first 32x32 tile of every segment
  unsigned char transpose[PREP_MAX_SEG];
  int nseg;
};

// one 32x32 tile per block: dst[r][c] = src[r][c] (copy) or dst[c][r] = src[r][c] (transpose), both sides coalesced
__global__ __launch_bounds__(256) void prep_multi_kernel(const PrepSegs t) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int lo = 0, hi = t.nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.tile0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int R = t.rows[lo], Cn = t.cols[lo];
  const int tc = (Cn + 31) >> 5;
  const int tl = b - t.tile0[lo];
  const int by = (tl / tc) * 32, bx = (tl % tc) * 32;
  const float* __restrict__ src = t.src[lo];
  float* __restrict__ dst = t.dst[lo];
  const int sld = t.src_ld[lo], dld = t.dst_ld[lo];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (!t.transpose[lo]) {
    for (int j = ty; j < 32; j += 8)
      if (by + j < R && bx + tx < Cn) dst[(size_t)(by + j) * dld + bx + tx] = src[(size_t)(by + j) * sld + bx + tx];
    return;
  }
  for (int j = ty; j < 32; j += 8)
    if (by + j < R && bx + tx < Cn) tile[j][tx] = src[(size_t)(by + j) * sld + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < Cn && by + tx < R) dst[(size_t)(bx + j) * dld + by + tx] = tile[tx][j];
}

// ------------------------------------------------------------------------------------------------ multi-tensor copy
constexpr int COPY_MAX_SEG = 96;
struct CopySegs {
  float* y[COPY_MAX_SEG];
  const float* x[COPY_MAX_SEG];
  long long n[COPY_MAX_SEG];
  int chunk0[COPY_MAX_SEG + 1];
  int nseg;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopySegs t) {
  const int b = blockIdx.x;
  int lo = 0, hi = t.nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.chunk0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  float* __restrict__ y = t.y[lo];
  const float* __restrict__ x = t.x[lo];
  const long long n = t.n[lo];
  const long long base = (long long)(b - t.chunk0[lo]) * 1024 + threadIdx.x;
  float xv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) xv[k] = x[i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) y[i] = xv[k];
  }
}

// ------------------------------------------------------------------------------------------------ AdamW, device-side scalars
constexpr int ADAMW_DEV_MAX_SEG = 48;
struct AdamwDevSegs {
  float* p[ADAMW_DEV_MAX_SEG];
  const float* g[ADAMW_DEV_MAX_SEG];
  float* m[ADAMW_DEV_MAX_SEG];
  float* v[ADAMW_DEV_MAX_SEG];
  long long n[ADAMW_DEV_MAX_SEG];
  int chunk0[ADAMW_DEV_MAX_SEG + 1];
  int nseg, seg_base;                 // hyper index of segment k = seg_base + k
};
// hyper: [0] = sqrt(1 - beta2^step), then per tensor (decay = 1 - lr wd, step_size = lr / (1 - beta1^step)) pairs; the same
// f32 values acx_adamw_multi forms on the host, so both entry points update bit-identically.  grad_scale multiplies every
// gradient as it is read (1 / world size after a summing all-reduce; exactly 1.0f otherwise -- x * 1.0f == x).
template <bool SCALE>
__global__ __launch_bounds__(256) void adamw_multi_dev_kernel(const AdamwDevSegs t, const float* __restrict__ hyper, float b1, float b2,
                                                              float omb1, float omb2, float eps, float grad_scale) {
  const int b = blockIdx.x;
  int lo = 0, hi = t.nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.chunk0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int sgi = lo;
  float* __restrict__ p = t.p[sgi];
  float* __restrict__ g = const_cast<float*>(t.g[sgi]);
  float* __restrict__ m = t.m[sgi];
  float* __restrict__ v = t.v[sgi];
  const long long n = t.n[sgi];
  const float bc2_sqrt = hyper[0];
  const float decay = hyper[1 + 2 * (t.seg_base + sgi)], step_size = hyper[2 + 2 * (t.seg_base + sgi)];
  const long long base = (long long)(b - t.chunk0[sgi]) * 1024 + threadIdx.x;
  float pi[4], gi[4], mi[4], vi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) { pi[k] = p[i]; gi[k] = SCALE ? g[i] * grad_scale : g[i]; mi[k] = m[i]; vi[k] = v[i]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) {
      const float pw = pi[k] * decay;
      const float mn = b1 * mi[k] + omb1 * gi[k];
      const float vn = b2 * vi[k] + omb2 * gi[k] * gi[k];
      m[i] = mn;
      v[i] = vn;
      p[i] = pw - step_size * (mn / (sqrtf(vn) / bc2_sqrt + eps));
      if (SCALE) g[i] = gi[k];         // the flat gradient buffer is left holding the MEAN, as DDP leaves .grad
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused column sums
// out[c] = sum_r x[r][c] in ONE launch, deterministic: block (slab, column group of 64) writes its partial row, the LAST block
// of a column group to arrive (device-scope counter) adds the slabs' partials in slab order.  The counter returns to zero,
// so the launch is replayable from a HIP graph.  Replaces colsum_partials + reduce_rows (two launches per bias gradient).
// LANES float4 lanes per row slice (column group = 4 * LANES columns), 256 / LANES row slices per block: narrow matrices take
// narrow groups so that a launch still has >= 256 blocks x 8 rows in flight per lane (D = 512 at 64 slabs was 128 blocks on 128
// of the 256 CUs: 3.3 TB/s)
template <int LANES>
__device__ __forceinline__ void colsum_fused_body(const float* __restrict__ x, int ld, long long rows, int D, int rpb,
                                                  float* __restrict__ part, float* __restrict__ out,
                                                  unsigned int* __restrict__ counters, const int cg, const int slab, const int nslab,
                                                  const float beta) {
  // block = 4 LANES columns x NS row slices; slice k adds rows r0 + k, r0 + k + NS, ... in order, the slices are added in slice order
  constexpr int NS = 256 / LANES;
  __shared__ float4 red[NS][LANES];
  __shared__ bool last;
  const int q = threadIdx.x % LANES, sl = threadIdx.x / LANES;
  const int c = cg * (4 * LANES) + 4 * q;
  const long long r0 = (long long)slab * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < D) {
    long long r = r0 + sl;
    for (; r + 7 * NS < r1; r += 8 * NS) {         // eight rows in flight per lane (independent loads, fixed add order)
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(x + (r + NS * k) * ld + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
    }
    for (; r < r1; r += NS) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sl][q] = a;
  __syncthreads();
  // Publishing the partial row WITHOUT a release fence: `__threadfence()` at device scope writes back every dirty line of this
  // XCD's L2 -- after a fat GEMM that is megabytes, and it made this kernel take 33-124 us inside a training step.  The
  // partials go out as agent-scope (write-through, sc1) stores, `s_waitcnt vmcnt(0)` waits for their acknowledgement, the
  // arrival counter is a relaxed agent-scope atomic, and the last arriver reads the partials with agent-scope loads
  // (MI355X_MICROARCH.md, hand-off price list: publish-large / handoff-flag).  What orders what: the inline-asm wait carries a
  // "memory" clobber (a compiler barrier: no store may be moved below it, nothing that follows above it) and makes the wave
  // wait until every write-through store has been acknowledged by the fabric; __syncthreads() orders the workgroup's waves
  // around lane 0's counter increment; the last arriver's loads are issued after its fetch_add RETURNED (the `last` flag goes
  // through LDS and a barrier) and bypass the L1 (sc1).  The guide lists exactly this pairing ("sc1 payload -> asm vmcnt(0)
  // -> flag", sc1 loads on the reading side) among the valid hand-off forms of gfx950; it is not expressed in C++ memory
  // orders because a release at agent scope lowers to the L2 write-back above.  tests: the determinism soak
  // (profiles/r04_determinism_soak.txt, test_colsum_fused_one_launch) runs it under uneven load.
  if (sl == 0 && c < D) {
    float4 t = red[0][q];
#pragma unroll
    for (int k = 1; k < NS; ++k) { t.x += red[k][q].x; t.y += red[k][q].y; t.z += red[k][q].z; t.w += red[k][q].w; }
    float* dst = part + (size_t)slab * D + c;
    __hip_atomic_store(dst + 0, t.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, t.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 2, t.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 3, t.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ACX_HANDOFF_RELEASE();
    last = __hip_atomic_fetch_add(&counters[cg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nslab - 1;
  }
  __syncthreads();
  if (!last) return;
  ACX_HANDOFF_ACQUIRE();
  // last block of the column group to arrive: LANES float4 columns x NS slab lanes; lane l adds slabs l, l + NS, ... in order
  // (four slabs = sixteen scalar loads in flight), the NS lane sums are then added in lane order -- a fixed tree
  {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) {
      int p = sl;
      for (; p + 3 * NS < nslab; p += 4 * NS) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[4 * k + j] = __hip_atomic_load(part + (size_t)(p + NS * k) * D + c + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0.x += v[4 * k]; s0.y += v[4 * k + 1]; s0.z += v[4 * k + 2]; s0.w += v[4 * k + 3]; }
      }
      for (; p < nslab; p += NS) {
        s0.x += __hip_atomic_load(part + (size_t)p * D + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s0.y += __hip_atomic_load(part + (size_t)p * D + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s0.z += __hip_atomic_load(part + (size_t)p * D + c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s0.w += __hip_atomic_load(part + (size_t)p * D + c + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    red[sl][q] = s0;
    __syncthreads();
    if (sl == 0 && c < D) {
      float4 t = red[0][q];
#pragma unroll
      for (int k = 1; k < NS; ++k) { t.x += red[k][q].x; t.y += red[k][q].y; t.z += red[k][q].z; t.w += red[k][q].w; }
      if (beta != 0.f) {
        const float4 o = *reinterpret_cast<const float4*>(out + c);
        t.x += beta * o.x; t.y += beta * o.y; t.z += beta * o.z; t.w += beta * o.w;
      }
      *reinterpret_cast<float4*>(out + c) = t;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(&counters[cg], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void colsum_fused_dispatch(int lanes, const float* __restrict__ x, int ld, long long rows, int D, int rpb,
                                                      float* __restrict__ part, float* __restrict__ out,
                                                      unsigned int* __restrict__ counters, int cg, int slab, int nslab, float beta) {
  if (lanes == 64) colsum_fused_body<64>(x, ld, rows, D, rpb, part, out, counters, cg, slab, nslab, beta);
  else if (lanes == 32) colsum_fused_body<32>(x, ld, rows, D, rpb, part, out, counters, cg, slab, nslab, beta);
  else colsum_fused_body<16>(x, ld, rows, D, rpb, part, out, counters, cg, slab, nslab, beta);
}

__global__ __launch_bounds__(256) void colsum_fused_kernel(const float* __restrict__ x, int ld, long long rows, int D, int rpb,
                                                           float* __restrict__ part, float* __restrict__ out,
                                                           unsigned int* __restrict__ counters, float beta, int lanes) {
  colsum_fused_dispatch(lanes, x, ld, rows, D, rpb, part, out, counters, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, beta);
}

// several column sums in one launch (the bias gradients of a backward pass): flat block index -> (problem, column group, slab)
constexpr int COLSUM_GROUP_MAX = 12;
struct ColsumGroup {
  const float* x[COLSUM_GROUP_MAX];
  float* part[COLSUM_GROUP_MAX];
  float* out[COLSUM_GROUP_MAX];
  long long rows[COLSUM_GROUP_MAX];
  int ld[COLSUM_GROUP_MAX], D[COLSUM_GROUP_MAX], rpb[COLSUM_GROUP_MAX], cgs[COLSUM_GROUP_MAX], nslab[COLSUM_GROUP_MAX];
  int lanes[COLSUM_GROUP_MAX];
  int ctr0[COLSUM_GROUP_MAX];                       // first arrival counter of the problem
  int blk0[COLSUM_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void colsum_fused_group_kernel(const ColsumGroup G, unsigned int* __restrict__ counters) {
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < COLSUM_GROUP_MAX; ++i)
    if (i < G.n && G.blk0[i] <= b) k = i;
  const int local = b - G.blk0[k];
  const int cgs = G.cgs[k];
  const int slab = local / cgs;
  colsum_fused_dispatch(G.lanes[k], G.x[k], G.ld[k], G.rows[k], G.D[k], G.rpb[k], G.part[k], G.out[k], counters + G.ctr0[k],
                        local - slab * cgs, slab, G.nslab[k], 0.f);
}

// out_k[c] = sum_p part_k[p][c] for several partial tables in one launch (LayerNorm / classifier parameter gradients): fixed
// order, the same tree as reduce_rows_kernel (16 slices of the parts, then the slices in order)
constexpr int RR_GROUP_MAX = 12;
struct ReduceRowsGroup {
  const float* part[RR_GROUP_MAX];
  float* out[RR_GROUP_MAX];
  int nparts[RR_GROUP_MAX], width[RR_GROUP_MAX];
  int blk0[RR_GROUP_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void reduce_rows_group_kernel(const ReduceRowsGroup G) {
  __shared__ float red[16][17];
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < RR_GROUP_MAX; ++i)
    if (i < G.n && G.blk0[i] <= b) k = i;
  const float* __restrict__ part = G.part[k];
  const int nparts = G.nparts[k], width = G.width[k];
  const int ci = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = (b - G.blk0[k]) * 16 + ci;
  float s = 0.f;
  if (c < width) {
    int p = sl;
    for (; p + 112 < nparts; p += 128) {            // eight loads in flight per thread, added in part order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + 16 * u) * width + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nparts; p += 16) s += part[(size_t)p * width + c];
  }
  red[sl][ci] = s;
  __syncthreads();
  if (sl == 0 && c < width) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][ci];
    G.out[k][c] = t;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm bookkeeping
// SyncBN payload of a rank: out = [mean (C) | biased var * rows (C) | rows]   (parallel.sync_bn_stats' torch.cat)
__global__ void bn_pack_kernel(const float* __restrict__ mean, const float* __restrict__ var_b, float rows, int C,
                               float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < C) { out[i] = mean[i]; out[C + i] = var_b[i] * rows; }
  if (i == 0) out[2 * C] = rows;
}
// nn.BatchNorm1d's running statistics: r = (1 - momentum) r + momentum x (selector_model.py:30,65), batch counter + 1
__global__ void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var_u, float* __restrict__ rm,
                                  float* __restrict__ rv, long long* __restrict__ nbt, int C, float momentum, float om) {
  const int i = threadIdx.x;
  if (i < C) {
    rm[i] = acx_bn_running(momentum, om, mean[i], rm[i]);
    rv[i] = acx_bn_running(momentum, om, var_u[i], rv[i]);
  }
  if (i == 0 && nbt) *nbt += 1;
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}

}  // namespace

static inline void colsum_geometry(int64_t rows, int32_t D, int* cgs, int* nslab, int* rpb, int* lanes) {
  // at most 64 slabs of >= 64 rows: the last arriver walks the slabs' partials with memory-latency-bound batches (~1 us per
  // four slabs per lane), so the slab count, not the block count, sets the kernel's tail (512 slabs: 32 batches = 30 us);
  // the column groups narrow (256 -> 128 -> 64 columns) until the launch has 256 blocks
  long long ns = rows / 64;
  if (ns > 64) ns = 64;
  if (ns < 1) ns = 1;
  const int r = (int)((rows + ns - 1) / ns);
  const int nsl = (int)((rows + r - 1) / r);
  int ln = 64;
  while (ln > 16 && (long long)((D + 4 * ln - 1) / (4 * ln)) * nsl < 256) ln >>= 1;
  *lanes = ln; *cgs = (D + 4 * ln - 1) / (4 * ln); *rpb = r; *nslab = nsl;
}

extern "C" int acx_prep_multi(acx_ctx* ctx, int32_t nseg, const acx_prep_seg* segs, void* stream) {
  if (nseg <= 0) return ACX_OK;
  if (!segs) return acx_fail(ctx, ACX_E_BADARG, "acx_prep_multi: null segment table%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nseg) {
    PrepSegs t;
    memset(&t, 0, sizeof(t));
    long long tiles = 0;
    int k = 0;
    for (; i < nseg && k < PREP_MAX_SEG; ++i) {
      const acx_prep_seg& g = segs[i];
      if (g.rows <= 0 || g.cols <= 0) continue;
      if (!g.src || !g.dst) return acx_fail(ctx, ACX_E_BADARG, "acx_prep_multi: null tensor pointer%s");
      if (g.src_ld < g.cols || g.dst_ld < (g.transpose ? g.rows : g.cols))
        return acx_fail(ctx, ACX_E_BADARG, "acx_prep_multi: leading dimension smaller than the row length%s");
      t.src[k] = (const float*)g.src; t.dst[k] = (float*)g.dst;
      t.rows[k] = g.rows; t.cols[k] = g.cols; t.src_ld[k] = g.src_ld; t.dst_ld[k] = g.dst_ld;
      t.transpose[k] = g.transpose ? 1 : 0;
      t.tile0[k] = (int)tiles;
      tiles += (long long)((g.rows + 31) / 32) * ((g.cols + 31) / 32);
      ++k;
    }
    if (k == 0) continue;
    if (tiles > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_prep_multi: too many tiles for one launch%s");
    t.tile0[k] = (int)tiles;
    t.nseg = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(prep_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, s, t);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_prep_multi");
  return ACX_OK;
}

extern "C" int acx_multi_copy(acx_ctx* ctx, int32_t nseg, void* const* y, const void* const* x, const int64_t* n, void* stream) {
  if (nseg <= 0) return ACX_OK;
  if (!y || !x || !n) return acx_fail(ctx, ACX_E_BADARG, "acx_multi_copy: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nseg) {
    CopySegs t;
    memset(&t, 0, sizeof(t));
    long long chunks = 0;
    int k = 0;
    for (; i < nseg && k < COPY_MAX_SEG; ++i) {
      if (n[i] <= 0) continue;
      if (!y[i] || !x[i]) return acx_fail(ctx, ACX_E_BADARG, "acx_multi_copy: null tensor pointer%s");
      t.y[k] = (float*)y[i]; t.x[k] = (const float*)x[i]; t.n[k] = n[i];
      t.chunk0[k] = (int)chunks;
      chunks += (n[i] + 1023) / 1024;
      ++k;
    }
    if (k == 0) continue;
    if (chunks > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_multi_copy: too many elements for one launch%s");
    t.chunk0[k] = (int)chunks;
    t.nseg = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_multi_copy");
  return ACX_OK;
}

extern "C" int acx_adamw_hyper(int32_t nseg, const double* lr, const double* weight_decay, double beta1, double beta2, int32_t step,
                               float* out) {
  if (nseg < 0 || !out || (nseg > 0 && (!lr || !weight_decay)) || step <= 0) return ACX_E_BADARG;
  // the scalar terms exactly as acx_adamw_multi forms them: f64 on the host, rounded ONCE (torch.optim.AdamW's Python floats)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  out[0] = (float)sqrt(1.0 - pow(beta2, (double)step));
  for (int i = 0; i < nseg; ++i) {
    out[1 + 2 * i] = (float)(1.0 - lr[i] * weight_decay[i]);
    out[2 + 2 * i] = (float)(lr[i] / bc1);
  }
  return ACX_OK;
}

extern "C" int acx_adamw_multi_dev(acx_ctx* ctx, int32_t nseg, void* const* p, const void* const* g, void* const* m, void* const* v,
                                   const int64_t* n, const float* hyper_dev, float grad_scale, double beta1, double beta2, double eps,
                                   void* stream) {
  if (nseg <= 0) return ACX_OK;
  if (!p || !g || !m || !v || !n || !hyper_dev) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw_multi_dev: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nseg) {
    AdamwDevSegs t;
    memset(&t, 0, sizeof(t));
    long long chunks = 0;
    int k = 0;
    t.seg_base = i;
    for (; i < nseg && k < ADAMW_DEV_MAX_SEG; ++i) {
      // hyper is indexed by the CALLER's tensor index, so empty tensors are kept as zero-chunk segments, not skipped
      if (n[i] > 0 && (!p[i] || !g[i] || !m[i] || !v[i]))
        return acx_fail(ctx, ACX_E_BADARG, "acx_adamw_multi_dev: null tensor pointer%s");
      t.p[k] = (float*)p[i]; t.g[k] = (const float*)g[i]; t.m[k] = (float*)m[i]; t.v[k] = (float*)v[i];
      t.n[k] = n[i] > 0 ? n[i] : 0;
      t.chunk0[k] = (int)chunks;
      chunks += (t.n[k] + 1023) / 1024;
      ++k;
    }
    if (chunks == 0) continue;
    if (chunks > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_adamw_multi_dev: too many elements for one launch%s");
    // a zero-chunk segment shares its chunk0 with the next one: the search ("last segment with chunk0 <= b") then lands on the
    // LAST of the equal entries, which is the non-empty one -- unless the empty ones trail; cut those off
    while (k > 0 && t.n[k - 1] == 0) --k;
    t.chunk0[k] = (int)chunks;
    t.nseg = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    if (grad_scale == 1.0f)
      hipLaunchKernelGGL(adamw_multi_dev_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, s, t, hyper_dev, (float)beta1,
                         (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, grad_scale);
    else
      hipLaunchKernelGGL(adamw_multi_dev_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, s, t, hyper_dev, (float)beta1,
                         (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, grad_scale);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_adamw_multi_dev");
  return ACX_OK;
}

extern "C" int acx_colsum_fused(acx_ctx* ctx, const float* x, int32_t ld, int64_t rows, int32_t D, float* out, float* part,
                                size_t part_bytes, uint32_t* counters, float beta, void* stream) {
  if (!x || !out || !part || !counters) return acx_fail(ctx, ACX_E_BADARG, "acx_colsum_fused: null pointer%s");
  if (rows <= 0 || D <= 0) return ACX_OK;
  if (D % 4 || ld % 4 || D > 256 * 256 || (((uintptr_t)x | (uintptr_t)part) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_colsum_fused: D / ld multiples of 4, D <= 16384, 16-byte aligned x / part%s");
  // slabs: enough blocks to fill the chip a few times over, at least 64 rows each
  int cgs, nslab_i, rpb, lanes;
  colsum_geometry(rows, D, &cgs, &nslab_i, &rpb, &lanes);
  long long nslab = nslab_i;
  if ((size_t)nslab * D * sizeof(float) > part_bytes) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_colsum_fused: partial buffer too small%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(colsum_fused_kernel, dim3((unsigned)cgs, (unsigned)nslab), dim3(256), 0, (hipStream_t)stream, x, ld,
                     (long long)rows, D, rpb, part, out, counters, beta, lanes);
  ACX_CHECK_LAUNCH(ctx, "acx_colsum_fused");
  return ACX_OK;
}

extern "C" int acx_colsum_fused_group(acx_ctx* ctx, int32_t nprob, const void* const* x, const int32_t* ld, const int64_t* rows,
                                      const int32_t* D, void* const* out, void* const* part, uint32_t* counters, int32_t ncounters,
                                      void* stream) {
  if (nprob <= 0) return ACX_OK;
  if (!x || !ld || !rows || !D || !out || !part || !counters) return acx_fail(ctx, ACX_E_BADARG, "acx_colsum_fused_group: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nprob) {
    ColsumGroup G;
    memset(&G, 0, sizeof(G));
    int k = 0, blocks = 0, ctr = 0;
    for (; i < nprob && k < COLSUM_GROUP_MAX; ++i) {
      if (rows[i] <= 0 || D[i] <= 0) continue;
      if (!x[i] || !out[i] || !part[i] || D[i] % 4 || ld[i] % 4 || (((uintptr_t)x[i] | (uintptr_t)part[i]) & 15))
        return acx_fail(ctx, ACX_E_BADARG, "acx_colsum_fused_group: D / ld multiples of 4, 16-byte aligned x / part%s");
      int cgs, nslab, rpb, lanes;
      colsum_geometry(rows[i], D[i], &cgs, &nslab, &rpb, &lanes);
      if (ctr + cgs > ncounters) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_colsum_fused_group: not enough arrival counters%s");
      G.x[k] = (const float*)x[i]; G.part[k] = (float*)part[i]; G.out[k] = (float*)out[i];
      G.rows[k] = rows[i]; G.ld[k] = ld[i]; G.D[k] = D[i]; G.rpb[k] = rpb; G.cgs[k] = cgs; G.nslab[k] = nslab; G.lanes[k] = lanes;
      G.ctr0[k] = ctr; ctr += cgs;
      G.blk0[k] = blocks; blocks += cgs * nslab;
      ++k;
    }
    if (k == 0) continue;
    G.blk0[k] = blocks; G.n = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(colsum_fused_group_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G, counters);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_colsum_fused_group");
  return ACX_OK;
}

extern "C" int acx_reduce_rows_group(acx_ctx* ctx, int32_t nprob, const void* const* part, void* const* out, const int32_t* nparts,
                                     const int32_t* width, void* stream) {
  if (nprob <= 0) return ACX_OK;
  if (!part || !out || !nparts || !width) return acx_fail(ctx, ACX_E_BADARG, "acx_reduce_rows_group: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nprob) {
    ReduceRowsGroup G;
    memset(&G, 0, sizeof(G));
    int k = 0, blocks = 0;
    for (; i < nprob && k < RR_GROUP_MAX; ++i) {
      if (nparts[i] <= 0 || width[i] <= 0) continue;
      if (!part[i] || !out[i]) return acx_fail(ctx, ACX_E_BADARG, "acx_reduce_rows_group: null tensor pointer%s");
      G.part[k] = (const float*)part[i]; G.out[k] = (float*)out[i]; G.nparts[k] = nparts[i]; G.width[k] = width[i];
      G.blk0[k] = blocks; blocks += (width[i] + 15) / 16;
      ++k;
    }
    if (k == 0) continue;
    G.blk0[k] = blocks; G.n = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(reduce_rows_group_kernel, dim3((unsigned)blocks), dim3(256), 0, s, G);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_reduce_rows_group");
  return ACX_OK;
}

extern "C" size_t acx_colsum_fused_part_bytes(int64_t rows, int32_t D) {
  if (rows <= 0 || D <= 0) return 0;
  int cgs, nslab, rpb, lanes;
  colsum_geometry(rows, D, &cgs, &nslab, &rpb, &lanes);
  return (size_t)(nslab + 1) * D * sizeof(float);
}

extern "C" int acx_bn_pack(acx_ctx* ctx, const float* mean, const float* var_biased, int64_t rows, int32_t C1, float* out,
                           void* stream) {
  if (!mean || !var_biased || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_pack: null pointer%s");
  if (C1 <= 0 || C1 > 1024) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_bn_pack: 1 <= C1 <= 1024%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_pack_kernel, dim3(1), dim3((unsigned)((C1 + 63) / 64 * 64)), 0, (hipStream_t)stream, mean, var_biased,
                     (float)rows, C1, out);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_pack");
  return ACX_OK;
}

extern "C" int acx_bn_running_update(acx_ctx* ctx, const float* mean, const float* var_unbiased, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, int32_t C1, float momentum, float one_minus,
                                     void* stream) {
  if (!mean || !var_unbiased || !running_mean || !running_var)
    return acx_fail(ctx, ACX_E_BADARG, "acx_bn_running_update: null pointer%s");
  if (C1 <= 0 || C1 > 1024) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_bn_running_update: 1 <= C1 <= 1024%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3((unsigned)((C1 + 63) / 64 * 64)), 0, (hipStream_t)stream, mean, var_unbiased,
                     running_mean, running_var, (long long*)num_batches_tracked, C1, momentum, one_minus);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_running_update");
  return ACX_OK;
}

extern "C" int acx_fill_f32(acx_ctx* ctx, float* p, int64_t n, float value, void* stream) {
  if (n <= 0) return ACX_OK;
  if (!p) return acx_fail(ctx, ACX_E_BADARG, "acx_fill_f32: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, (long long)n, value);
  ACX_CHECK_LAUNCH(ctx, "acx_fill_f32");
  return ACX_OK;
}
