// acx_gemm -- the MFMA workhorse:  C[M,N] = epilogue( amap(A)[M,K] . W[N,K]^T )
//
// Replaces every dense contraction on the AnomalyCLIP hot path (see include/acx.h for the
// reference call sites).  Both operands are K-contiguous (PyTorch nn.Linear stores W as [N,K]),
// which is exactly the MFMA A/B fragment orientation: a lane needs consecutive k of ONE row.
//
// CDNA4 design
//   * block tile 128x128, 4 wavefronts (2x2), each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32
//     accumulators (64 acc VGPRs/lane).  2 blocks per CU (LDS 2 x 72 KiB, <=128 VGPR).
//   * K-step = 128 BYTES of K per row for both precisions (32 f32 / 64 bf16), staged through LDS
//     rows of 128 B + 16 B pad (stride 144 B = 9 x 16-B slots -> the 16-lane groups of
//     ds_read_b128 and the 8-lane groups of ds_write_b128 are bank-conflict free).
//   * ACX_PREC_F32 : v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 157 TF roof).  One
//     ds_read_b128 feeds FOUR MFMAs per operand by permuting K inside the step: lane half h of
//     MFMA (q,j) supplies k = 4*(2q+h)+j for A and B alike (a dot product does not care about
//     the order of k as long as both operands agree).
//   * ACX_PREC_BF16: v_mfma_f32_32x32x16_bf16, one ds_read_b128 (8 bf16) per operand per MFMA.
//   * register-staged double buffering (guide T14): global loads of K-step t+1 are issued before
//     the MFMAs of step t and written to the other LDS buffer after them; one barrier per step.
//   * A-operand prologue fused into the staging pass: f32->bf16 conversion, re-centring
//     (a - ncentroid[k]), the 3x3-conv row gather (implicit GEMM over the (gn,gl) token grid,
//     zero padding) and the reference's test-mode tiling gather.
//   * epilogue fused: bias, QuickGELU / LeakyReLU, axial positional embedding, residual add,
//     f32 or bf16 store.
//   * XCD-aware 1-D grid: block b runs on XCD b%8; the bijective remap gives every XCD a
//     contiguous run of tiles (n fastest) so the A tile of a row of tiles stays in that XCD's L2.
#include "acx_internal.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;                // LDS row stride in bytes (128 data + 16 pad)
constexpr int TILE_B = BM * ROWB;        // 18432
constexpr int NTHREADS = 256;

struct Args {
  acx_gemm_desc d;
  int tiles_n;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ uint4 pack_bf16x8(float4 a, float4 b) {
  uint4 r;
  r.x = (uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16);
  r.y = (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16);
  r.z = (uint32_t)f2bf(b.x) | ((uint32_t)f2bf(b.y) << 16);
  r.w = (uint32_t)f2bf(b.z) | ((uint32_t)f2bf(b.w) << 16);
  return r;
}

// PREC: 0 f32 MFMA, 1 bf16 MFMA.  A_BF16: A stored as bf16 in global (PREC 1 only).
template <int PREC, int A_BF16, int C_BF16>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const acx_gemm_desc& d = g.d;
  constexpr int KE = PREC == 0 ? 32 : 64;     // K elements per step
  constexpr int CE = PREC == 0 ? 4 : 8;       // elements per 16-B LDS chunk

  // ---- XCD-aware tile assignment (bijective remap, guide T1)
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int tm = wg / g.tiles_n, tn = wg % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int t = threadIdx.x;
  const int chunk = t & 7, rbase = t >> 3;

  // ---- per-thread source rows for the 4 staged A rows and 4 staged W rows
  const int grid_sz = d.gn * d.gl;
  long a_row[4];        // identity/testtile: source row (or -1); conv: base row of the tile
  int a_n[4], a_l[4];   // conv: grid coordinates
  bool a_ok[4];
  const char* w_ptr[4];
  bool w_ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + rbase + 32 * r;
    a_ok[r] = m < d.M;
    a_row[r] = m;
    a_n[r] = a_l[r] = 0;
    if (d.amap == ACX_AMAP_TESTTILE) {
      const int per = grid_sz * d.seg;
      const int b = m / per, rem = m - b * per;
      const int s = rem / grid_sz, rem2 = rem - s * grid_sz;
      const int n = rem2 / d.gl, l = rem2 - n * d.gl;
      a_row[r] = (((long)b * d.gn + n) * d.seg + s) * d.gl + l;
    } else if (d.amap == ACX_AMAP_CONV3X3) {
      const int tile = m / grid_sz, rem = m - tile * grid_sz;
      a_n[r] = rem / d.gl;
      a_l[r] = rem - a_n[r] * d.gl;
      a_row[r] = (long)tile * grid_sz;
    }
    const int n = n0 + rbase + 32 * r;
    w_ok[r] = n < d.N;
    w_ptr[r] = (const char*)d.W + (size_t)(w_ok[r] ? n : 0) * d.ldw * (PREC == 0 ? 4 : 2);
  }

  // ---- staging registers
  float4 ra[4][PREC == 1 && !A_BF16 ? 2 : 1];
  uint4 rw[4];

  auto load_tiles = [&](int k0) {
    // A operand
    int kcol = k0 + chunk * CE;     // first K element of this thread's chunk
    int dn = 0, dl = 0, kc = kcol;
    if (d.amap == ACX_AMAP_CONV3X3) {
      const int tap = k0 / d.cin;   // uniform over the block (cin % KE == 0)
      dn = tap / 3 - 1;
      dl = tap - (tap / 3) * 3 - 1;
      kc = kcol - tap * d.cin;
    }
    const bool kin = kcol < d.K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bool ok = a_ok[r] && kin;
      long srow = a_row[r];
      if (d.amap == ACX_AMAP_CONV3X3) {
        const int nn = a_n[r] + dn, ll = a_l[r] + dl;
        ok = ok && nn >= 0 && nn < d.gn && ll >= 0 && ll < d.gl;
        srow += (long)nn * d.gl + ll;
      }
      if constexpr (A_BF16) {
        const u16* p = (const u16*)d.A + (size_t)srow * d.lda + kc;
        uint4 v = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
        ra[r][0] = *reinterpret_cast<float4*>(&v);
      } else {
        const float* p = (const float*)d.A + (size_t)srow * d.lda + kc;
        constexpr int NL = PREC == 1 ? 2 : 1;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
          float4 v = ok ? ld4(p + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (d.a_sub && ok) {
            const float4 s = ld4(d.a_sub + kc + 4 * u);
            v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w;
          }
          ra[r][u] = v;
        }
      }
    }
    // W operand
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = w_ok[r] && kin;
      const char* p = w_ptr[r] + (size_t)kcol * (PREC == 0 ? 4 : 2);
      rw[r] = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    }
  };

  auto store_tiles = [&](int stage) {
    char* sA = smem + stage * 2 * TILE_B;
    char* sW = sA + TILE_B;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = (rbase + 32 * r) * ROWB + chunk * 16;
      if constexpr (PREC == 1 && !A_BF16) {
        *reinterpret_cast<uint4*>(sA + off) = pack_bf16x8(ra[r][0], ra[r][1]);
      } else {
        *reinterpret_cast<float4*>(sA + off) = ra[r][0];
      }
      *reinterpret_cast<uint4*>(sW + off) = rw[r];
    }
  };

  // ---- accumulators
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
  const int a_off = (wm * 64 + li) * ROWB + hh * 16;
  const int w_off = (wn * 64 + li) * ROWB + hh * 16;

  const int nk = (d.K + KE - 1) / KE;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * KE);
    const char* sA = smem + cur * 2 * TILE_B;
    const char* sW = sA + TILE_B;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (PREC == 0) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const float4*>(sA + a_off + i * 32 * ROWB + q * 32);
          b[i] = *reinterpret_cast<const float4*>(sW + w_off + i * 32 * ROWB + q * 32);
        }
        const float av[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
        const float bv[2][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][j], bv[ni][j], acc[mi][ni], 0, 0, 0);
      } else {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(sA + a_off + i * 32 * ROWB + q * 32);
          b[i] = *reinterpret_cast<const bf16x8*>(sW + w_off + i * 32 * ROWB + q * 32);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn * 64 + ni * 32 + li;
    if (col >= d.N) continue;
    const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row >= d.M) continue;
        float v = acc[mi][ni][r] + bias;
        if (d.act == ACX_ACT_QUICKGELU) {
          v = v * (1.f / (1.f + __expf(-1.702f * v)));
        } else if (d.act == ACX_ACT_LEAKYRELU) {
          v = v > 0.f ? v : 0.01f * v;
        }
        if (d.pos0) {
          const int l = row % d.gl, n = (row / d.gl) % d.gn;
          v += d.pos0[(size_t)n * d.N + col];
          v += d.pos1[(size_t)l * d.N + col];
        }
        if (d.residual) v += d.residual[(size_t)row * d.ldr + col];
        if constexpr (C_BF16) {
          ((u16*)d.C)[(size_t)row * d.ldc + col] = f2bf(v);
        } else {
          ((float*)d.C)[(size_t)row * d.ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int acx_gemm(acx_ctx* ctx, const acx_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->C) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: null pointer%s");
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: empty shape%s");
  const int prec = d->prec;
  if (prec != ACX_PREC_F32 && prec != ACX_PREC_BF16)
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown precision%s");
  const int a_bf16 = d->a_dtype == ACX_BF16, c_bf16 = d->c_dtype == ACX_BF16;
  if (prec == ACX_PREC_F32 && (a_bf16))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: PREC_F32 needs f32 A%s");
  const int kal = prec == ACX_PREC_F32 ? 4 : 8;
  if (d->K % kal || d->lda % (a_bf16 ? 8 : 4) || d->ldw % kal)
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: K/lda/ldw must be multiples of %s%ld elements", "", kal);
  if (((uintptr_t)d->A | (uintptr_t)d->W) & 15)
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: A/W must be 16-byte aligned%s");
  if (d->a_sub && (a_bf16 || ((uintptr_t)d->a_sub & 15)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: a_sub needs f32 A and 16-byte alignment%s");
  if (d->amap == ACX_AMAP_CONV3X3) {
    const int ke = prec == ACX_PREC_F32 ? 32 : 64;
    if (d->cin <= 0 || d->K != 9 * d->cin || d->cin % ke || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad conv3x3 geometry%s");
    if (d->a_sub) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: a_sub with conv3x3%s");
  } else if (d->amap == ACX_AMAP_TESTTILE) {
    if (d->seg <= 0 || d->gn <= 0 || d->gl <= 0 || d->M % (d->gn * d->gl * d->seg))
      return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: bad test-tile geometry%s");
  } else if (d->amap != ACX_AMAP_IDENTITY) {
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_gemm: unknown amap%s");
  }
  if ((d->pos0 != nullptr) != (d->pos1 != nullptr) || (d->pos0 && (d->gn <= 0 || d->gl <= 0)))
    return acx_fail(ctx, ACX_E_BADARG, "acx_gemm: pos0/pos1 need both pointers and a grid%s");

  Args g;
  g.d = *d;
  const int tiles_m = (d->M + BM - 1) / BM;
  g.tiles_n = (d->N + BN - 1) / BN;
  const dim3 grid((unsigned)(tiles_m * g.tiles_n)), block(NTHREADS);
  const size_t lds = 4 * TILE_B;
  hipStream_t s = (hipStream_t)stream;
#define ACX_LAUNCH(P, AB, CB)                                                                       \
  do {                                                                                              \
    static bool attr_done = false;                                                                  \
    if (!attr_done) {                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_kernel<P, AB, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_done = true;                                                                             \
    }                                                                                               \
    hipLaunchKernelGGL((gemm_kernel<P, AB, CB>), grid, block, lds, s, g);                           \
  } while (0)
  if (prec == ACX_PREC_F32) {
    if (c_bf16) ACX_LAUNCH(0, 0, 1); else ACX_LAUNCH(0, 0, 0);
  } else if (a_bf16) {
    if (c_bf16) ACX_LAUNCH(1, 1, 1); else ACX_LAUNCH(1, 1, 0);
  } else {
    if (c_bf16) ACX_LAUNCH(1, 0, 1); else ACX_LAUNCH(1, 0, 0);
  }
#undef ACX_LAUNCH
  ACX_CHECK_LAUNCH(ctx, "acx_gemm");
  return ACX_OK;
}
