// Training-side kernels of the AnomalyCLIP head (backward passes, MIL selection, loss, AdamW).
// All reductions over rows are two-stage and fixed-order (per-block partials -> acx_reduce_rows), so a
// training step is bit-reproducible run to run; no float atomics on the gradient path.
#include "acx_internal.h"

namespace {

// ------------------------------------------------------------------ generic helpers
// out[c] = sum_p part[p][c]   (fixed order; part is [nparts][width])
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int nparts, int width) {
  // 16 columns x 16 slices of the parts per block: slice s sums parts s, s+16, ... in order, then the 16 slice
  // sums are added in order -- a fixed summation tree, with width/16 blocks instead of width/256
  __shared__ float red[16][17];
  const int ci = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ci;
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
    for (int k = 0; k < 16; ++k) t += red[k][ci];
    out[c] = t;
  }
}

// block-level sum of a per-thread value (256 threads), result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ LayerNorm / ChanLayerNorm backward
// one wave per row, 16 rows per wave, 64 rows per block; per-block partial dw/db -> part[blk][2*D]
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ part, int64_t rows, float eps, int mode,
                                                            float dx_scale, const float* __restrict__ add, int rpw) {
  // rpw rows per wave: 16 when the per-block parameter-gradient partials are wanted (64 rows per block), 1 otherwise (the
  // frozen text layers: dX only -- 154 rows then fill 39 workgroups instead of walking 16 rows per wave on 3);
  // add (may be NULL): dx = add + dx_scale * dL/dx, the residual branch of a pre-norm block folded into the same pass
  constexpr int D = 64 * VPL;
  constexpr float invD = 1.f / D;
  __shared__ float sred[4][2 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ww[VPL], dwa[VPL], dba[VPL];
  load_row<VPL>(w, lane, ww);
#pragma unroll
  for (int i = 0; i < VPL; ++i) dwa[i] = dba[i] = 0.f;
  const int64_t rbase = ((int64_t)blockIdx.x * 4 + wave) * rpw;
  for (int rr = 0; rr < rpw; ++rr) {
    const int64_t row = rbase + rr;
    if (row >= rows) break;
    float v[VPL], g[VPL];
    load_row<VPL>(x + row * D, lane, v);
    load_row<VPL>(dy + row * D, lane, g);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += v[i];
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float var = wave_sum(q) * invD;
    float sc, c2;   // y = xc * sc * w + b
    if (mode == ACX_NORM_LAYER) { sc = 1.f / sqrtf(var + eps); } else { sc = 1.f / (sqrtf(var) + eps); }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      dba[i] += g[i];
      dwa[i] += g[i] * v[i] * sc;
      g[i] *= ww[i];
      sg += g[i];
      sgx += g[i] * v[i];
    }
    sg = wave_sum(sg) * invD;
    sgx = wave_sum(sgx);
    if (mode == ACX_NORM_LAYER) c2 = sc * sc * sc * sgx * invD;          // d(rstd)/dx term
    else { const float sd = sqrtf(var); c2 = sd > 0.f ? sc * sc * sgx * invD / sd : 0.f; }
    if (dx) {
      float o[VPL];
#pragma unroll
      for (int i = 0; i < VPL; ++i) o[i] = (sc * (g[i] - sg) - c2 * v[i]) * dx_scale;
      if (add) {
        float ad[VPL];
        load_row<VPL>(add + row * D, lane, ad);
#pragma unroll
        for (int i = 0; i < VPL; ++i) o[i] += ad[i];
      }
      if constexpr (VPL % 4 == 0) {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i)
          *reinterpret_cast<float4*>(dx + row * D + 4 * lane + 256 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
      } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) dx[row * D + lane + 64 * i] = o[i];
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int e = elem_index<VPL>(lane, i);
    sred[wave][e] = dwa[i];
    sred[wave][D + e] = dba[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += 256)
    part[(size_t)blockIdx.x * 2 * D + e] = sred[0][e] + sred[1][e] + sred[2][e] + sred[3][e];
}

// ------------------------------------------------------------------ classifier head backward
// forward: a=(x1+x2)/2; z=LN(a)*lw+lb; s=sigmoid(z.w+b).  part[blk] = [dlw(E) | dlb(E) | dw(E) | db(1) pad]
template <int VPL>
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                           const float* __restrict__ lw, const float* __restrict__ lb,
                                                           const float* __restrict__ w, const float* __restrict__ scores,
                                                           const float* __restrict__ dscores, float* __restrict__ dx,
                                                           float* __restrict__ part, int64_t rows, int rpw) {
  constexpr int E = 64 * VPL;
  constexpr float invD = 1.f / E;
  constexpr int PW = 3 * E + 4;
  __shared__ float sred[4][PW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lww[VPL], lbb[VPL], wv[VPL], a_lw[VPL], a_lb[VPL], a_w[VPL];
  load_row<VPL>(lw, lane, lww);
  load_row<VPL>(lb, lane, lbb);
  load_row<VPL>(w, lane, wv);
  float a_b = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) a_lw[i] = a_lb[i] = a_w[i] = 0.f;
  const int64_t rbase = ((int64_t)blockIdx.x * 4 + wave) * rpw;
  for (int rr = 0; rr < rpw; ++rr) {
    const int64_t row = rbase + rr;
    if (row >= rows) break;
    float v[VPL], c[VPL];
    load_row<VPL>(x1 + row * E, lane, v);
    load_row<VPL>(x2 + row * E, lane, c);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] = (v[i] + c[i]) * 0.5f; s += v[i]; }
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rstd = 1.f / sqrtf(wave_sum(q) * invD + 1e-5f);
    const float sg = scores[row];
    const float ddot = dscores[row] * sg * (1.f - sg);
    a_b += ddot;
    float g[VPL], sgm = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float xh = v[i] * rstd;
      const float z = xh * lww[i] + lbb[i];
      a_w[i] += ddot * z;
      const float dz = ddot * wv[i];
      a_lb[i] += dz;
      a_lw[i] += dz * xh;
      g[i] = dz * lww[i];
      sgm += g[i];
      sgx += g[i] * v[i];
    }
    sgm = wave_sum(sgm) * invD;
    sgx = wave_sum(sgx) * invD * rstd * rstd * rstd;
    float o[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) o[i] = 0.5f * (rstd * (g[i] - sgm) - sgx * v[i]);   // d(x1) == d(x2)
    if constexpr (VPL % 4 == 0) {
#pragma unroll
      for (int i = 0; i < VPL / 4; ++i)
        *reinterpret_cast<float4*>(dx + row * E + 4 * lane + 256 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < VPL; ++i) dx[row * E + lane + 64 * i] = o[i];
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int e = elem_index<VPL>(lane, i);
    sred[wave][e] = a_lw[i];
    sred[wave][E + e] = a_lb[i];
    sred[wave][2 * E + e] = a_w[i];
  }
  if (lane == 0) sred[wave][3 * E] = a_b;    // a_b is identical in all lanes of the wave
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * E + 1; e += 256)
    part[(size_t)blockIdx.x * PW + e] = sred[0][e] + sred[1][e] + sred[2][e] + sred[3][e];
}

// ------------------------------------------------------------------ elementwise
// mode 0: LeakyReLU backward from the saved OUTPUT u (sign(u) == sign(pre-activation)): d *= u>0 ? 1 : 0.01
// mode 1: QuickGELU backward from the saved pre-activation: d *= sig*(1 + 1.702*x*(1-sig))
// mode 2: QuickGELU forward: out = x*sigmoid(1.702x)
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ saved, const float* __restrict__ d,
                                                  float* __restrict__ out, int64_t n4, int mode) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 s = reinterpret_cast<const float4*>(saved)[i];
  float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
  if (d) g = reinterpret_cast<const float4*>(d)[i];
  float sv[4] = {s.x, s.y, s.z, s.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (mode == 0) gv[k] *= sv[k] > 0.f ? 1.f : 0.01f;
    else {
      const float sg = 1.f / (1.f + __expf(-1.702f * sv[k]));
      if (mode == 1) gv[k] *= sg * (1.f + 1.702f * sv[k] * (1.f - sg));
      else gv[k] = sv[k] * sg;
    }
  }
  reinterpret_cast<float4*>(out)[i] = make_float4(gv[0], gv[1], gv[2], gv[3]);
}

// LeakyReLU backward with the saved activation as the HI bf16 plane of its three-plane form (sign(hi) = sign(u)); the result
// goes out as f32 (bias-gradient column sums) AND as three bf16 planes (the operand of the next bf16 x 6 products)
__global__ __launch_bounds__(256) void leaky_grad_planes_kernel(const u16* __restrict__ u_hi, const float* __restrict__ d,
                                                                float* __restrict__ out, u16* __restrict__ planes, int64_t plane,
                                                                int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const uint2 uh = reinterpret_cast<const uint2*>(u_hi)[i];
  const float4 g = reinterpret_cast<const float4*>(d)[i];
  const float sv[4] = {__uint_as_float(uh.x << 16), __uint_as_float(uh.x & 0xffff0000u), __uint_as_float(uh.y << 16),
                       __uint_as_float(uh.y & 0xffff0000u)};
  float gv[4] = {g.x, g.y, g.z, g.w};
  u16 h[4], m[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gv[k] *= sv[k] > 0.f ? 1.f : 0.01f;
    h[k] = f2bf(gv[k]);
    const float r1 = gv[k] - bf2f(h[k]);
    m[k] = f2bf(r1);
    l[k] = f2bf(r1 - bf2f(m[k]));
  }
  reinterpret_cast<float4*>(out)[i] = make_float4(gv[0], gv[1], gv[2], gv[3]);
  uint2 pk;
  pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
  reinterpret_cast<uint2*>(planes)[i] = pk;
  pk.x = (uint32_t)m[0] | ((uint32_t)m[1] << 16); pk.y = (uint32_t)m[2] | ((uint32_t)m[3] << 16);
  reinterpret_cast<uint2*>(planes + plane)[i] = pk;
  pk.x = (uint32_t)l[0] | ((uint32_t)l[1] << 16); pk.y = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
  reinterpret_cast<uint2*>(planes + 2 * plane)[i] = pk;
}

// out = a + b
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
  reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// out[c, r] = in[r, c]   (weight transposes for the dX GEMMs), 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cn) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (by + j < R && bx + tx < Cn) tile[j][tx] = in[(size_t)(by + j) * Cn + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < Cn && by + tx < R) out[(size_t)(bx + j) * R + by + tx] = tile[tx][j];
}

// conv weight for the dX implicit GEMM: out[ci][tap'][co] = w[co][ci][8 - tap']   (w is [Cout][Cin][3][3])
__global__ __launch_bounds__(256) void conv_w_dx_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cin * 9;
  if (i >= total) return;
  const int co = (int)(i % Cout);
  const int tp = (int)((i / Cout) % 9);
  const int ci = (int)(i / ((int64_t)Cout * 9));
  out[i] = w[((size_t)co * Cin + ci) * 9 + (8 - tp)];
}

// ------------------------------------------------------------------ sequence attention backward
// qkv rows = [q | k | v] (each heads*E wide); a "line" is a sequence of T tokens whose rows are
// row_of(line, j).  One thread per (line, head, query/key index); two passes over LDS-staged operands.
template <int E>
__global__ __launch_bounds__(256) void seq_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                           float* __restrict__ dqkv, int tiles, int gn, int gl, int heads,
                                                           int axis, int causal, float scale, int T, int gpb,
                                                           int64_t ngroups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sA = reinterpret_cast<float*>(smem);            // [gpb][T][E]   K then Q
  float* sB = sA + gpb * T * E;                           // [gpb][T][E]   V then dO
  float* sM = sB + gpb * T * E;                           // [gpb][T] max
  float* sL = sM + gpb * T;                               // [gpb][T] 1/sum
  float* sD = sL + gpb * T;                               // [gpb][T] D_i
  const int He = heads * E, ld = 3 * He;
  const int t = threadIdx.x;
  const int gi = t / T, i = t - gi * T;
  const int64_t grp = (int64_t)blockIdx.x * gpb + gi;     // (line, head) group
  const bool active = gi < gpb && grp < ngroups;
  const int other = axis == 0 ? gl : gn;
  auto row_of = [&](int64_t line, int j) -> int64_t {
    const int64_t tile = line / other;
    const int o = (int)(line - tile * other);
    return axis == 0 ? (tile * gn + j) * gl + o : (tile * gn + o) * gl + j;
  };
  // cooperative staging helper: which = 0 (K,V) or 1 (Q,dO)
  auto stage = [&](int which) {
    const int f4 = E / 4;
    for (int idx = t; idx < gpb * T * f4; idx += 256) {
      const int c4 = idx % f4, tk = idx / f4;
      const int g2 = tk / T, j = tk - g2 * T;
      const int64_t gg = (int64_t)blockIdx.x * gpb + g2;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (gg < ngroups) {
        const int64_t line = gg / heads;
        const int h = (int)(gg - line * heads);
        const int64_t r = row_of(line, j);
        if (which == 0) {
          a = *reinterpret_cast<const float4*>(qkv + r * ld + He + h * E + 4 * c4);
          b = *reinterpret_cast<const float4*>(qkv + r * ld + 2 * He + h * E + 4 * c4);
        } else {
          a = *reinterpret_cast<const float4*>(qkv + r * ld + h * E + 4 * c4);
          b = *reinterpret_cast<const float4*>(dout + r * He + h * E + 4 * c4);
        }
      }
      reinterpret_cast<float4*>(sA)[idx] = a;
      reinterpret_cast<float4*>(sB)[idx] = b;
    }
  };
  stage(0);
  __syncthreads();
  int64_t line = 0, myrow = 0;
  int h = 0;
  float q[E], dO[E];
  if (active) {
    line = grp / heads;
    h = (int)(grp - line * heads);
    myrow = row_of(line, i);
#pragma unroll
    for (int c = 0; c < E / 4; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(qkv + myrow * ld + h * E + 4 * c);
      const float4 b = *reinterpret_cast<const float4*>(dout + myrow * He + h * E + 4 * c);
      q[4 * c] = a.x; q[4 * c + 1] = a.y; q[4 * c + 2] = a.z; q[4 * c + 3] = a.w;
      dO[4 * c] = b.x; dO[4 * c + 1] = b.y; dO[4 * c + 2] = b.z; dO[4 * c + 3] = b.w;
    }
    const float* kb = sA + gi * T * E;
    const float* vb = sB + gi * T * E;
    const int jmax = causal ? i + 1 : T;
    // sweep 1: max and sum
    float mx = -INFINITY;
    for (int j = 0; j < jmax; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) acc += q[e] * kb[j * E + e];
      mx = fmaxf(mx, acc * scale);
    }
    float sum = 0.f, Di = 0.f;
    for (int j = 0; j < jmax; ++j) {
      float acc = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { acc += q[e] * kb[j * E + e]; dp += dO[e] * vb[j * E + e]; }
      const float p = __expf(acc * scale - mx);
      sum += p;
      Di += p * dp;
    }
    const float inv = 1.f / sum;
    Di *= inv;
    // sweep 3: dq
    float dq[E];
#pragma unroll
    for (int e = 0; e < E; ++e) dq[e] = 0.f;
    for (int j = 0; j < jmax; ++j) {
      float acc = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { acc += q[e] * kb[j * E + e]; dp += dO[e] * vb[j * E + e]; }
      const float ds = __expf(acc * scale - mx) * inv * (dp - Di) * scale;
#pragma unroll
      for (int e = 0; e < E; ++e) dq[e] += ds * kb[j * E + e];
    }
#pragma unroll
    for (int c = 0; c < E / 4; ++c)
      *reinterpret_cast<float4*>(dqkv + myrow * ld + h * E + 4 * c) = make_float4(dq[4 * c], dq[4 * c + 1], dq[4 * c + 2], dq[4 * c + 3]);
    sM[gi * T + i] = mx;
    sL[gi * T + i] = inv;
    sD[gi * T + i] = Di;
  }
  __syncthreads();
  stage(1);       // Q and dO of every token
  __syncthreads();
  if (active) {
    // this thread is now KEY/VALUE index j = i
    float k[E], v[E], dk[E], dv[E];
#pragma unroll
    for (int c = 0; c < E / 4; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(qkv + myrow * ld + He + h * E + 4 * c);
      const float4 b = *reinterpret_cast<const float4*>(qkv + myrow * ld + 2 * He + h * E + 4 * c);
      k[4 * c] = a.x; k[4 * c + 1] = a.y; k[4 * c + 2] = a.z; k[4 * c + 3] = a.w;
      v[4 * c] = b.x; v[4 * c + 1] = b.y; v[4 * c + 2] = b.z; v[4 * c + 3] = b.w;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) dk[e] = dv[e] = 0.f;
    const float* qb = sA + gi * T * E;
    const float* ob = sB + gi * T * E;
    for (int qi = causal ? i : 0; qi < T; ++qi) {
      float acc = 0.f, dp = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { acc += qb[qi * E + e] * k[e]; dp += ob[qi * E + e] * v[e]; }
      const float p = __expf(acc * scale - sM[gi * T + qi]) * sL[gi * T + qi];
      const float ds = p * (dp - sD[gi * T + qi]) * scale;
#pragma unroll
      for (int e = 0; e < E; ++e) { dk[e] += ds * qb[qi * E + e]; dv[e] += p * ob[qi * E + e]; }
    }
#pragma unroll
    for (int c = 0; c < E / 4; ++c) {
      *reinterpret_cast<float4*>(dqkv + myrow * ld + He + h * E + 4 * c) = make_float4(dk[4 * c], dk[4 * c + 1], dk[4 * c + 2], dk[4 * c + 3]);
      *reinterpret_cast<float4*>(dqkv + myrow * ld + 2 * He + h * E + 4 * c) = make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
    }
  }
}


// ------------------------------------------------------------------ sequence attention backward, wave-per-row form
// For the CLIP text tower (few sequences: classes x heads, T = 77, E = 64) the thread-per-row kernel above leaves
// the chip empty (9 k threads with 10^4-long serial loops: 0.73 ms per call).  Here ONE WAVEFRONT owns one query
// row (pass 1) / one key row (pass 2): lanes hold up to KPL keys (queries) each for the dot products, then switch
// to owning the E output dims for the weighted sums, with the softmax statistics exchanged through a small
// global scratch ([rows*heads][3] = max, 1/sum, D).  Reads come straight from L2 (K/V/Q/dO of a head are 20 KB).
template <int E, int KPL>
__global__ __launch_bounds__(256) void seq_attn_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, float* __restrict__ stats,
                                                                int T, int heads, int causal, float scale, int pass,
                                                                int64_t nrows) {
  // row index r = ((seq * heads) + h) * T + i
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int i = (int)(r % T);
  const int64_t sh = r / T;
  const int h = (int)(sh % heads);
  const int64_t seq = sh / heads;
  const int He = heads * E, ld = 3 * He;
  const float* base = qkv + seq * T * ld + h * E;          // token 0 of this sequence/head (q section)
  const float* dob = dout + seq * T * He + h * E;
  float* dqb = dqkv + seq * T * ld + h * E;
  float* st = stats + sh * T * 3;                          // [T][3] of this (sequence, head)
  constexpr int EPL = E / 64 > 0 ? E / 64 : 1;             // output dims per lane (E = 64 -> 1)
  static_assert(E == 64, "rows kernel is built for head dim 64");

  if (pass == 0) {
    // ---- query row i: s_j, dP_j for this lane's keys
    float sv[KPL], dp[KPL];
    float mx = -INFINITY;
    const int jmax = causal ? i + 1 : T;
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int j = lane + 64 * u;
      sv[u] = -INFINITY;
      dp[u] = 0.f;
      if (j < jmax) {
        float a = 0.f, b = 0.f;
        const float* kp = base + (int64_t)j * ld + He;
        const float* vp = base + (int64_t)j * ld + 2 * He;
        const float* qp = base + (int64_t)i * ld;
        const float* op = dob + (int64_t)i * He;
#pragma unroll
        for (int c = 0; c < E / 4; ++c) {
          const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * c), q4 = *reinterpret_cast<const float4*>(qp + 4 * c);
          const float4 v4 = *reinterpret_cast<const float4*>(vp + 4 * c), o4 = *reinterpret_cast<const float4*>(op + 4 * c);
          a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
          b += o4.x * v4.x + o4.y * v4.y + o4.z * v4.z + o4.w * v4.w;
        }
        sv[u] = a * scale;
        dp[u] = b;
      }
      mx = fmaxf(mx, sv[u]);
    }
    mx = wave_max(mx);
    float sum = 0.f, Di = 0.f;
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const float p = __expf(sv[u] - mx);     // exp(-inf) = 0 for masked / absent keys
      sv[u] = p;
      sum += p;
      Di += p * dp[u];
    }
    sum = wave_sum(sum);
    Di = wave_sum(Di);
    const float inv = 1.f / sum;
    Di *= inv;
    // dS_j = p_j (dP_j - D) * scale ; dq[e] = sum_j dS_j k[j][e]  (lane = e)
#pragma unroll
    for (int u = 0; u < KPL; ++u) sv[u] = sv[u] * inv * (dp[u] - Di) * scale;
    float dq = 0.f;
    for (int j = 0; j < jmax; ++j) {
      const int u_ = j >> 6;
      float mine = sv[0];
#pragma unroll
      for (int u = 1; u < KPL; ++u) mine = u_ == u ? sv[u] : mine;
      const float ds = __shfl(mine, j & 63, 64);
      dq += ds * base[(int64_t)j * ld + He + lane];
    }
    dqb[(int64_t)i * ld + lane] = dq;
    if (lane == 0) { st[i * 3] = mx; st[i * 3 + 1] = inv; st[i * 3 + 2] = Di; }
  } else {
    // ---- key/value row j = i: lanes hold queries qi = lane + 64u
    const int j = i;
    float pv[KPL], dsv[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int qi = lane + 64 * u;
      pv[u] = 0.f;
      dsv[u] = 0.f;
      if (qi < T && (!causal || qi >= j)) {
        float a = 0.f, b = 0.f;
        const float* kp = base + (int64_t)j * ld + He;
        const float* vp = base + (int64_t)j * ld + 2 * He;
        const float* qp = base + (int64_t)qi * ld;
        const float* op = dob + (int64_t)qi * He;
#pragma unroll
        for (int c = 0; c < E / 4; ++c) {
          const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * c), q4 = *reinterpret_cast<const float4*>(qp + 4 * c);
          const float4 v4 = *reinterpret_cast<const float4*>(vp + 4 * c), o4 = *reinterpret_cast<const float4*>(op + 4 * c);
          a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
          b += o4.x * v4.x + o4.y * v4.y + o4.z * v4.z + o4.w * v4.w;
        }
        const float p = __expf(a * scale - st[qi * 3]) * st[qi * 3 + 1];
        pv[u] = p;
        dsv[u] = p * (b - st[qi * 3 + 2]) * scale;
      }
    }
    float dk = 0.f, dv = 0.f;
    for (int qi = causal ? j : 0; qi < T; ++qi) {
      const int u_ = qi >> 6;
      float mp = pv[0], md = dsv[0];
#pragma unroll
      for (int u = 1; u < KPL; ++u) { mp = u_ == u ? pv[u] : mp; md = u_ == u ? dsv[u] : md; }
      const float p = __shfl(mp, qi & 63, 64), ds = __shfl(md, qi & 63, 64);
      dk += ds * base[(int64_t)qi * ld + lane];
      dv += p * dob[(int64_t)qi * He + lane];
    }
    dqb[(int64_t)j * ld + He + lane] = dk;
    dqb[(int64_t)j * ld + 2 * He + lane] = dv;
  }
  (void)EPL;
}

// One WORKGROUP per (sequence, head) for the text tower (T <= 128, E = 64): q, k, v, dO of the head are staged
// ONCE in LDS (4 x T x 272 B) instead of being re-read from L2 by every row (77 x 20 KB per row-wave: the rows
// kernel above moves 0.67 GB per launch), both passes run in the same launch with the softmax statistics in LDS,
// and a wave walks the rows r = wave, wave + NWV, ...  Same arithmetic and summation order per row as the rows
// kernel.
constexpr int SB_ROWF = 68;            // floats per staged row (64 + 4: conflict-free ds_read_b128 across rows)
template <int KPL>
__global__ __launch_bounds__(512) void seq_attn_bwd_block_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, int T, int heads, int causal,
                                                                 float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_sb[];
  float* sQ = reinterpret_cast<float*>(smem_sb);
  float* sK = sQ + T * SB_ROWF;
  float* sV = sK + T * SB_ROWF;
  float* sO = sV + T * SB_ROWF;
  float* sS = sO + T * SB_ROWF;        // [T][3] max, 1/sum, D
  const int TP = (T + 3) & ~3;
  float* sW = sS + 3 * T + ((4 - (3 * T) % 4) % 4);      // per-wave rows [nwv][2][TP] (16-byte aligned): dS / P of the current row
  const int h = blockIdx.x % heads;
  const int64_t seq = blockIdx.x / heads;
  const int He = heads * 64, ld = 3 * He;
  const float* base = qkv + seq * T * ld + h * 64;
  const float* dob = dout + seq * T * He + h * 64;
  float* dqb = dqkv + seq * T * ld + h * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwv = blockDim.x >> 6;
  float* wA = sW + (size_t)wave * 2 * TP;                 // dS (pass 0) / dS^T (pass 1) of this wave's current row
  float* wB = wA + TP;                                    // P^T (pass 1)
  for (int i = t; i < T * 16; i += blockDim.x) {
    const int row = i >> 4, c4 = i & 15;
    const float* p = base + (int64_t)row * ld + 4 * c4;
    *reinterpret_cast<float4*>(sQ + row * SB_ROWF + 4 * c4) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(sK + row * SB_ROWF + 4 * c4) = *reinterpret_cast<const float4*>(p + He);
    *reinterpret_cast<float4*>(sV + row * SB_ROWF + 4 * c4) = *reinterpret_cast<const float4*>(p + 2 * He);
    *reinterpret_cast<float4*>(sO + row * SB_ROWF + 4 * c4) = *reinterpret_cast<const float4*>(dob + (int64_t)row * He + 4 * c4);
  }
  __syncthreads();
  // ---- pass 0: query rows
  for (int i = wave; i < T; i += nwv) {
    float sv[KPL], dp[KPL];
    float mx = -INFINITY;
    const int jmax = causal ? i + 1 : T;
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int j = lane + 64 * u;
      sv[u] = -INFINITY;
      dp[u] = 0.f;
      if (j < jmax) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float4 k4 = *reinterpret_cast<const float4*>(sK + j * SB_ROWF + 4 * c), q4 = *reinterpret_cast<const float4*>(sQ + i * SB_ROWF + 4 * c);
          const float4 v4 = *reinterpret_cast<const float4*>(sV + j * SB_ROWF + 4 * c), o4 = *reinterpret_cast<const float4*>(sO + i * SB_ROWF + 4 * c);
          a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
          b += o4.x * v4.x + o4.y * v4.y + o4.z * v4.z + o4.w * v4.w;
        }
        sv[u] = a * scale;
        dp[u] = b;
      }
      mx = fmaxf(mx, sv[u]);
    }
    mx = wave_max(mx);
    float sum = 0.f, Di = 0.f;
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const float p = __expf(sv[u] - mx);
      sv[u] = p;
      sum += p;
      Di += p * dp[u];
    }
    sum = wave_sum(sum);
    Di = wave_sum(Di);
    const float inv = 1.f / sum;
    Di *= inv;
#pragma unroll
    for (int u = 0; u < KPL; ++u) sv[u] = sv[u] * inv * (dp[u] - Di) * scale;
    // dq[e] = sum_j dS_j k[j][e] (lane = e): the row of dS goes through this wave's LDS scratch, so the loop is a
    // broadcast float4 read + 4 K reads per 4 keys instead of a cross-lane shuffle per key
#pragma unroll
    for (int u = 0; u < KPL; ++u) { const int j = lane + 64 * u; if (j < TP) wA[j] = j < jmax ? sv[u] : 0.f; }
    float dq = 0.f;
    for (int j = 0; j < jmax; j += 4) {
      const float4 d4 = *reinterpret_cast<const float4*>(wA + j);
      dq += d4.x * sK[j * SB_ROWF + lane];
      if (j + 1 < jmax) dq += d4.y * sK[(j + 1) * SB_ROWF + lane];
      if (j + 2 < jmax) dq += d4.z * sK[(j + 2) * SB_ROWF + lane];
      if (j + 3 < jmax) dq += d4.w * sK[(j + 3) * SB_ROWF + lane];
    }
    dqb[(int64_t)i * ld + lane] = dq;
    if (lane == 0) { sS[i * 3] = mx; sS[i * 3 + 1] = inv; sS[i * 3 + 2] = Di; }
  }
  __syncthreads();
  // ---- pass 1: key / value rows
  for (int j = wave; j < T; j += nwv) {
    float pv[KPL], dsv[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int qi = lane + 64 * u;
      pv[u] = 0.f;
      dsv[u] = 0.f;
      if (qi < T && (!causal || qi >= j)) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float4 k4 = *reinterpret_cast<const float4*>(sK + j * SB_ROWF + 4 * c), q4 = *reinterpret_cast<const float4*>(sQ + qi * SB_ROWF + 4 * c);
          const float4 v4 = *reinterpret_cast<const float4*>(sV + j * SB_ROWF + 4 * c), o4 = *reinterpret_cast<const float4*>(sO + qi * SB_ROWF + 4 * c);
          a += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
          b += o4.x * v4.x + o4.y * v4.y + o4.z * v4.z + o4.w * v4.w;
        }
        const float p = __expf(a * scale - sS[qi * 3]) * sS[qi * 3 + 1];
        pv[u] = p;
        dsv[u] = p * (b - sS[qi * 3 + 2]) * scale;
      }
    }
#pragma unroll
    for (int u = 0; u < KPL; ++u) { const int qi = lane + 64 * u; if (qi < TP) { wA[qi] = dsv[u]; wB[qi] = pv[u]; } }
    float dk = 0.f, dv = 0.f;
    const int q0 = causal ? (j & ~3) : 0;                 // entries below j are zero (masked above)
    for (int qi = q0; qi < T; qi += 4) {
      const float4 d4 = *reinterpret_cast<const float4*>(wA + qi), p4 = *reinterpret_cast<const float4*>(wB + qi);
      dk += d4.x * sQ[qi * SB_ROWF + lane];
      dv += p4.x * sO[qi * SB_ROWF + lane];
      if (qi + 1 < T) { dk += d4.y * sQ[(qi + 1) * SB_ROWF + lane]; dv += p4.y * sO[(qi + 1) * SB_ROWF + lane]; }
      if (qi + 2 < T) { dk += d4.z * sQ[(qi + 2) * SB_ROWF + lane]; dv += p4.z * sO[(qi + 2) * SB_ROWF + lane]; }
      if (qi + 3 < T) { dk += d4.w * sQ[(qi + 3) * SB_ROWF + lane]; dv += p4.w * sO[(qi + 3) * SB_ROWF + lane]; }
    }
    dqb[(int64_t)j * ld + He + lane] = dk;
    dqb[(int64_t)j * ld + 2 * He + lane] = dv;
  }
}

// Axial attention backward on the f32 MFMA (temporal model: sequences of T = 32 segments or T = 16 frames, head dim 32 or 16,
// thousands of independent (line, head) groups per launch; seq_attn_bwd_kernel above spends 145 us per launch on the VALU).
// One WAVE per group, four groups per workgroup, grid-stride; q, k, v, dO of the group in a wave-private LDS slice (rows
// padded by 4 floats).  Two passes, so that every register tile is consumed in the layout the MFMA leaves it in:
//   pass 1  S^T = K Q^T and dP^T = V dO^T (lane = query column): softmax statistics are in-lane sums + two cross-quarter
//           shuffles; dS^T tiles are the A operand of dQ = dS K as they stand (A[i][j] = dS^T[j][i]: lane & 15 = i,
//           MFMA step r <-> key j = 4 kq + r);
//   pass 2  S = Q K^T and dP = dO V^T (lane = key column) with the row statistics read back from LDS: the P and dS tiles are
//           the A operand of dV = P^T dO and dK = dS^T Q as they stand.
// 224 MFMAs of 16x16x4 per (T = 32, E = 32) group; exact f32 products, fixed summation order.
template <int T, int E>
__global__ __launch_bounds__(256) void axial_attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                  float* __restrict__ dqkv, int gn, int gl, int heads, int axis,
                                                                  float scale, int64_t ngroups) {
  constexpr int ES = E + 4;                      // LDS row stride (floats)
  constexpr int TT = T / 16, ET = E / 16, KS = E / 16;
  constexpr int WF = 4 * T * ES + 4 * T;          // floats per wave: Q, K, V, dO, stats (m, inv, D, pad)
  extern __shared__ __attribute__((aligned(16))) char smem_ax[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  float* sQ = reinterpret_cast<float*>(smem_ax) + wave * WF;
  float* sK = sQ + T * ES;
  float* sV = sK + T * ES;
  float* sO = sV + T * ES;
  float* sS = sO + T * ES;                        // [3][T]
  const int He = heads * E, ld = 3 * He;
  const int other = axis == 0 ? gl : gn;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t line = grp / heads;
    const int h = (int)(grp - line * heads);
    const int64_t tile = line / other;
    const int o = (int)(line - tile * other);
    // row of sequence element t: axis 0 walks n (stride gl rows), axis 1 walks l (stride 1)
    const int64_t row0 = axis == 0 ? tile * gn * gl + o : (tile * gn + o) * gl;
    const int64_t rstep = axis == 0 ? gl : 1;
    // ---- stage the group (float4 per lane, E/4 lanes per row)
#pragma unroll
    for (int idx = lane; idx < T * (E / 4); idx += 64) {
      const int t = idx / (E / 4), c4 = idx - t * (E / 4);
      const int64_t r = row0 + t * rstep;
      const float* p = qkv + r * ld + h * E + 4 * c4;
      *reinterpret_cast<float4*>(sQ + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(sK + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p + He);
      *reinterpret_cast<float4*>(sV + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(p + 2 * He);
      *reinterpret_cast<float4*>(sO + t * ES + 4 * c4) = *reinterpret_cast<const float4*>(dout + r * He + h * E + 4 * c4);
    }
    // (wave-private LDS: the wave's own ds_write -> ds_read order is enough, no workgroup barrier)
    // ---- pass 1: transposed scores, lane column = query i
    f32x4 st[TT][TT], dt[TT][TT];                 // [jt][it]
#pragma unroll
    for (int jt = 0; jt < TT; ++jt)
#pragma unroll
      for (int it = 0; it < TT; ++it) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 k4 = *reinterpret_cast<const float4*>(sK + (16 * jt + li) * ES + 16 * ks + 4 * kq);
          const float4 q4 = *reinterpret_cast<const float4*>(sQ + (16 * it + li) * ES + 16 * ks + 4 * kq);
          const float4 v4 = *reinterpret_cast<const float4*>(sV + (16 * jt + li) * ES + 16 * ks + 4 * kq);
          const float4 o4 = *reinterpret_cast<const float4*>(sO + (16 * it + li) * ES + 16 * ks + 4 * kq);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.x, q4.x, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, o4.x, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.y, q4.y, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, o4.y, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.z, q4.z, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, o4.z, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.w, q4.w, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, o4.w, b, 0, 0, 0);
        }
        st[jt][it] = a; dt[jt][it] = b;
      }
#pragma unroll
    for (int it = 0; it < TT; ++it) {
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[jt][it][r] *= scale; mx = fmaxf(mx, st[jt][it][r]); }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[jt][it][r] = __expf(st[jt][it][r] - mx); sum += st[jt][it][r]; }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.f / sum;
      float Di = 0.f;
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[jt][it][r] *= inv; Di += st[jt][it][r] * dt[jt][it][r]; }
      Di += __shfl_xor(Di, 16, 64);
      Di += __shfl_xor(Di, 32, 64);
#pragma unroll
      for (int jt = 0; jt < TT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dt[jt][it][r] = st[jt][it][r] * (dt[jt][it][r] - Di) * scale;     // dS^T
      if (kq == 0) { sS[16 * it + li] = mx; sS[T + 16 * it + li] = inv; sS[2 * T + 16 * it + li] = Di; }
      // dQ rows of this query tile: A = dS^T tiles as they stand, B = K[j][e]
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < TT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dt[jt][it][r], sK[(16 * jt + 4 * kq + r) * ES + 16 * et + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dqkv[(row0 + (16 * it + 4 * kq + r) * rstep) * ld + h * E + 16 * et + li] = acc[r];
      }
    }
    // ---- pass 2: scores with lane column = key j; row statistics from LDS
#pragma unroll
    for (int jt = 0; jt < TT; ++jt) {
      f32x4 pt[TT], gt[TT];                        // [it]: P and dS tiles (rows i, column j)
#pragma unroll
      for (int it = 0; it < TT; ++it) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 q4 = *reinterpret_cast<const float4*>(sQ + (16 * it + li) * ES + 16 * ks + 4 * kq);
          const float4 k4 = *reinterpret_cast<const float4*>(sK + (16 * jt + li) * ES + 16 * ks + 4 * kq);
          const float4 o4 = *reinterpret_cast<const float4*>(sO + (16 * it + li) * ES + 16 * ks + 4 * kq);
          const float4 v4 = *reinterpret_cast<const float4*>(sV + (16 * jt + li) * ES + 16 * ks + 4 * kq);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.x, k4.x, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.x, v4.x, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.y, k4.y, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.y, v4.y, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.z, k4.z, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.z, v4.z, b, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.w, k4.w, a, 0, 0, 0);
          b = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.w, v4.w, b, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * it + 4 * kq + r;
          const float p = __expf(a[r] * scale - sS[i]) * sS[T + i];
          a[r] = p;
          b[r] = p * (b[r] - sS[2 * T + i]) * scale;
        }
        pt[it] = a; gt[it] = b;
      }
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = dk;
#pragma unroll
        for (int it = 0; it < TT; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * kq + r;
            dk = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[it][r], sQ[i * ES + 16 * et + li], dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_16x16x4f32(pt[it][r], sO[i * ES + 16 * et + li], dv, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = dqkv + (row0 + (16 * jt + 4 * kq + r) * rstep) * ld + h * E + 16 * et + li;
          dst[He] = dk[r];
          dst[2 * He] = dv[r];
        }
      }
    }
  }
}

// Text-tower attention backward on the f32 MFMA (T <= 80 tokens, head dim 64; CLIP's context length is 77).
// seq_attn_bwd_block_kernel above spends 85 us per launch on 16 workgroups at two classes per GPU -- one FMA per lane per
// (query, key, channel) on the VALU -- 12 times per step.  The five products of the backward are small GEMMs:
//   S = Q K^T, dP = dO V^T (T x T x 64),   dV = P^T dO, dQ = dS K, dK = dS^T Q (T x 64 x T)
// One workgroup per (sequence, head), eight waves, v_mfma_f32_16x16x4_f32 (77 -> 5 x 16 = 80: 4 % padding):
//   A  Q, K, V, dO staged once in LDS (rows padded to 68 floats: conflict-free ds_read_b128 across 16 rows);
//   B  the (i, j) score tiles -- lower triangle only when causal -- dealt round-robin to the waves: S and dP of a tile from
//      float4 fragment reads (both operands are k-contiguous), written to two T x T LDS matrices;
//   C  softmax rows (wave per row, lane per key): P, D_i = sum_j P_ij dP_ij, dS = P (dP - D_i) scale, in place;
//   D  the 3 x 5 x 4 output tiles, round-robin: dQ reads dS rows as float4 and K columns as dwords, dK / dV read dS / P
//      columns (consecutive lanes -> consecutive addresses); the k range is cut to the non-zero band when causal.
// Exact f32 products and fixed summation orders: results differ from the VALU kernels by f32 round-off only.
constexpr int AB_TP = 80, AB_LDP = 84, AB_ROWF = 68;
constexpr int AB_LDS_B = (4 * AB_TP * AB_ROWF + 2 * AB_TP * AB_LDP) * 4;        // 87,040 + 53,760 B
__global__ __launch_bounds__(512) void seq_attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int T, int heads, int causal,
                                                                float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_ab[];
  float* sQ = reinterpret_cast<float*>(smem_ab);
  float* sK = sQ + AB_TP * AB_ROWF;
  float* sV = sK + AB_TP * AB_ROWF;
  float* sO = sV + AB_TP * AB_ROWF;
  float* sP = sO + AB_TP * AB_ROWF;               // [TP][LDP]  S -> P
  float* sD = sP + AB_TP * AB_LDP;                // [TP][LDP]  dP -> dS
  const int h = blockIdx.x % heads;
  const int64_t seq = blockIdx.x / heads;
  const int He = heads * 64, ld = 3 * He;
  const float* base = qkv + seq * T * ld + h * 64;
  const float* dob = dout + seq * T * He + h * 64;
  float* dqb = dqkv + seq * T * ld + h * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;        // 8 waves: two per SIMD
  const int li = lane & 15, kq = lane >> 4;
  const int TT = (T + 15) >> 4;
  // ---- A: stage (rows >= T zero)
  for (int i = t; i < AB_TP * 16; i += 512) {
    const int row = i >> 4, c4 = i & 15;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4, o4 = q4;
    if (row < T) {
      const float* p = base + (int64_t)row * ld + 4 * c4;
      q4 = *reinterpret_cast<const float4*>(p);
      k4 = *reinterpret_cast<const float4*>(p + He);
      v4 = *reinterpret_cast<const float4*>(p + 2 * He);
      o4 = *reinterpret_cast<const float4*>(dob + (int64_t)row * He + 4 * c4);
    }
    *reinterpret_cast<float4*>(sQ + row * AB_ROWF + 4 * c4) = q4;
    *reinterpret_cast<float4*>(sK + row * AB_ROWF + 4 * c4) = k4;
    *reinterpret_cast<float4*>(sV + row * AB_ROWF + 4 * c4) = v4;
    *reinterpret_cast<float4*>(sO + row * AB_ROWF + 4 * c4) = o4;
  }
  __syncthreads();
  // ---- B: score tiles.  tile (ib, jb): lane holds column j = 16 jb + li, rows i = 16 ib + 4 kq + r
  {
    int n = 0;
    for (int ib = 0; ib < TT; ++ib) {
      const int jend = causal ? ib + 1 : TT;
      for (int jb = 0; jb < jend; ++jb, ++n) {
        if ((n & 7) != wave) continue;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = sa;
        const float* qa = sQ + (16 * ib + li) * AB_ROWF + 4 * kq;
        const float* oa = sO + (16 * ib + li) * AB_ROWF + 4 * kq;
        const float* kb = sK + (16 * jb + li) * AB_ROWF + 4 * kq;
        const float* vb = sV + (16 * jb + li) * AB_ROWF + 4 * kq;
        float4 q4[4], k4[4], o4[4], v4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          q4[s4] = *reinterpret_cast<const float4*>(qa + 16 * s4); k4[s4] = *reinterpret_cast<const float4*>(kb + 16 * s4);
          o4[s4] = *reinterpret_cast<const float4*>(oa + 16 * s4); v4[s4] = *reinterpret_cast<const float4*>(vb + 16 * s4);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[s4].x, k4[s4].x, sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(o4[s4].x, v4[s4].x, da, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[s4].y, k4[s4].y, sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(o4[s4].y, v4[s4].y, da, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[s4].z, k4[s4].z, sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(o4[s4].z, v4[s4].z, da, 0, 0, 0);
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[s4].w, k4[s4].w, sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(o4[s4].w, v4[s4].w, da, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sP[(16 * ib + 4 * kq + r) * AB_LDP + 16 * jb + li] = sa[r] * scale;
          sD[(16 * ib + 4 * kq + r) * AB_LDP + 16 * jb + li] = da[r];
        }
      }
    }
  }
  __syncthreads();
  // ---- C: softmax rows and dS, four rows per wave at a time: 16 lanes per row, lane li holds columns li, li + 16, ...
  // (<= 5 per lane), reductions over the 16-lane group only.  Columns of the computed band: j < jw; valid keys: j < jmax
  for (int i = 4 * wave + kq; i < 16 * TT; i += 32) {
    const int jmax = i < T ? (causal ? i + 1 : T) : 0;
    const int jw = causal ? 16 * ((i >> 4) + 1) : 16 * TT;
    float* pr = sP + i * AB_LDP;
    float* dr = sD + i * AB_LDP;
    float sv[5], dv[5];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int j = li + 16 * u;
      sv[u] = j < jmax ? pr[j] : -INFINITY;
      dv[u] = j < jmax ? dr[j] : 0.f;
      mx = fmaxf(mx, sv[u]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u) { sv[u] = li + 16 * u < jmax ? __expf(sv[u] - mx) : 0.f; sum += sv[u]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = jmax > 0 ? 1.f / sum : 0.f;
    float Di = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u) { sv[u] *= inv; Di += sv[u] * dv[u]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) Di += __shfl_xor(Di, o, 64);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int j = li + 16 * u;
      if (j < jw) { pr[j] = sv[u]; dr[j] = sv[u] * (dv[u] - Di) * scale; }
    }
  }
  __syncthreads();
  // ---- D: output tiles.  which 0: dQ (rows = queries), 1: dK, 2: dV (rows = keys); tile (rb, eb): lane holds channel
  // e = 16 eb + li of rows 16 rb + 4 kq + r.  The reduction walks 16-wide k blocks kb (uniform bounds: the non-zero band
  // when causal); every block's operands are requested before its four MFMAs
  {
    int n = 0;
    for (int which = 0; which < 3; ++which) {
      for (int rb = 0; rb < TT; ++rb) {
        const int kb_lo = (which != 0 && causal) ? rb : 0;                 // keys see queries i >= j only
        const int kb_hi = (which == 0 && causal) ? rb + 1 : TT;            // queries see keys j <= i only
        for (int eb = 0; eb < 4; ++eb, ++n) {
          if ((n & 7) != wave) continue;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if (which == 0) {
            // dQ[i][e] = sum_j dS[i][j] K[j][e]: a = dS row i (float4 along j), b = K[j][e] (one dword per MFMA)
            const float* ar = sD + (16 * rb + li) * AB_LDP + 4 * kq;
            const float* bc = sK + 16 * eb + li + 4 * kq * AB_ROWF;
#pragma unroll
            for (int kb = 0; kb < AB_TP / 16; ++kb) {
              if (kb < kb_lo || kb >= kb_hi) continue;
              const float4 a4 = *reinterpret_cast<const float4*>(ar + 16 * kb);
              const float* bb = bc + 16 * kb * AB_ROWF;
              const float b0 = bb[0], b1 = bb[AB_ROWF], b2 = bb[2 * AB_ROWF], b3 = bb[3 * AB_ROWF];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b2, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b3, acc, 0, 0, 0);
            }
          } else {
            // dK[j][e] = sum_i dS[i][j] Q[i][e];  dV[j][e] = sum_i P[i][j] dO[i][e]: both operands walk rows i = 16 kb + 4 u + kq
            const float* ac = (which == 1 ? sD : sP) + 16 * rb + li + kq * AB_LDP;
            const float* bc = (which == 1 ? sQ : sO) + 16 * eb + li + kq * AB_ROWF;
#pragma unroll
            for (int kb = 0; kb < AB_TP / 16; ++kb) {
              if (kb < kb_lo || kb >= kb_hi) continue;
              const float* aa = ac + 16 * kb * AB_LDP;
              const float* bb = bc + 16 * kb * AB_ROWF;
              const float a0 = aa[0], a1 = aa[4 * AB_LDP], a2 = aa[8 * AB_LDP], a3 = aa[12 * AB_LDP];
              const float b0 = bb[0], b1 = bb[4 * AB_ROWF], b2 = bb[8 * AB_ROWF], b3 = bb[12 * AB_ROWF];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rb + 4 * kq + r;
            if (row < T) dqb[(int64_t)row * ld + which * He + 16 * eb + li] = acc[r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ positional-embedding gradients
// d_pos0[n][e] = sum_{tile,l} dx[(tile,n,l)][e];  d_pos1[l][e] = sum_{tile,n} dx[(tile,n,l)][e]
// stage 1: block (tile, group of 4 segment rows n): p0[tile][n][e] = sum_l dx, p1[tile*ng + g][l][e] = sum_{n in group} dx;
// stage 2: acx_reduce_rows over the tiles (p0) and over the tiles x groups (p1).  Fixed order throughout.
__global__ __launch_bounds__(256) void pos_grad_part_kernel(const float* __restrict__ dx, float* __restrict__ p0,
                                                            float* __restrict__ p1, int gn, int gl, int E) {
  const int ng = (gn + 3) / 4;
  const int tile = blockIdx.x / ng, g = blockIdx.x - tile * ng;
  for (int e = threadIdx.x; e < E; e += 256) {
    for (int l0 = 0; l0 < gl; l0 += 16) {            // 16 l-accumulators in registers per pass
      float a1[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) a1[j] = 0.f;
      for (int n = 4 * g; n < min(4 * g + 4, gn); ++n) {
        float a0 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int l = l0 + j;
          if (l < gl) {
            const float v = dx[(((size_t)tile * gn + n) * gl + l) * E + e];
            a0 += v;
            a1[j] += v;
          }
        }
        float* o = p0 + ((size_t)tile * gn + n) * E + e;
        *o = (l0 ? *o : 0.f) + a0;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (l0 + j < gl) p1[((size_t)blockIdx.x * gl + l0 + j) * E + e] = a1[j];
    }
  }
}

// ------------------------------------------------------------------ BatchNorm (training) backward
// logits = (raw - mean) * rstd ;  part A (per column sums sdl[c] = sum dl, sdx[c] = sum dl*xhat): acx_head.hip, bn_partial_kernel<1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ logits, const float* __restrict__ dl,
                                                           const float* __restrict__ var_b, const float* __restrict__ sdl,
                                                           const float* __restrict__ sdx, float* __restrict__ draw, int ldo,
                                                           int64_t total, int C1, float eps, float inv_rows_host,
                                                           const float* __restrict__ total_rows_dev) {
  // one thread per element of draw [rows, ldo]: the pad columns c >= C1 (the TN product's 4-column granularity) are zeroed here
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= total) return;
  const int64_t r = j / ldo;
  const int c = (int)(j - r * ldo);
  if (c >= C1) { draw[j] = 0.f; return; }
  const int64_t i = r * C1 + c;
  const float inv_rows = total_rows_dev ? 1.f / total_rows_dev[0] : inv_rows_host;
  const float rstd = 1.f / sqrtf(var_b[c] + eps);
  draw[j] = rstd * (dl[i] - sdl[c] * inv_rows - logits[i] * sdx[c] * inv_rows);
}
// y = a*x + b*y  (running-statistics update of BatchNorm1d, momentum 0.1)
__global__ void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, int n, float a, float b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + b * y[i];
}
// per-block column partial sums (deterministic colsum): part[blk][D]
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t rows,
                                                          int D, int ld, int rows_per_block) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += x[r * ld + c];
    part[(size_t)blockIdx.x * D + c] = s;
  }
}
// the same partials for D % 4 == 0, 16-byte aligned rows: 64 float4 columns x 4 row slices per block, grid
// (row blocks, 256-column blocks); slice k sums rows r0+k, r0+k+4, ... and the four slices are added in order
__global__ __launch_bounds__(256) void colsum_part4_kernel(const float* __restrict__ x, float* __restrict__ part, int64_t rows,
                                                           int D, int ld, int rows_per_block) {
  __shared__ float4 red[4][64];
  const int q = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + 4 * q;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < D)
    for (int64_t r = r0 + sl; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[sl][q] = s;
  __syncthreads();
  if (sl == 0 && c < D) {
    float4 t = red[0][q];
#pragma unroll
    for (int k = 1; k < 4; ++k) { t.x += red[k][q].x; t.y += red[k][q].y; t.z += red[k][q].z; t.w += red[k][q].w; }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * D + c) = t;
  }
}

// d_text from d_dirs: dirs = v/|v|, v = text[src] - nc  (selector_model.py:44-59)
// nparts > 1: d_dirs arrives as the K-split partial images of the TN product that forms it (acx_gemm_tn_parts: image s at ddirs +
// s * part_stride), added here in image order exactly as tn_reduce_kernel adds them -- one launch less in the step's middle
__global__ __launch_bounds__(256) void text_dirs_bwd_kernel(const float* __restrict__ text, const float* __restrict__ nc,
                                                            const float* __restrict__ ddirs, float* __restrict__ dtext,
                                                            int C, int D, int normal_id, int nparts, int64_t part_stride) {
  __shared__ float red[4];
  const int row = blockIdx.x;              // row of text (0..C-1)
  if (row == normal_id) {
    for (int e = threadIdx.x; e < D; e += 256) dtext[(size_t)row * D + e] = 0.f;
    return;
  }
  const int c = row < normal_id ? row : row - 1;
  auto dd = [&](int e) {
    float v = ddirs[(size_t)c * D + e];
    for (int k = 1; k < nparts; ++k) v += ddirs[(size_t)k * part_stride + (size_t)c * D + e];
    return v;
  };
  // this thread's d_dirs values (columns t, t + 256, ...): summed once, kept in registers for both passes (D <= 2048; beyond: re-summed)
  float dv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) dv[q] = (int)threadIdx.x + 256 * q < D ? dd(threadIdx.x + 256 * q) : 0.f;
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = threadIdx.x + 256 * q;
    if (e < D) {
      const float v = text[(size_t)row * D + e] - nc[e];
      ss += v * v;
      dot += v * dv[q];
    }
  }
  for (int e = threadIdx.x + 2048; e < D; e += 256) {
    const float v = text[(size_t)row * D + e] - nc[e];
    ss += v * v;
    dot += v * dd(e);
  }
  const float tss = block_sum(ss, red);
  const float tdot = block_sum(dot, red);
  const float norm = sqrtf(tss);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = threadIdx.x + 256 * q;
    if (e < D) {
      const float v = text[(size_t)row * D + e] - nc[e];
      dtext[(size_t)row * D + e] = (dv[q] - v * tdot / tss) / norm;
    }
  }
  for (int e = threadIdx.x + 2048; e < D; e += 256) {
    const float v = text[(size_t)row * D + e] - nc[e];
    dtext[(size_t)row * D + e] = (dd(e) - v * tdot / tss) / norm;
  }
}

// ------------------------------------------------------------------ MIL top-k / bottom-k selection
// one block per video.  seg[n][c] = sum_l logits[(v,n,l)][c]; masked to -/+1e6; abnormal half: own-class
// column; normal half: sum over classes; k largest / smallest, ties -> lower index first.
// keys + picks of video v from its segment sums seg [N][C1] (LDS); the picks also land in pick[0 .. ktop + kbot) (LDS) when given
__device__ __forceinline__ void select_from_seg(float* seg, float* keyt, float* keyb, const int64_t* __restrict__ labels,
                                                const float* __restrict__ mask_top, const float* __restrict__ mask_bot,
                                                int64_t* __restrict__ idx_top, int64_t* __restrict__ idx_bot, int* pick, int v, int B,
                                                int N, int C1, int normal_id, int ktop, int kbot) {
  const bool abn = v < B / 2;
  int col = 0;
  if (abn) {
    const int64_t lab = labels[v];
    col = (int)(lab > normal_id ? lab - 1 : lab);
  }
  for (int n = threadIdx.x; n < N; n += 256) {
    float vt, vb;
    if (abn) {
      vt = mask_top[v * N + n] == 0.f ? -1e6f : seg[n * C1 + col];
      vb = mask_bot[v * N + n] == 0.f ? 1e6f : seg[n * C1 + col];
    } else {   // masked values are summed over the classes too (selector_model.py:127-130,152-153)
      float st = 0.f, sb = 0.f;
      const bool mt = mask_top[v * N + n] == 0.f, mb = mask_bot[v * N + n] == 0.f;
      for (int c = 0; c < C1; ++c) {
        st += mt ? -1e6f : seg[n * C1 + c];
        sb += mb ? 1e6f : seg[n * C1 + c];
      }
      vt = st; vb = sb;
    }
    keyt[n] = vt;
    keyb[n] = vb;
  }
  __syncthreads();
  // the k picks: a taken segment's key is overwritten in LDS (the earlier form re-read its own picks from GLOBAL memory inside
  // the serial loop: 44 us per launch for 3 + 3 picks out of 32); thread 0 picks the largest, thread 64 the smallest
  if (threadIdx.x == 0) {
    for (int k = 0; k < ktop; ++k) {
      int best = 0;
      for (int n = 1; n < N; ++n)
        if (keyt[n] > keyt[best]) best = n;                    // ties -> lower index first
      idx_top[v * ktop + k] = best;
      if (pick) pick[k] = best;
      keyt[best] = -__builtin_huge_valf();
    }
  } else if (threadIdx.x == 64) {
    for (int k = 0; k < kbot; ++k) {
      int best = 0;
      for (int n = 1; n < N; ++n)
        if (keyb[n] < keyb[best]) best = n;
      idx_bot[v * kbot + k] = best;
      if (pick) pick[ktop + k] = best;
      keyb[best] = __builtin_huge_valf();
    }
  }
}
__global__ __launch_bounds__(256) void select_idx_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ mask_top, const float* __restrict__ mask_bot,
                                                         int64_t* __restrict__ idx_top, int64_t* __restrict__ idx_bot,
                                                         int B, int N, int Lg, int C1, int normal_id, int ktop, int kbot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* seg = reinterpret_cast<float*>(smem);       // [N][C1]
  float* keyt = seg + N * C1;                        // [N]
  float* keyb = keyt + N;                            // [N]
  const int v = blockIdx.x;
  for (int i = threadIdx.x; i < N * C1; i += 256) {
    const int n = i / C1, c = i - n * C1;
    float s = 0.f;
    for (int l = 0; l < Lg; ++l) s += logits[(((size_t)v * N + n) * Lg + l) * C1 + c];   // torch.sum over dim 2, in order
    seg[i] = s;
  }
  __syncthreads();
  select_from_seg(seg, keyt, keyb, labels, mask_top, mask_bot, idx_top, idx_bot, nullptr, v, B, N, C1, normal_id, ktop, kbot);
}

// The selector's forward tail as ONE launch, one workgroup per video (the whole-step graph; selector_model.py:60-99,119-225):
// [SyncBN combine of the gathered statistics] -> BatchNorm of the video's rows -> [running statistics, block 0] -> segment sums ->
// top-k / bottom-k picks -> gather of the top-k segments' logits.  Every quantity is formed by the same expressions, in the same
// order, as in acx_bn_combine / acx_selector_bn / acx_bn_running_update / acx_select_idx / acx_gather_segments: bit-identical
// results (tests/test_gpu_train.py::test_selector_tail_equals_separate_launches), five launches (four without SyncBN) less.
struct SelTailArgs {
  const float* raw;                                   // [rows][C1]
  const float* gathered; int R;                       // SyncBN: [R][2 C1 + 1] (nullptr: mean_in / var_b_in / var_u_in)
  const float *mean_in, *var_b_in, *var_u_in;
  float* stat_out;                                    // SyncBN: [3 C1 + 1] mean | biased var | unbiased var | total rows
  float *rm, *rv; long long* nbt; float momentum, om; // running statistics (nullptr: none)
  float* logits; int64_t ldl;
  const int64_t* labels; const float *mask_top, *mask_bot;
  int64_t *idx_top, *idx_bot; float* logits_topk;
  int B, N, Lg, C1, normal_id, ktop, kbot; float eps;
};
__global__ __launch_bounds__(256) void selector_tail_kernel(const SelTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, Lg = a.Lg, C1 = a.C1, per = N * Lg;
  float* slog = reinterpret_cast<float*>(smem);      // [N Lg][C1] this video's logits
  float* seg = slog + (size_t)per * C1;              // [N][C1]
  float* keyt = seg + N * C1;                        // [N]
  float* keyb = keyt + N;                            // [N]
  float* smean = keyb + N;                           // [C1]
  float* svar = smean + C1;                          // [C1]
  int* pick = reinterpret_cast<int*>(svar + C1);     // [ktop + kbot]
  const int v = blockIdx.x;
  if ((int)threadIdx.x < C1) {
    const int c = threadIdx.x;
    float n = 0.f, m, vb, vu;
    if (a.gathered) {
      acx_bn_combine_col(a.gathered, a.R, C1, c, n, m, vb, vu);
      if (v == 0) {
        a.stat_out[c] = m; a.stat_out[C1 + c] = vb; a.stat_out[2 * C1 + c] = vu;
        if (c == 0) a.stat_out[3 * C1] = n;
      }
    } else {
      m = a.mean_in[c]; vb = a.var_b_in[c]; vu = a.var_u_in[c];
    }
    smean[c] = m; svar[c] = vb;
    if (v == 0 && a.rm) {
      a.rm[c] = acx_bn_running(a.momentum, a.om, m, a.rm[c]);
      a.rv[c] = acx_bn_running(a.momentum, a.om, vu, a.rv[c]);
      if (c == 0 && a.nbt) *a.nbt += 1;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < per * C1; i += 256) {
    const int rl = i / C1, c = i - rl * C1;
    const int64_t r = (int64_t)v * per + rl;
    const float y = acx_bn_apply(a.raw[r * C1 + c], smean[c], svar[c], a.eps);
    a.logits[r * a.ldl + c] = y;
    slog[i] = y;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N * C1; i += 256) {
    const int n = i / C1, c = i - n * C1;
    float s = 0.f;
    for (int l = 0; l < Lg; ++l) s += slog[((size_t)n * Lg + l) * C1 + c];   // torch.sum over dim 2, in order
    seg[i] = s;
  }
  __syncthreads();
  select_from_seg(seg, keyt, keyb, a.labels, a.mask_top, a.mask_bot, a.idx_top, a.idx_bot, pick, v, a.B, N, C1, a.normal_id, a.ktop,
                  a.kbot);
  __syncthreads();
  // logits_topk[(v K + k) Lg + l][:] = logits[(v, idx_top[v][k], l)][:]
  for (int i = threadIdx.x; i < a.ktop * Lg * C1; i += 256) {
    const int k = i / (Lg * C1), rem = i - k * Lg * C1;
    a.logits_topk[((size_t)v * a.ktop + k) * Lg * C1 + rem] = slog[(size_t)pick[k] * Lg * C1 + rem];
  }
}

// out[(v*K + k)*Lg + l][:] = logits[(v, idx[v][k], l)][:]      (selector_model.py:160-225)
__global__ __launch_bounds__(256) void gather_segments_kernel(const float* __restrict__ logits, const int64_t* __restrict__ idx,
                                                              float* __restrict__ out, int64_t total, int N, int Lg, int C1, int K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C1);
  const int64_t r = i / C1;
  const int l = (int)(r % Lg);
  const int64_t vk = r / Lg;
  const int64_t v = vk / K;
  const int64_t n = idx[vk];
  out[i] = logits[(((size_t)v * N + n) * Lg + l) * C1 + c];
}
// dlogits[(v, idx[v][k], l)][:] += dout[(v*K+k)*Lg + l][:]    (segments of one video are distinct)
__global__ __launch_bounds__(256) void scatter_segments_kernel(const float* __restrict__ dout, const int64_t* __restrict__ idx,
                                                               float* __restrict__ dlogits, int64_t total, int N, int Lg, int C1, int K) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C1);
  const int64_t r = i / C1;
  const int l = (int)(r % Lg);
  const int64_t vk = r / Lg;
  const int64_t v = vk / K;
  const int64_t n = idx[vk];
  dlogits[(((size_t)v * N + n) * Lg + l) * C1 + c] += dout[i];
}

// ------------------------------------------------------------------ MIL loss forward + backward (loss.py:51-195)
struct LossArgs {
  const float* sim; const float* sim_topk; const int64_t* labels; const float* scores;
  const int64_t* idx_topk_abn; const int64_t* idx_topk_nor; const int64_t* idx_bottomk_abn;
  float* dsim; float* dsim_topk; float* dscores; float* part;
  int B, N, Lg, C1, K, normal_id;
  float l_dir_abn, l_dir_nor, l_topk_abn, l_bottomk_abn, l_topk_nor, l_smooth, l_sparse;
  const float* gout_ptr;   // device scalar: upstream gradient of the total cost (NULL -> 1)
};
// thread per frame row; block partial sums of the 7 terms -> part[blk][8].  PUBLISH: the partials leave as agent-scope
// (write-through) stores for the one-launch form, whose last-arriving block adds them up (see loss_fused_kernel)
// FOLD (acx_mil_loss_bn): the gradient of the gathered top-k rows (loss_topk_body's dsim_topk: the constant -l_dir_abn gout / RA in the
// own-class column of an abnormal video's top-k segments, zero elsewhere) is added to this row's dsim HERE, after the row's own terms --
// the value and the order of acx_scatter_segments(dsim, dsim_topk, idx_topk) -- and the finished row is mirrored into sdl [256][C1] (LDS)
template <bool PUBLISH, bool FOLD = false>
__device__ __forceinline__ void loss_rows_body(const LossArgs& a, const int blk, float* red, float* sdl = nullptr) {
  const int64_t R = (int64_t)a.B * a.N * a.Lg;
  const int64_t r = (int64_t)blk * 256 + threadIdx.x;
  const float gout = a.gout_ptr ? a.gout_ptr[0] : 1.f;
  float t_dir_nor = 0.f, t_topk_abn = 0.f, t_bot_abn = 0.f, t_topk_nor = 0.f, t_smooth = 0.f, t_sparse = 0.f;
  if (r < R) {
    const int C1 = a.C1;
    const int64_t per = (int64_t)a.N * a.Lg;
    const int v = (int)(r / per);
    const int n = (int)((r / a.Lg) % a.N);
    const bool abn = v < a.B / 2;
    const float* s = a.sim + r * C1;
    const float sc = a.scores[r];
    float ds = 0.f;
    float* dsim = a.dsim + r * C1;       // this thread owns row r of the gradient
    for (int c = 0; c < C1; ++c) dsim[c] = 0.f;
    const float cntK = (float)((int64_t)(a.B / 2) * a.K * a.Lg);
    if (abn) {
      const int64_t lab = a.labels[v];
      const int y = (int)(lab > a.normal_id ? lab - 1 : lab);
      bool in_top = false, in_bot = false;
      for (int k = 0; k < a.K; ++k) {
        in_top |= a.idx_topk_abn[v * a.K + k] == n;
        in_bot |= a.idx_bottomk_abn[v * a.K + k] == n;
      }
      if (in_top) {   // NLL(log(softmax(sim)[y] * score))
        float mx = -INFINITY;
        for (int c = 0; c < C1; ++c) mx = fmaxf(mx, s[c]);
        float sum = 0.f;
        for (int c = 0; c < C1; ++c) sum += expf(s[c] - mx);
        const float py = expf(s[y] - mx) / sum;
        t_topk_abn = -logf(py * sc);
        const float w = a.l_topk_abn / cntK * gout;
        for (int c = 0; c < C1; ++c) dsim[c] += w * (expf(s[c] - mx) / sum - (c == y ? 1.f : 0.f));
        ds += -w / sc;
        if constexpr (FOLD) dsim[y] += -a.l_dir_abn * gout / cntK;      // (cntK == (float)RA of loss_topk_body)
      }
      if (in_bot) {   // NLL(log(1 - score))
        t_bot_abn = -logf(1.f - sc);
        ds += a.l_bottomk_abn / cntK * gout / (1.f - sc);
      }
      // smoothness over the FLATTENED abnormal scores (crosses video boundaries) + sparsity
      const int64_t Ta = (int64_t)(a.B / 2) * per;
      float g = 0.f;
      if (r + 1 < Ta) { const float d1 = a.scores[r + 1] - sc; t_smooth = d1 * d1; g -= 2.f * d1; }
      if (r > 0) { g += 2.f * (sc - a.scores[r - 1]); }
      ds += a.l_smooth * gout * g;
      t_sparse = sc;
      ds += a.l_sparse * gout / (float)Ta;
    } else {
      const int vn = v - a.B / 2;
      // max over directions, mean over normal frames
      int am = 0;
      for (int c = 1; c < C1; ++c) if (s[c] > s[am]) am = c;
      t_dir_nor = s[am];
      dsim[am] += a.l_dir_nor * gout / (float)((int64_t)(a.B - a.B / 2) * per);
      bool in_top = false;
      for (int k = 0; k < a.K; ++k) in_top |= a.idx_topk_nor[vn * a.K + k] == n;
      if (in_top) {
        t_topk_nor = -logf(1.f - sc);
        ds += a.l_topk_nor / (float)((int64_t)(a.B - a.B / 2) * a.K * a.Lg) * gout / (1.f - sc);
      }
    }
    a.dscores[r] = ds;
    if constexpr (FOLD)
      for (int c = 0; c < C1; ++c) sdl[threadIdx.x * C1 + c] = dsim[c];
  }
  float vals[6] = {t_dir_nor, t_topk_abn, t_bot_abn, t_topk_nor, t_smooth, t_sparse};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float tot = block_sum(vals[k], red);
    if (threadIdx.x == 0) {
      if (PUBLISH) __hip_atomic_store(a.part + (size_t)blk * 8 + k, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else a.part[(size_t)blk * 8 + k] = tot;
    }
  }
}
__global__ __launch_bounds__(256) void loss_rows_kernel(const LossArgs a) {
  __shared__ float red[4];
  loss_rows_body<false>(a, (int)blockIdx.x, red);
}
// thread per row of sim_topk (abnormal part only contributes): ldir_abn = -mean(own-class logit)
template <bool PUBLISH, bool FOLD = false>
__device__ __forceinline__ void loss_topk_body(const LossArgs& a, float* __restrict__ part2, const int blk, float* red) {
  const int64_t RT = (int64_t)a.B * a.K * a.Lg;
  const int64_t RA = (int64_t)(a.B / 2) * a.K * a.Lg;
  const int64_t r = (int64_t)blk * 256 + threadIdx.x;
  const float gout = a.gout_ptr ? a.gout_ptr[0] : 1.f;
  float t = 0.f;
  if (r < RT) {
    if constexpr (!FOLD)
      for (int c = 0; c < a.C1; ++c) a.dsim_topk[r * a.C1 + c] = 0.f;
    if (r < RA) {
      const int v = (int)(r / ((int64_t)a.K * a.Lg));
      const int64_t lab = a.labels[v];
      const int y = (int)(lab > a.normal_id ? lab - 1 : lab);
      t = a.sim_topk[r * a.C1 + y];
      if constexpr (!FOLD) a.dsim_topk[r * a.C1 + y] = -a.l_dir_abn * gout / (float)RA;
    }
  }
  const float tot = block_sum(t, red);
  if (threadIdx.x == 0) {
    if (PUBLISH) __hip_atomic_store(part2 + blk, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else part2[blk] = tot;
  }
}
__global__ __launch_bounds__(256) void loss_topk_kernel(const LossArgs a, float* __restrict__ part2) {
  __shared__ float red[4];
  loss_topk_body<false>(a, part2, (int)blockIdx.x, red);
}
// final: losses[8] = (cost, ldir_abn, ldir_nor, ltopk_abn, lbottomk_abn, ltopk_nor, lsmooth, lsparse)
__global__ void loss_final_kernel(const LossArgs a, const float* __restrict__ part, int nparts, const float* __restrict__ part2,
                                  int nparts2, float* __restrict__ losses) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t[6] = {0, 0, 0, 0, 0, 0}, d = 0;
  for (int p = 0; p < nparts; ++p)
    for (int k = 0; k < 6; ++k) t[k] += part[(size_t)p * 8 + k];
  for (int p = 0; p < nparts2; ++p) d += part2[p];
  const double per = (double)a.N * a.Lg, Bh = a.B / 2, Bn = a.B - a.B / 2;
  const float ldir_abn = (float)(-a.l_dir_abn * d / (Bh * a.K * a.Lg));
  const float ldir_nor = (float)(a.l_dir_nor * t[0] / (Bn * per));
  const float ltopk_abn = (float)(a.l_topk_abn * t[1] / (Bh * a.K * a.Lg));
  const float lbot_abn = (float)(a.l_bottomk_abn * t[2] / (Bh * a.K * a.Lg));
  const float ltopk_nor = (float)(a.l_topk_nor * t[3] / (Bn * a.K * a.Lg));
  const float lsmooth = (float)(a.l_smooth * t[4]);
  const float lsparse = (float)(a.l_sparse * t[5] / (Bh * per));
  losses[1] = ldir_abn; losses[2] = ldir_nor; losses[3] = ltopk_abn; losses[4] = lbot_abn;
  losses[5] = ltopk_nor; losses[6] = lsmooth; losses[7] = lsparse;
  losses[0] = ldir_abn + ldir_nor + ltopk_abn + lbot_abn + ltopk_nor + lsmooth + lsparse;
}

// the eight loss terms from the 7 x 32 lane sums of the last arriver (thread 0): a fixed tree; meter != nullptr: meter += losses
__device__ __forceinline__ void loss_terms_from_sums(const LossArgs& a, const double (*fin)[32], float* __restrict__ losses,
                                                     float* __restrict__ meter) {
  double t[7];
  for (int q = 0; q < 7; ++q) {
    double sacc = 0.0;
    for (int l = 0; l < 32; ++l) sacc += fin[q][l];
    t[q] = sacc;
  }
  const double d = t[6];
  const double per = (double)a.N * a.Lg, Bh = a.B / 2, Bn = a.B - a.B / 2;
  const float ldir_abn = (float)(-a.l_dir_abn * d / (Bh * a.K * a.Lg));
  const float ldir_nor = (float)(a.l_dir_nor * t[0] / (Bn * per));
  const float ltopk_abn = (float)(a.l_topk_abn * t[1] / (Bh * a.K * a.Lg));
  const float lbot_abn = (float)(a.l_bottomk_abn * t[2] / (Bh * a.K * a.Lg));
  const float ltopk_nor = (float)(a.l_topk_nor * t[3] / (Bn * a.K * a.Lg));
  const float lsmooth = (float)(a.l_smooth * t[4]);
  const float lsparse = (float)(a.l_sparse * t[5] / (Bh * per));
  float out[8];
  out[1] = ldir_abn; out[2] = ldir_nor; out[3] = ltopk_abn; out[4] = lbot_abn;
  out[5] = ltopk_nor; out[6] = lsmooth; out[7] = lsparse;
  out[0] = ldir_abn + ldir_nor + ltopk_abn + lbot_abn + ltopk_nor + lsmooth + lsparse;
  for (int i = 0; i < 8; ++i) {
    losses[i] = out[i];
    if (meter) meter[i] = 1.f * out[i] + 1.f * meter[i];       // acx_axpby(meter, losses, 1, 1)
  }
}

// The three launches above as ONE: blocks [0, np1) run the frame rows, blocks [np1, np1 + np2) the top-k rows; every block
// publishes its partial sums write-through (no release fence: a device-scope fence would write back the whole L2 of its
// XCD), bumps a relaxed arrival counter, and the last block to arrive forms the eight loss terms: 7 x 32 lanes add the parts
// p = lane, lane + 32, ... in f64, the 32 lane sums of a term are added in lane order -- a fixed tree.
__global__ __launch_bounds__(256) void loss_fused_kernel(const LossArgs a, float* __restrict__ part2, int np1, int np2,
                                                         float* __restrict__ losses, unsigned int* __restrict__ counter) {
  __shared__ float red[4];
  __shared__ double fin[7][32];
  __shared__ bool last;
  const int b = (int)blockIdx.x;
  if (b < np1) loss_rows_body<true>(a, b, red);
  else loss_topk_body<true>(a, part2, b - np1, red);
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_HANDOFF_RELEASE();
    last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(np1 + np2) - 1;
  }
  __syncthreads();
  if (!last) return;
  ACX_HANDOFF_ACQUIRE();
  const int k = threadIdx.x >> 5, j = threadIdx.x & 31;
  if (k < 7) {
    double acc = 0.0;
    if (k < 6) {
      for (int p = j; p < np1; p += 32)
        acc += (double)__hip_atomic_load(a.part + (size_t)p * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int p = j; p < np2; p += 32) acc += (double)__hip_atomic_load(part2 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    fin[k][j] = acc;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  loss_terms_from_sums(a, fin, losses, nullptr);
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acx_mil_loss_bn -- the loss, its gradients and the selector BatchNorm's backward statistics as ONE launch (whole-step graph):
//   blocks [0, np1): frame rows (loss_rows_body<.., FOLD>: dsim already holds the scattered top-k gradient = dlogits), then the
//     f64 column sums (dl, dl * xhat) of the SAME 256 rows -- slab b of bn_partial_kernel<1> when rows % 256 == 0 and rows / 256 <= 512
//     (dlogits from LDS, the arithmetic and the order of acx_bn_bwd_stats);
//   blocks [np1, np1 + np2): the gathered top-k rows (their term of the loss; their gradient is folded into the frame rows);
//   the last block to arrive forms the eight loss terms, adds them to the module's running sums (meter: y = 1 x + 1 y of acx_axpby) and
//     the BatchNorm sums (bn_pair_total: acx_bn_bwd_stats' stage 2).
// Replaces acx_mil_loss_one + acx_axpby + acx_scatter_segments + acx_bn_bwd_stats (two launches): five launches -> one, same bits.
__global__ __launch_bounds__(256) void loss_bn_kernel(const LossArgs a, float* __restrict__ part2, int np1, int np2,
                                                      float* __restrict__ losses, float* __restrict__ meter, double* __restrict__ bn_part,
                                                      float* __restrict__ bn_sums, unsigned int* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sdl = reinterpret_cast<float*>(smem);       // [256][C1] dlogits of this block's rows
  __shared__ float red[4];
  __shared__ double fin[7][32];
  __shared__ double redd[2][256];
  __shared__ bool last;
  const int b = (int)blockIdx.x, C1 = a.C1;
  constexpr int CP = 64;
  if (b < np1) {
    loss_rows_body<true, true>(a, b, red, sdl);
    __syncthreads();
    const int RL = 256 / C1;
    const int c = threadIdx.x % C1, rl = threadIdx.x / C1;
    const int64_t r0 = (int64_t)b * 256;
    double s = 0.0, qq = 0.0;
    if (rl < RL) {
      for (int rr = rl; rr < 256; rr += RL) {
        const double v = (double)a.sim[(r0 + rr) * C1 + c];
        const double g = (double)sdl[rr * C1 + c];
        acx_bn_acc<1>(s, qq, v, g);
      }
    }
    redd[0][threadIdx.x] = s;
    redd[1][threadIdx.x] = qq;
    __syncthreads();
    if ((int)threadIdx.x < 2 * C1) {
      const int k = threadIdx.x / C1, cc = threadIdx.x - k * C1;
      double v = 0.0;
      for (int l = 0; l < RL; ++l) v += redd[k][l * C1 + cc];
      __hip_atomic_store(bn_part + ((size_t)b * 2 + k) * CP + cc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every lane's published partials have left before the barrier below
    __syncthreads();
  } else {
    loss_topk_body<true, true>(a, part2, b - np1, red);
  }
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ACX_HANDOFF_RELEASE();
    last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(np1 + np2) - 1;
  }
  __syncthreads();
  if (!last) return;
  ACX_HANDOFF_ACQUIRE();
  {
    const int k = threadIdx.x >> 5, j = threadIdx.x & 31;
    if (k < 7) {
      double acc = 0.0;
      if (k < 6) {
        for (int p = j; p < np1; p += 32)
          acc += (double)__hip_atomic_load(a.part + (size_t)p * 8 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int p = j; p < np2; p += 32) acc += (double)__hip_atomic_load(part2 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fin[k][j] = acc;
    }
  }
  __syncthreads();
  const double tot = bn_pair_total<true>(bn_part, np1, CP, C1, &redd[0][0]);
  if ((int)threadIdx.x < 2 * C1) bn_sums[threadIdx.x] = (float)tot;
  if (threadIdx.x != 0) return;
  loss_terms_from_sums(a, fin, losses, meter);
  __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW, no amsgrad)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float pi = p[i];
  const float gi = g[i];
  pi *= 1.f - lr * wd;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

// Multi-tensor AdamW: ONE launch updates up to ADAMW_MAX_SEG parameter tensors (the reference's four param groups differ in
// lr only, anomaly_clip_module.py:693-746).  The segment table travels in the kernel arguments (no device-side table, no
// extra copy, replayable inside a HIP graph as long as lr / step do not change); block b finds its segment by a binary
// search over the prefix of 1024-element chunks and updates 4 x 256 elements with coalesced dword accesses (the flat
// gradient buffer hands out 4-byte-aligned views, so 16-byte accesses are not available for g).
constexpr int ADAMW_MAX_SEG = 48;
struct AdamwSegs {
  float* p[ADAMW_MAX_SEG];
  const float* g[ADAMW_MAX_SEG];
  float* m[ADAMW_MAX_SEG];
  float* v[ADAMW_MAX_SEG];
  long long n[ADAMW_MAX_SEG];
  float decay[ADAMW_MAX_SEG];         // 1 - lr * weight_decay
  float step_size[ADAMW_MAX_SEG];     // lr / (1 - beta1^step)
  int chunk0[ADAMW_MAX_SEG + 1];      // first 1024-element chunk of every segment
  int nseg;
};
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwSegs t, float b1, float b2, float omb1, float omb2, float eps,
                                                          float bc2_sqrt) {
  const int b = blockIdx.x;
  int lo = 0, hi = t.nseg - 1;
  while (lo < hi) {                                                    // last segment with chunk0 <= b (uniform per block)
    const int mid = (lo + hi + 1) >> 1;
    if (t.chunk0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int sgi = lo;
  float* __restrict__ p = t.p[sgi];
  const float* __restrict__ g = t.g[sgi];
  float* __restrict__ m = t.m[sgi];
  float* __restrict__ v = t.v[sgi];
  const long long n = t.n[sgi];
  const float decay = t.decay[sgi], step_size = t.step_size[sgi];
  const long long base = (long long)(b - t.chunk0[sgi]) * 1024 + threadIdx.x;
  float pi[4], gi[4], mi[4], vi[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) { pi[k] = p[i]; gi[k] = g[i]; mi[k] = m[i]; vi[k] = v[i]; }
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
    }
  }
}

// Multi-tensor y += a * x: ONE launch accumulates a list of gradient tensors into their destinations (the views of
// parallel.GradBuckets.flat) -- what autograd's AccumulateGrad does with one elementwise launch per parameter.
constexpr int AXPY_MAX_SEG = 96;
struct AxpySegs {
  float* y[AXPY_MAX_SEG];
  const float* x[AXPY_MAX_SEG];
  long long n[AXPY_MAX_SEG];
  int chunk0[AXPY_MAX_SEG + 1];
  int nseg;
};
__global__ __launch_bounds__(256) void multi_axpy_kernel(const AxpySegs t, float a) {
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
  float yv[4], xv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) { yv[k] = y[i]; xv[k] = x[i]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + 256 * k;
    if (i < n) y[i] = yv[k] + a * xv[k];
  }
}

// d_ctx[c][t][:] = dx[c][1+t][:]  (or summed over classes when the context is shared)  (coop.py:74-90)
__global__ __launch_bounds__(256) void ctx_grad_kernel(const float* __restrict__ dx, float* __restrict__ dctx, int C, int n_ctx,
                                                       int Lc, int W, int shared_ctx) {
  const int64_t total = (int64_t)(shared_ctx ? 1 : C) * n_ctx * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int w = (int)(i % W);
  const int tk = (int)((i / W) % n_ctx);
  const int c = (int)(i / ((int64_t)W * n_ctx));
  if (!shared_ctx) { dctx[i] = dx[((size_t)c * Lc + 1 + tk) * W + w]; return; }
  float s = 0.f;
  for (int cc = 0; cc < C; ++cc) s += dx[((size_t)cc * Lc + 1 + tk) * W + w];
  dctx[i] = s;
}

// out[idx[i], :] = src[i, :]  into a zero-filled [rows, W] buffer (backward of acx_gather_rows; idx distinct)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                           float* __restrict__ out, int64_t total4, int W) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int W4 = W / 4;
  const int64_t r = i / W4;
  const int w4 = (int)(i - r * W4);
  *reinterpret_cast<float4*>(out + idx[r] * W + 4 * w4) = *reinterpret_cast<const float4*>(src + r * W + 4 * w4);
}

}  // namespace

#define COMMA ,
#define DISPATCH_VPL(D, CALL)                          \
  switch ((D) / 64) {                                  \
    case 1: { constexpr int V = 1; CALL; } break;      \
    case 2: { constexpr int V = 2; CALL; } break;      \
    case 4: { constexpr int V = 4; CALL; } break;      \
    case 8: { constexpr int V = 8; CALL; } break;      \
    case 12: { constexpr int V = 12; CALL; } break;    \
    case 16: { constexpr int V = 16; CALL; } break;    \
    default: return acx_fail(ctx, ACX_E_UNSUPPORTED, "row width %s%ld not in {64,128,256,512,768,1024}", "", (long)(D)); \
  }
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int acx_reduce_rows(acx_ctx* ctx, const float* part, float* out, int32_t nparts, int32_t width, void* stream) {
  if (!part || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_reduce_rows: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((width + 15) / 16)), dim3(256), 0, (hipStream_t)stream, part, out, nparts, width);
  ACX_CHECK_LAUNCH(ctx, "acx_reduce_rows");
  return ACX_OK;
}

// Rows one wave walks in the kernels that leave per-block parameter-gradient partials (LayerNorm backward, classifier
// backward): 16 at >= 32 k rows (64 rows per block, 512 blocks), fewer below so that a data-parallel rank's 4096 rows still
// fill the chip (rows / 2048 rounded down to a power of two: 4096 rows -> 2 rows per wave, 512 blocks; was 64 blocks of 16
// sequential rows per wave = 28 us for a 4 MB pass).  acx_row_parts(rows) = the number of partial rows those kernels write.
static inline int acx_rows_per_wave(int64_t rows) {
  int r = 16;
  while (r > 1 && rows < (int64_t)2048 * r) r >>= 1;
  return r;
}
extern "C" int64_t acx_row_parts(int64_t rows) {
  if (rows <= 0) return 0;
  const int rpw = acx_rows_per_wave(rows);
  return (rows + 4 * rpw - 1) / (4 * rpw);
}

extern "C" int acx_layernorm_bwd(acx_ctx* ctx, const float* x, const float* w, const float* dy, float* dx, float* part,
                                 int64_t rows, int32_t D, float eps, int32_t mode, float dx_scale, const float* add,
                                 void* stream) {
  if (!x || !w || !dy) return acx_fail(ctx, ACX_E_BADARG, "acx_layernorm_bwd: null pointer%s");
  if (add && !dx) return acx_fail(ctx, ACX_E_BADARG, "acx_layernorm_bwd: add without dx%s");
  if (rows <= 0) return ACX_OK;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  const int rpw = part ? acx_rows_per_wave(rows) : 1;
  const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
  DISPATCH_VPL(D, layernorm_bwd_kernel<V><<<grid COMMA block COMMA 0 COMMA s>>>(x, w, dy, dx, part, rows, eps, mode, dx_scale, add, rpw));
  ACX_CHECK_LAUNCH(ctx, "acx_layernorm_bwd");
  return ACX_OK;
}

extern "C" int acx_cls_head_bwd(acx_ctx* ctx, const float* x1, const float* x2, const float* ln_w, const float* ln_b,
                                const float* lin_w, const float* scores, const float* dscores, float* dx, float* part,
                                int64_t rows, int32_t E, void* stream) {
  if (!x1 || !x2 || !ln_w || !ln_b || !lin_w || !scores || !dscores || !dx || !part)
    return acx_fail(ctx, ACX_E_BADARG, "acx_cls_head_bwd: null pointer%s");
  if (rows <= 0) return ACX_OK;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_NORM, s);
  const int rpw = acx_rows_per_wave(rows);
  const dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), block(256);
  DISPATCH_VPL(E, cls_head_bwd_kernel<V><<<grid COMMA block COMMA 0 COMMA s>>>(x1, x2, ln_w, ln_b, lin_w, scores, dscores, dx, part, rows, rpw));
  ACX_CHECK_LAUNCH(ctx, "acx_cls_head_bwd");
  return ACX_OK;
}

extern "C" int acx_act(acx_ctx* ctx, const float* saved, const float* d, float* out, int64_t n, int32_t mode, void* stream) {
  if (!saved || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_act: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (n % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_act: n%%4%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(act_kernel, GRID1(n / 4), dim3(256), 0, (hipStream_t)stream, saved, d, out, n / 4, mode);
  ACX_CHECK_LAUNCH(ctx, "acx_act");
  return ACX_OK;
}

extern "C" int acx_leaky_grad_planes(acx_ctx* ctx, const void* u_hi, const float* d, float* out, void* planes, int64_t plane_elems,
                                     int64_t n, void* stream) {
  if (!u_hi || !d || !out || !planes) return acx_fail(ctx, ACX_E_BADARG, "acx_leaky_grad_planes: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (n % 4 || plane_elems % 4 || plane_elems < n) return acx_fail(ctx, ACX_E_BADARG, "acx_leaky_grad_planes: n %% 4, plane_elems >= n%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(leaky_grad_planes_kernel, GRID1(n / 4), dim3(256), 0, (hipStream_t)stream, (const u16*)u_hi, d, out, (u16*)planes,
                     plane_elems, n / 4);
  ACX_CHECK_LAUNCH(ctx, "acx_leaky_grad_planes");
  return ACX_OK;
}

extern "C" int acx_add(acx_ctx* ctx, const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_add: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (n % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_add: n%%4%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(add_kernel, GRID1(n / 4), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
  ACX_CHECK_LAUNCH(ctx, "acx_add");
  return ACX_OK;
}

extern "C" int acx_transpose(acx_ctx* ctx, const float* in, float* out, int32_t R, int32_t Cn, void* stream) {
  if (!in || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_transpose: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(transpose_kernel, dim3((Cn + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, Cn);
  ACX_CHECK_LAUNCH(ctx, "acx_transpose");
  return ACX_OK;
}

extern "C" int acx_conv_weight_dx(acx_ctx* ctx, const float* w, float* out, int32_t Cout, int32_t Cin, void* stream) {
  if (!w || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_conv_weight_dx: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const int64_t total = (int64_t)Cout * Cin * 9;
  hipLaunchKernelGGL(conv_w_dx_kernel, GRID1(total), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin);
  ACX_CHECK_LAUNCH(ctx, "acx_conv_weight_dx");
  return ACX_OK;
}

extern "C" int acx_seq_attention_bwd(acx_ctx* ctx, const float* qkv, const float* dout, float* dqkv, int32_t tiles, int32_t gn,
                                     int32_t gl, int32_t heads, int32_t e, int32_t axis, int32_t causal, float* stats_ws, void* stream) {
  if (!qkv || !dout || !dqkv) return acx_fail(ctx, ACX_E_BADARG, "acx_seq_attention_bwd: null pointer%s");
  if (tiles <= 0) return ACX_OK;
  const int T = axis == 0 ? gn : gl;
  const bool sab_rows = ACX_DBG_SWITCH("SAB_ROWS", false);    // keep the two-launch rows kernel (A/B, debug builds)
  if (e == 64 && gn == 1 && axis == 1 && T <= AB_TP && !sab_rows && ACX_DBG_SWITCH("SAB_MFMA", true)) {
    // text-tower shape up to 80 tokens: the five products on the f32 MFMA, one workgroup per (sequence, head)
    hipStream_t s4 = (hipStream_t)stream;
    AcxProfScope prof4__(ctx, ACX_K_ATTN, s4);
    (void)hipFuncSetAttribute((const void*)seq_attn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AB_LDS_B);
    hipLaunchKernelGGL(seq_attn_bwd_mfma_kernel, dim3((unsigned)(tiles * heads)), dim3(512), (size_t)AB_LDS_B, s4, qkv, dout, dqkv, T,
                       heads, causal, 0.125f);
    ACX_CHECK_LAUNCH(ctx, "acx_seq_attention_bwd(mfma)");
    return ACX_OK;
  }
  if (e == 64 && gn == 1 && axis == 1 && T <= 128 && !sab_rows) {
    // text-tower shape: one workgroup per (sequence, head), operands staged once in LDS, both passes in one launch
    hipStream_t s3 = (hipStream_t)stream;
    AcxProfScope prof3__(ctx, ACX_K_ATTN, s3);
    const size_t lds3 = ((size_t)4 * T * SB_ROWF + 3 * T + 4 + 8 * 2 * ((T + 3) & ~3)) * sizeof(float);
    const dim3 grid3((unsigned)(tiles * heads)), block3(512);
    if (T <= 64) {
      (void)hipFuncSetAttribute((const void*)seq_attn_bwd_block_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      hipLaunchKernelGGL((seq_attn_bwd_block_kernel<1>), grid3, block3, lds3, s3, qkv, dout, dqkv, T, heads, causal, 0.125f);
    } else {
      (void)hipFuncSetAttribute((const void*)seq_attn_bwd_block_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      hipLaunchKernelGGL((seq_attn_bwd_block_kernel<2>), grid3, block3, lds3, s3, qkv, dout, dqkv, T, heads, causal, 0.125f);
    }
    ACX_CHECK_LAUNCH(ctx, "acx_seq_attention_bwd(block)");
    return ACX_OK;
  }
  if (e == 64 && gn == 1 && axis == 1 && T <= 256 && stats_ws) {
    // text-tower shape: wave-per-row kernel (two launches: rows as queries, then rows as keys)
    hipStream_t s2 = (hipStream_t)stream;
    AcxProfScope prof2__(ctx, ACX_K_ATTN, s2);
    const int64_t nrows = (int64_t)tiles * heads * T;
    const dim3 grid2((unsigned)((nrows + 3) / 4)), block2(256);
    const float scale2 = 0.125f;
    for (int pass = 0; pass < 2; ++pass) {
      if (T <= 64) hipLaunchKernelGGL((seq_attn_bwd_rows_kernel<64, 1>), grid2, block2, 0, s2, qkv, dout, dqkv, stats_ws, T, heads, causal, scale2, pass, nrows);
      else if (T <= 128) hipLaunchKernelGGL((seq_attn_bwd_rows_kernel<64, 2>), grid2, block2, 0, s2, qkv, dout, dqkv, stats_ws, T, heads, causal, scale2, pass, nrows);
      else hipLaunchKernelGGL((seq_attn_bwd_rows_kernel<64, 4>), grid2, block2, 0, s2, qkv, dout, dqkv, stats_ws, T, heads, causal, scale2, pass, nrows);
    }
    ACX_CHECK_LAUNCH(ctx, "acx_seq_attention_bwd(rows)");
    return ACX_OK;
  }
  if (!causal && (T == 16 || T == 32) && (e == 16 || e == 32) && ACX_DBG_SWITCH("AXB_MFMA", true)) {
    // axial attention of the temporal model: one wave per (line, head) group on the f32 MFMA
    const int64_t ngroups_m = (int64_t)tiles * (axis == 0 ? gl : gn) * heads;
    hipStream_t sm = (hipStream_t)stream;
    AcxProfScope profm__(ctx, ACX_K_ATTN, sm);
    const float scale_m = 1.f / sqrtf((float)e);
    const size_t lds_m = (size_t)4 * (4 * T * (e + 4) + 4 * T) * sizeof(float);
    int64_t nbm = (ngroups_m + 3) / 4;
    const int64_t capm = 2 * (int64_t)(ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256);
    if (nbm > capm) nbm = capm;
#define ACX_AXB(TT_, EE_)                                                                                    \
  do {                                                                                                       \
    (void)hipFuncSetAttribute((const void*)axial_attn_bwd_mfma_kernel<TT_, EE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m); \
    hipLaunchKernelGGL((axial_attn_bwd_mfma_kernel<TT_, EE_>), dim3((unsigned)nbm), dim3(256), lds_m, sm, qkv, dout, dqkv, gn, gl, heads, \
                       axis, scale_m, ngroups_m);                                                            \
  } while (0)
    if (T == 32) { if (e == 32) ACX_AXB(32, 32); else ACX_AXB(32, 16); }
    else { if (e == 32) ACX_AXB(16, 32); else ACX_AXB(16, 16); }
#undef ACX_AXB
    ACX_CHECK_LAUNCH(ctx, "acx_seq_attention_bwd(axial mfma)");
    return ACX_OK;
  }
  if (T <= 0 || T > 256 || (e != 16 && e != 32 && e != 64))
    return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_seq_attention_bwd: need sequence <= 256, head dim in {16,32,64}%s");
  const int64_t nlines = (int64_t)tiles * (axis == 0 ? gl : gn);
  const int64_t ngroups = nlines * heads;
  int gpb = 256 / T;
  while (gpb > 1 && (size_t)gpb * T * (2 * e + 3) * 4 > 96 * 1024) --gpb;
  const size_t lds = (size_t)gpb * T * (2 * e + 3) * 4;
  if (lds > 160 * 1024) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_seq_attention_bwd: sequence too long for LDS%s");
  const float scale = 1.f / sqrtf((float)e);
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_ATTN, s);
  const dim3 grid((unsigned)((ngroups + gpb - 1) / gpb)), block(256);
#define ACX_SAB(EE)                                                                                          \
  do {                                                                                                       \
    (void)hipFuncSetAttribute((const void*)seq_attn_bwd_kernel<EE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((seq_attn_bwd_kernel<EE>), grid, block, lds, s, qkv, dout, dqkv, tiles, gn, gl, heads, axis, causal, \
                       scale, T, gpb, ngroups);                                                              \
  } while (0)
  if (e == 16) ACX_SAB(16); else if (e == 32) ACX_SAB(32); else ACX_SAB(64);
#undef ACX_SAB
  ACX_CHECK_LAUNCH(ctx, "acx_seq_attention_bwd");
  return ACX_OK;
}

extern "C" int acx_pos_grad(acx_ctx* ctx, const float* dx, float* d0, float* d1, float* part, int32_t tiles, int32_t gn,
                            int32_t gl, int32_t E, void* stream) {
  if (!dx || !d0 || !d1 || !part) return acx_fail(ctx, ACX_E_BADARG, "acx_pos_grad: null pointer%s");
  if (tiles <= 0 || gn <= 0 || gl <= 0 || E <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_pos_grad: empty shape%s");
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  const int ng = (gn + 3) / 4;
  float* p0 = part;                                   // [tiles][gn*E]
  float* p1 = part + (size_t)tiles * gn * E;          // [tiles*ng][gl*E]
  hipLaunchKernelGGL(pos_grad_part_kernel, dim3((unsigned)(tiles * ng)), dim3(256), 0, s, dx, p0, p1, gn, gl, E);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((gn * E + 15) / 16)), dim3(256), 0, s, (const float*)p0, d0, tiles, gn * E);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((gl * E + 15) / 16)), dim3(256), 0, s, (const float*)p1, d1, tiles * ng,
                     gl * E);
  ACX_CHECK_LAUNCH(ctx, "acx_pos_grad");
  return ACX_OK;
}

extern "C" int acx_bn_bwd_apply(acx_ctx* ctx, const float* logits, const float* dlogits, const float* var_biased,
                                const float* sums, float* draw, int32_t ldo, int64_t rows, int64_t total_rows, int32_t C1,
                                float eps, const float* total_rows_dev, void* stream) {
  if (!logits || !dlogits || !var_biased || !sums || !draw) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_apply: null pointer%s");
  if (rows <= 0) return ACX_OK;
  if (!total_rows_dev && total_rows <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_apply: total_rows must be positive%s");
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  if (ldo < C1) return acx_fail(ctx, ACX_E_BADARG, "acx_bn_bwd_apply: ldo < C1%s");
  const int64_t total = rows * ldo;      // (pad columns included: the kernel zeroes them)
  hipLaunchKernelGGL(bn_bwd_apply_kernel, GRID1(total), dim3(256), 0, s, logits, dlogits, var_biased, sums, sums + C1, draw, ldo,
                     total, C1, eps, total_rows > 0 ? 1.f / (float)total_rows : 0.f, total_rows_dev);
  ACX_CHECK_LAUNCH(ctx, "acx_bn_bwd_apply");
  return ACX_OK;
}

extern "C" int acx_axpby(acx_ctx* ctx, const float* x, float* y, int32_t n, float a, float b, void* stream) {
  if (!x || !y) return acx_fail(ctx, ACX_E_BADARG, "acx_axpby: null pointer%s");
  if (n <= 0) return ACX_OK;
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(axpby_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, n, a, b);
  ACX_CHECK_LAUNCH(ctx, "acx_axpby");
  return ACX_OK;
}

extern "C" int acx_colsum_partials(acx_ctx* ctx, const float* x, int32_t ld, float* part, int64_t rows, int32_t D,
                                   int32_t rows_per_block, void* stream) {
  if (!x || !part) return acx_fail(ctx, ACX_E_BADARG, "acx_colsum_partials: null pointer%s");
  if (rows <= 0) return ACX_OK;
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const unsigned nb = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
  if (D % 4 == 0 && ld % 4 == 0 && !(((uintptr_t)x | (uintptr_t)part) & 15))
    hipLaunchKernelGGL(colsum_part4_kernel, dim3(nb, (unsigned)((D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, part, rows, D,
                       ld, rows_per_block);
  else
    hipLaunchKernelGGL(colsum_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, part, rows, D, ld, rows_per_block);
  ACX_CHECK_LAUNCH(ctx, "acx_colsum_partials");
  return ACX_OK;
}

extern "C" int acx_text_directions_bwd(acx_ctx* ctx, const float* text, const float* ncentroid, const float* ddirs, float* dtext,
                                       int32_t C, int32_t D, int32_t normal_id, void* stream) {
  if (!text || !ncentroid || !ddirs || !dtext) return acx_fail(ctx, ACX_E_BADARG, "acx_text_directions_bwd: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(text_dirs_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, text, ncentroid, ddirs, dtext, C, D, normal_id, 1,
                     (int64_t)0);
  ACX_CHECK_LAUNCH(ctx, "acx_text_directions_bwd");
  return ACX_OK;
}
extern "C" int acx_text_directions_bwd_parts(acx_ctx* ctx, const float* text, const float* ncentroid, const float* ddirs_parts,
                                             int32_t nparts, int64_t part_stride, float* dtext, int32_t C, int32_t D, int32_t normal_id,
                                             void* stream) {
  if (!text || !ncentroid || !ddirs_parts || !dtext) return acx_fail(ctx, ACX_E_BADARG, "acx_text_directions_bwd_parts: null pointer%s");
  if (nparts < 1 || (nparts > 1 && part_stride < (int64_t)(C - 1) * D))
    return acx_fail(ctx, ACX_E_BADARG, "acx_text_directions_bwd_parts: nparts >= 1, part_stride >= (C - 1) * D%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  hipLaunchKernelGGL(text_dirs_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, text, ncentroid, ddirs_parts, dtext, C, D, normal_id,
                     nparts, part_stride);
  ACX_CHECK_LAUNCH(ctx, "acx_text_directions_bwd_parts");
  return ACX_OK;
}

extern "C" int acx_select_idx(acx_ctx* ctx, const float* logits, const int64_t* labels, const float* mask_top,
                              const float* mask_bot, int64_t* idx_top, int64_t* idx_bot, int32_t B, int32_t N, int32_t Lg,
                              int32_t C1, int32_t normal_id, int32_t ktop, int32_t kbot, void* stream) {
  if (!logits || !labels || !mask_top || !mask_bot || !idx_top || !idx_bot) return acx_fail(ctx, ACX_E_BADARG, "acx_select_idx: null pointer%s");
  if (B <= 0 || B % 2 || ktop > N || kbot > N) return acx_fail(ctx, ACX_E_BADARG, "acx_select_idx: need even B and k <= N%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const size_t lds = (size_t)(N * C1 + 2 * N) * 4;
  hipLaunchKernelGGL(select_idx_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logits, labels, mask_top, mask_bot, idx_top,
                     idx_bot, B, N, Lg, C1, normal_id, ktop, kbot);
  ACX_CHECK_LAUNCH(ctx, "acx_select_idx");
  return ACX_OK;
}

extern "C" int acx_gather_segments(acx_ctx* ctx, const float* logits, const int64_t* idx, float* out, int32_t B, int32_t N,
                                   int32_t Lg, int32_t C1, int32_t K, void* stream) {
  if (!logits || !idx || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_gather_segments: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const int64_t total = (int64_t)B * K * Lg * C1;
  hipLaunchKernelGGL(gather_segments_kernel, GRID1(total), dim3(256), 0, (hipStream_t)stream, logits, idx, out, total, N, Lg, C1, K);
  ACX_CHECK_LAUNCH(ctx, "acx_gather_segments");
  return ACX_OK;
}

extern "C" int acx_scatter_segments(acx_ctx* ctx, const float* dout, const int64_t* idx, float* dlogits, int32_t B, int32_t N,
                                    int32_t Lg, int32_t C1, int32_t K, void* stream) {
  if (!dout || !idx || !dlogits) return acx_fail(ctx, ACX_E_BADARG, "acx_scatter_segments: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const int64_t total = (int64_t)B * K * Lg * C1;
  hipLaunchKernelGGL(scatter_segments_kernel, GRID1(total), dim3(256), 0, (hipStream_t)stream, dout, idx, dlogits, total, N, Lg, C1, K);
  ACX_CHECK_LAUNCH(ctx, "acx_scatter_segments");
  return ACX_OK;
}

extern "C" int acx_mil_loss(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                            const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn,
                            float* dsim, float* dsim_topk, float* dscores, float* losses, float* workspace,
                            size_t workspace_floats, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t K, int32_t normal_id,
                            const float* lambdas /* [7] */, const float* gout, void* stream) {
  return acx_mil_loss_one(ctx, sim, sim_topk, labels, scores, idx_topk_abn, idx_topk_nor, idx_bottomk_abn, dsim, dsim_topk, dscores,
                          losses, workspace, workspace_floats, B, N, Lg, C1, K, normal_id, lambdas, gout, nullptr, stream);
}

extern "C" int acx_mil_loss_one(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                                const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn,
                                float* dsim, float* dsim_topk, float* dscores, float* losses, float* workspace,
                                size_t workspace_floats, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t K, int32_t normal_id,
                                const float* lambdas /* [7] */, const float* gout, uint32_t* counter, void* stream) {
  if (!sim || !sim_topk || !labels || !scores || !idx_topk_abn || !idx_topk_nor || !idx_bottomk_abn || !dsim || !dsim_topk ||
      !dscores || !losses || !workspace || !lambdas)
    return acx_fail(ctx, ACX_E_BADARG, "acx_mil_loss: null pointer%s");
  if (B <= 0 || B % 2 || C1 > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_mil_loss: need even B, C-1 <= 64%s");
  const int64_t R = (int64_t)B * N * Lg, RT = (int64_t)B * K * Lg;
  const int np1 = (int)((R + 255) / 256), np2 = (int)((RT + 255) / 256);
  if ((size_t)np1 * 8 + np2 > workspace_floats) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_mil_loss: workspace too small%s");
  LossArgs a;
  a.sim = sim; a.sim_topk = sim_topk; a.labels = labels; a.scores = scores;
  a.idx_topk_abn = idx_topk_abn; a.idx_topk_nor = idx_topk_nor; a.idx_bottomk_abn = idx_bottomk_abn;
  a.dsim = dsim; a.dsim_topk = dsim_topk; a.dscores = dscores; a.part = workspace;
  a.B = B; a.N = N; a.Lg = Lg; a.C1 = C1; a.K = K; a.normal_id = normal_id;
  a.l_dir_abn = lambdas[0]; a.l_dir_nor = lambdas[1]; a.l_topk_abn = lambdas[2]; a.l_bottomk_abn = lambdas[3];
  a.l_topk_nor = lambdas[4]; a.l_smooth = lambdas[5]; a.l_sparse = lambdas[6];
  a.gout_ptr = gout;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  float* part2 = workspace + (size_t)np1 * 8;
  if (counter) {
    hipLaunchKernelGGL(loss_fused_kernel, dim3(np1 + np2), dim3(256), 0, s, a, part2, np1, np2, losses, counter);
    ACX_CHECK_LAUNCH(ctx, "acx_mil_loss");
    return ACX_OK;
  }
  hipLaunchKernelGGL(loss_rows_kernel, dim3(np1), dim3(256), 0, s, a);
  hipLaunchKernelGGL(loss_topk_kernel, dim3(np2), dim3(256), 0, s, a, part2);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, a, (const float*)workspace, np1, (const float*)part2, np2, losses);
  ACX_CHECK_LAUNCH(ctx, "acx_mil_loss");
  return ACX_OK;
}

extern "C" int acx_mil_loss_bn(acx_ctx* ctx, const float* sim, const float* sim_topk, const int64_t* labels, const float* scores,
                               const int64_t* idx_topk_abn, const int64_t* idx_topk_nor, const int64_t* idx_bottomk_abn,
                               float* dlogits, float* dscores, float* losses, float* meter, float* bn_sums, float* workspace,
                               size_t workspace_floats, void* bn_workspace, size_t bn_workspace_bytes, int32_t B, int32_t N, int32_t Lg,
                               int32_t C1, int32_t K, int32_t normal_id, const float* lambdas /* [7] */, const float* gout,
                               uint32_t* counter, void* stream) {
  if (!sim || !sim_topk || !labels || !scores || !idx_topk_abn || !idx_topk_nor || !idx_bottomk_abn || !dlogits || !dscores ||
      !losses || !bn_sums || !workspace || !bn_workspace || !lambdas || !counter)
    return acx_fail(ctx, ACX_E_BADARG, "acx_mil_loss_bn: null pointer%s");
  if (B <= 0 || B % 2 || C1 <= 0 || C1 > 64) return acx_fail(ctx, ACX_E_BADARG, "acx_mil_loss_bn: need even B, 1 <= C-1 <= 64%s");
  const int64_t R = (int64_t)B * N * Lg, RT = (int64_t)B * K * Lg;
  // the block of 256 frame rows must BE the slab of acx_bn_bwd_stats' stage 1 (bn_blocks: ceil(rows / 256) blocks, at most 512)
  if (R % 256 || R / 256 > 512) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_mil_loss_bn: B N Lg must be a multiple of 256, at most 131072 rows%s");
  const int np1 = (int)(R / 256), np2 = (int)((RT + 255) / 256);
  if ((size_t)np1 * 8 + np2 > workspace_floats) return acx_fail(ctx, ACX_E_WORKSPACE, "acx_mil_loss_bn: workspace too small%s");
  if (bn_workspace_bytes < (size_t)np1 * 2 * 64 * sizeof(double) || ((uintptr_t)bn_workspace & 7))
    return acx_fail(ctx, ACX_E_WORKSPACE, "acx_mil_loss_bn: BatchNorm workspace too small (acx_bn_workspace_bytes) or misaligned%s");
  LossArgs a;
  a.sim = sim; a.sim_topk = sim_topk; a.labels = labels; a.scores = scores;
  a.idx_topk_abn = idx_topk_abn; a.idx_topk_nor = idx_topk_nor; a.idx_bottomk_abn = idx_bottomk_abn;
  a.dsim = dlogits; a.dsim_topk = nullptr; a.dscores = dscores; a.part = workspace;
  a.B = B; a.N = N; a.Lg = Lg; a.C1 = C1; a.K = K; a.normal_id = normal_id;
  a.l_dir_abn = lambdas[0]; a.l_dir_nor = lambdas[1]; a.l_topk_abn = lambdas[2]; a.l_bottomk_abn = lambdas[3];
  a.l_topk_nor = lambdas[4]; a.l_smooth = lambdas[5]; a.l_sparse = lambdas[6];
  a.gout_ptr = gout;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  hipLaunchKernelGGL(loss_bn_kernel, dim3(np1 + np2), dim3(256), (size_t)256 * C1 * sizeof(float), s, a, workspace + (size_t)np1 * 8, np1,
                     np2, losses, meter, (double*)bn_workspace, bn_sums, counter);
  ACX_CHECK_LAUNCH(ctx, "acx_mil_loss_bn");
  return ACX_OK;
}

extern "C" int acx_selector_tail(acx_ctx* ctx, const float* raw, const float* gathered, int32_t ranks, const float* mean,
                                 const float* var_biased, const float* var_unbiased, float* stat_out, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, float momentum, float one_minus, float* logits,
                                 int64_t ldl, const int64_t* labels, const float* mask_top, const float* mask_bot, int64_t* idx_top, int64_t* idx_bot,
                                 float* logits_topk, int32_t B, int32_t N, int32_t Lg, int32_t C1, int32_t normal_id, int32_t ktop,
                                 int32_t kbot, float eps, void* stream) {
  if (!raw || !logits || !labels || !mask_top || !mask_bot || !idx_top || !idx_bot || !logits_topk)
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_tail: null pointer%s");
  if (gathered ? (!stat_out || ranks < 1) : (!mean || !var_biased || !var_unbiased))
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_tail: gathered statistics need stat_out and ranks >= 1, local ones mean / var%s");
  if ((running_mean == nullptr) != (running_var == nullptr)) return acx_fail(ctx, ACX_E_BADARG, "acx_selector_tail: running_mean and running_var come together%s");
  if (B <= 0 || B % 2 || ktop > N || kbot > N || ktop < 0 || kbot < 0 || C1 <= 0 || C1 > 64 || ldl < C1)
    return acx_fail(ctx, ACX_E_BADARG, "acx_selector_tail: need even B, k <= N, 1 <= C-1 <= 64, ldl >= C-1%s");
  const size_t lds = ((size_t)N * Lg * C1 + (size_t)N * C1 + 2 * (size_t)N + 2 * (size_t)C1) * 4 + (size_t)(ktop + kbot) * 4;
  if (lds > 160 * 1024) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_selector_tail: a video's logits exceed the 160 KB of LDS%s");
  SelTailArgs a;
  a.raw = raw; a.gathered = gathered; a.R = ranks; a.mean_in = mean; a.var_b_in = var_biased; a.var_u_in = var_unbiased;
  a.stat_out = stat_out; a.rm = running_mean; a.rv = running_var; a.nbt = (long long*)num_batches_tracked;
  a.momentum = momentum; a.om = one_minus;
  a.logits = logits; a.ldl = ldl; a.labels = labels; a.mask_top = mask_top; a.mask_bot = mask_bot;
  a.idx_top = idx_top; a.idx_bot = idx_bot; a.logits_topk = logits_topk;
  a.B = B; a.N = N; a.Lg = Lg; a.C1 = C1; a.normal_id = normal_id; a.ktop = ktop; a.kbot = kbot; a.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  AcxProfScope prof__(ctx, ACX_K_OTHER, s);
  const int dev_slot = (ctx ? ctx->device : 0) & 63;
  static size_t attr_dev_[64] = {};
  if (lds > 64 * 1024 && attr_dev_[dev_slot] < lds) {
    (void)hipFuncSetAttribute((const void*)selector_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_dev_[dev_slot] = lds;
  }
  hipLaunchKernelGGL(selector_tail_kernel, dim3(B), dim3(256), lds, s, a);
  ACX_CHECK_LAUNCH(ctx, "acx_selector_tail");
  return ACX_OK;
}

extern "C" int acx_adamw(acx_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  if (!p || !g || !m || !v) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (step <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw: step starts at 1%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, GRID1(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2s);
  ACX_CHECK_LAUNCH(ctx, "acx_adamw");
  return ACX_OK;
}

extern "C" int acx_multi_axpy(acx_ctx* ctx, int32_t nseg, void* const* y, const void* const* x, const int64_t* n, float a,
                              void* stream) {
  if (nseg <= 0) return ACX_OK;
  if (!y || !x || !n) return acx_fail(ctx, ACX_E_BADARG, "acx_multi_axpy: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nseg) {
    AxpySegs t;
    memset(&t, 0, sizeof(t));
    long long chunks = 0;
    int k = 0;
    for (; i < nseg && k < AXPY_MAX_SEG; ++i) {
      if (n[i] <= 0) continue;
      if (!y[i] || !x[i]) return acx_fail(ctx, ACX_E_BADARG, "acx_multi_axpy: null tensor pointer%s");
      t.y[k] = (float*)y[i]; t.x[k] = (const float*)x[i]; t.n[k] = n[i];
      t.chunk0[k] = (int)chunks;
      chunks += (n[i] + 1023) / 1024;
      ++k;
    }
    if (k == 0) continue;
    if (chunks > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_multi_axpy: too many elements for one launch%s");
    t.chunk0[k] = (int)chunks;
    t.nseg = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(multi_axpy_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, a);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_multi_axpy");
  return ACX_OK;
}

extern "C" int acx_adamw_multi(acx_ctx* ctx, int32_t nseg, void* const* p, const void* const* g, void* const* m, void* const* v,
                               const int64_t* n, const double* lr, const double* weight_decay, double beta1, double beta2, double eps,
                               int32_t step, void* stream) {
  if (nseg <= 0) return ACX_OK;
  if (!p || !g || !m || !v || !n || !lr || !weight_decay) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw_multi: null pointer%s");
  if (step <= 0) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw_multi: step starts at 1%s");
  // scalar terms in f64 on the host and rounded ONCE, exactly as torch.optim.AdamW forms them from its Python floats
  // (1 - beta2 = 0.001 there; 1.f - 0.999f = 0.00100005 would put exp_avg_sq 5e-5 off)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const float bc2 = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  while (i < nseg) {
    AdamwSegs t;
    memset(&t, 0, sizeof(t));
    long long chunks = 0;
    int k = 0;
    for (; i < nseg && k < ADAMW_MAX_SEG; ++i) {
      if (n[i] <= 0) continue;
      if (!p[i] || !g[i] || !m[i] || !v[i]) return acx_fail(ctx, ACX_E_BADARG, "acx_adamw_multi: null tensor pointer%s");
      t.p[k] = (float*)p[i]; t.g[k] = (const float*)g[i]; t.m[k] = (float*)m[i]; t.v[k] = (float*)v[i];
      t.n[k] = n[i]; t.decay[k] = (float)(1.0 - lr[i] * weight_decay[i]); t.step_size[k] = (float)(lr[i] / bc1);
      t.chunk0[k] = (int)chunks;
      chunks += (n[i] + 1023) / 1024;
      ++k;
    }
    if (k == 0) continue;
    if (chunks > 0x7fffffffLL) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_adamw_multi: too many elements for one launch%s");
    t.chunk0[k] = (int)chunks;
    t.nseg = k;
    AcxProfScope prof__(ctx, ACX_K_OTHER, s);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, bc2);
  }
  ACX_CHECK_LAUNCH(ctx, "acx_adamw_multi");
  return ACX_OK;
}

extern "C" int acx_ctx_grad(acx_ctx* ctx, const float* dx, float* dctx, int32_t C, int32_t n_ctx, int32_t Lc, int32_t W,
                            int32_t shared_ctx, void* stream) {
  if (!dx || !dctx) return acx_fail(ctx, ACX_E_BADARG, "acx_ctx_grad: null pointer%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const int64_t total = (int64_t)(shared_ctx ? 1 : C) * n_ctx * W;
  hipLaunchKernelGGL(ctx_grad_kernel, GRID1(total), dim3(256), 0, (hipStream_t)stream, dx, dctx, C, n_ctx, Lc, W, shared_ctx);
  ACX_CHECK_LAUNCH(ctx, "acx_ctx_grad");
  return ACX_OK;
}

extern "C" int acx_scatter_rows(acx_ctx* ctx, const float* src, const int64_t* idx, float* out, int64_t n, int32_t W, void* stream) {
  if (!src || !idx || !out) return acx_fail(ctx, ACX_E_BADARG, "acx_scatter_rows: null pointer%s");
  if (n <= 0) return ACX_OK;
  if (W % 4) return acx_fail(ctx, ACX_E_BADARG, "acx_scatter_rows: W%%4%s");
  AcxProfScope prof__(ctx, ACX_K_OTHER, (hipStream_t)stream);
  const int64_t total4 = n * W / 4;
  hipLaunchKernelGGL(scatter_rows_kernel, GRID1(total4), dim3(256), 0, (hipStream_t)stream, src, idx, out, total4, W);
  ACX_CHECK_LAUNCH(ctx, "acx_scatter_rows");
  return ACX_OK;
}
