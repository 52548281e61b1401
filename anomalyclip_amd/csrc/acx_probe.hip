// Measured denominators for the roofline report ("the builder must measure a GEMM peak and a
// stream-copy peak on the box and report both nominal and measured denominators").  Two micro-kernels, no product path
// uses them:
//   acx_probe_mfma  -- register-only MFMA loop (v_mfma_f32_32x32x2_f32 or v_mfma_f32_32x32x16_bf16), four independent
//                      accumulator chains per wave, `waves_per_simd` waves on every SIMD of the chip: the issue-rate
//                      ceiling of the matrix pipe at the clock the chip sustains under that load (bf16 = 1: constant operands;
//                      bf16 = 2: random operands -- the rate the power limit leaves with operands that toggle like data);
//   acx_probe_copy  -- 16-byte-per-lane grid-stride copy (global_load_dwordx4 / global_store_dwordx4): the HBM stream
//                      ceiling (read + write) the row kernels are measured against.
#include "acx_internal.h"
#include <algorithm>

namespace {

template <int BF16>
__global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
  const float x = 1.0f + (float)(threadIdx.x & 7) * 0.125f, y = 0.5f;
  if constexpr (BF16 == 2) {
    // RANDOM operands: four different (A, B) fragment pairs per lane, bf16 values with random sign / mantissa and exponents in
    // [2^-3, 2^1) -- the matrix pipe's sustained rate at the chip's power limit with operands that toggle like real data (the
    // constant-operand loop above measures the issue rate at a clock real operands do not sustain)
    bf16x8 ra[4], rb[4];
    unsigned st = (unsigned)(blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef unsigned short u16x8_ __attribute__((ext_vector_type(8)));
      u16x8_ va, vb;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st = st * 1664525u + 1013904223u;
        va[e] = (unsigned short)(((st >> 16) & 0x807fu) | ((124u + ((st >> 9) & 3u)) << 7));
        st = st * 1664525u + 1013904223u;
        vb[e] = (unsigned short)(((st >> 16) & 0x807fu) | ((124u + ((st >> 9) & 3u)) << 7));
      }
      ra[q] = __builtin_bit_cast(bf16x8, va); rb[q] = __builtin_bit_cast(bf16x8, vb);
    }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[0], rb[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[1], rb[1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[2], rb[2], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[3], rb[3], a3, 0, 0, 0);
    }
  } else if constexpr (BF16 == 1) {
    bf16x8 xa, xb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)x; xb[e] = (__bf16)y; }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, a3, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  if (s == 12345.678f) sink[0] = s;              // keeps the chains alive; never true
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i = (int64_t)blockIdx.x * 2048 + threadIdx.x; i < n16; i += stride) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i + 256 * k < n16) v[k] = __builtin_nontemporal_load(src + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i + 256 * k < n16) __builtin_nontemporal_store(v[k], dst + i + 256 * k);
  }
}

// read-only stream: every lane folds its 16-byte pieces into one value, nothing is written (the sink store never happens) --
// the floor of a ONE-SHOT launch that reads `bytes` once (ramp-up and tail included): what the skinny reductions of the head
// (selector projection, column sums: 67 MB in, a few KB out) are measured against
__global__ __launch_bounds__(256) void probe_read_kernel(const f32x4* __restrict__ src, int64_t n16, float* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 2048 + threadIdx.x; i < n16; i += stride) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = i + 256 * k < n16 ? src[i + 256 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) a += v[k];
  }
  if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[0] = a[0];   // never true for the probe's inputs
}

}  // namespace

extern "C" int acx_probe_read(acx_ctx* ctx, const void* src, int64_t bytes, float* sink, void* stream) {
  if (!src || !sink || bytes <= 0 || (bytes & 15) || ((uintptr_t)src & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_probe_read: need a 16-byte aligned buffer and size%s");
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  const int64_t n16 = bytes / 16;
  const int64_t want = (n16 + 2047) / 2048;
  hipLaunchKernelGGL(probe_read_kernel, dim3((unsigned)std::min<int64_t>(want, (int64_t)ncu * 16)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)src, n16, sink);
  ACX_CHECK_LAUNCH(ctx, "acx_probe_read");
  return ACX_OK;
}

extern "C" int acx_probe_mfma(acx_ctx* ctx, int32_t bf16, int32_t iters, int32_t waves_per_simd, float* sink, double* flops_out,
                              void* stream) {
  if (!sink || iters <= 0 || waves_per_simd <= 0 || waves_per_simd > 8)
    return acx_fail(ctx, ACX_E_BADARG, "acx_probe_mfma: bad argument%s");
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  const dim3 grid((unsigned)(ncu * waves_per_simd)), block(256);           // 4 waves per block = one per SIMD
  hipStream_t s = (hipStream_t)stream;
  if (bf16 == 2) hipLaunchKernelGGL((probe_mfma_kernel<2>), grid, block, 0, s, iters, sink);
  else if (bf16) hipLaunchKernelGGL((probe_mfma_kernel<1>), grid, block, 0, s, iters, sink);
  else hipLaunchKernelGGL((probe_mfma_kernel<0>), grid, block, 0, s, iters, sink);
  if (flops_out) *flops_out = (double)grid.x * 4.0 * (double)iters * 4.0 * (bf16 ? 32768.0 : 4096.0);
  ACX_CHECK_LAUNCH(ctx, "acx_probe_mfma");
  return ACX_OK;
}

extern "C" int acx_probe_copy(acx_ctx* ctx, const void* src, void* dst, int64_t bytes, void* stream) {
  if (!src || !dst || bytes <= 0 || (bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15))
    return acx_fail(ctx, ACX_E_BADARG, "acx_probe_copy: need 16-byte aligned buffers and size%s");
  const int ncu = ctx && ctx->multiprocessors > 0 ? ctx->multiprocessors : 256;
  hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)(ncu * 16)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src,
                     (f32x4*)dst, bytes / 16);
  ACX_CHECK_LAUNCH(ctx, "acx_probe_copy");
  return ACX_OK;
}
