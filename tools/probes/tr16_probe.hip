// ds_read_b64_tr_b16 semantics probe (gfx950): the LDS image of a TN operand unit of acx_gemm_x6.h -- [32 m][256 channels] bf16,
// 512 B per m row, 64-byte channel chunks XOR-swizzled by (m & 7) -- read back as v_mfma_f32_32x32x16_bf16 fragments
// (lane -> channel blk * 32 + (lane & 31), k = 8 (lane >> 5) .. + 7 of substep s) by two transpose reads per fragment.
// Prints the number of mismatching elements (0 = the address map of the kernel is right).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(1024))) unsigned short img[32 * 256];
  for (int e = threadIdx.x; e < 32 * 256; e += blockDim.x) {
    const int m = e / 256, c = e % 256;
    const int pos = (((c >> 5) ^ (m & 7)) << 5) | (c & 31);      // 64-byte chunk (32 channels) swizzled by m & 7
    img[m * 256 + pos] = (unsigned short)(m * 256 + c);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 15, hh = lane >> 5;
  for (int blk = 0; blk < 8; ++blk)
    for (int s = 0; s < 2; ++s) {
      unsigned short v[8];
      for (int r = 0; r < 2; ++r) {
        const int m = s * 16 + 8 * hh + 4 * r + (i >> 2);
        const int c = blk * 32 + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
        const int off = m * 512 + (((c >> 5) ^ (m & 7)) * 64) + (c & 31) * 2;
        s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((char*)img + off));
        for (int j = 0; j < 4; ++j) v[4 * r + j] = (unsigned short)t[j];
      }
      for (int j = 0; j < 8; ++j) out[((blk * 2 + s) * 64 + lane) * 8 + j] = v[j];
    }
}
int main() {
  unsigned short* d; hipMalloc(&d, 8 * 2 * 64 * 8 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[8 * 2 * 64 * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 8; ++blk) for (int s = 0; s < 2; ++s) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
    const int m = s * 16 + 8 * (lane >> 5) + j, c = blk * 32 + (lane & 31);
    const int got = h[((blk * 2 + s) * 64 + lane) * 8 + j];
    if (got != m * 256 + c) { if (bad < 8) printf("blk %d s %d lane %d j %d: got (m %d, c %d) want (m %d, c %d)\n", blk, s, lane, j, got / 256, got % 256, m, c); ++bad; }
  }
  printf("tr16 probe: %d mismatches\n", bad);
  return bad != 0;
}
