// Where do the 8 waves of a 512-thread workgroup land?  (attn16_kernel geometry: 512 threads, 32 KB LDS, 128 VGPRs -> two
// workgroups per CU.)  Prints wave -> SIMD_ID for the first workgroups and the histogram of per-workgroup patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
__global__ __launch_bounds__(512, 4) void probe(unsigned* out) {
  extern __shared__ char smem[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  smem[threadIdx.x] = 1;
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 3000) { }
}
int main() {
  const int nb = 512;
  unsigned* d;
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 32768, 0, d);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
  std::map<std::string, int> pat;
  for (int b = 0; b < nb; ++b) {
    std::string s;
    for (int w = 0; w < 8; ++w) s += char('0' + ((h[b * 8 + w] >> 4) & 3));
    pat[s]++;
    if (b < 8 || (b >= 256 && b < 264)) {
      printf("block %3d cu %2u se %u: simd of waves 0..7 = %s   wave slots:", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7, s.c_str());
      for (int w = 0; w < 8; ++w) printf(" %u", h[b * 8 + w] & 15);
      printf("\n");
    }
  }
  for (auto& kv : pat) printf("pattern %s : %d workgroups\n", kv.first.c_str(), kv.second);
  return 0;
}
