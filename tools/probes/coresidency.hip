// Which blocks of a 1-D grid share a CU?  (placement probe for the GEMM's tile order: 512-thread blocks, 72 KB LDS ->
// two per CU).  Prints, for the first blocks, (xcc, se, cu) and the block ids that started on the same CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>
__global__ __launch_bounds__(512) void probe(unsigned* out, long long* t) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    t[blockIdx.x] = wall_clock64();
    smem[0] = 1;
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 3000) { }   // ~30 us at 100 MHz
}
int main() {
  const int nb = 4728;
  unsigned* d; long long* dt;
  hipMalloc(&d, nb * 8); hipMalloc(&dt, nb * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 73728, 0, d, dt);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * nb); std::vector<long long> ht(nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, nb * 8, hipMemcpyDeviceToHost);
  std::map<std::tuple<unsigned, unsigned, unsigned>, std::vector<int>> cu;
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[{xcc, se * 2 + sh, cuid}].push_back(b);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("xcc %u se/sh %u cu %u :", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first));
    for (size_t i = 0; i < kv.second.size() && i < 8; ++i) printf(" %d(t=%lld)", kv.second[i], (ht[kv.second[i]] - ht[0]) / 100);
    printf("\n");
  }
  // histogram of (second - first) block id difference per CU
  std::map<int, int> diff;
  for (auto& kv : cu) if (kv.second.size() >= 2) diff[kv.second[1] - kv.second[0]]++;
  for (auto& kv : diff) printf("first-two-blocks id difference %d : %d CUs\n", kv.first, kv.second);
  return 0;
}
