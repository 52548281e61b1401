// Does vmcnt retire in ISSUE order across loads and stores?  A cold (HBM) load is followed by 4 full-line stores to L2-hot lines and
// `s_waitcnt vmcnt(4)`; the load's destination register is sampled right after that wait and again after vmcnt(0).  If stores
// could retire ahead of the older load, the first sample would still hold the sentinel in some trials.
// Second experiment (timing): the same with the wait replaced by vmcnt(0): how long the stores add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, unsigned* __restrict__ res,
                                             int iters, size_t cold_words) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  unsigned bad = 0, seen = 0;
  unsigned* h = hot + (tid >> 6) * 2048 + (tid & 63) * 4;   // 1 KB per wave store instruction, 8 KB per wave: full lines
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (int it = 0; it < iters; ++it) {
    const u4 val4 = {(unsigned)it, 1u, 2u, 3u};
    const unsigned* p = cold + ((tid * 16 + (size_t)it * nthreads * 16) % cold_words);   // a fresh 64-B line per lane and iteration
    unsigned v, early, late;
    asm volatile(
        "v_mov_b32 %0, 0xdeadbeef\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "global_load_dword %0, %3, off\n\t"
        "global_store_dwordx4 %4, %5, off\n\t"
        "global_store_dwordx4 %4, %5, off offset:1024\n\t"
        "global_store_dwordx4 %4, %5, off offset:2048\n\t"
        "global_store_dwordx4 %4, %5, off offset:3072\n\t"



        "s_waitcnt vmcnt(4)\n\t"
        "v_mov_b32 %1, %0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_mov_b32 %2, %0\n\t"
        : "=&v"(v), "=&v"(early), "=&v"(late)
        : "v"(p), "v"(h), "v"(val4)
        : "memory");
    bad += (early != late);
    seen += (late != 0xdeadbeefu);
  }
  res[tid * 2] = bad; res[tid * 2 + 1] = seen;
}
// timing (s_memtime, 100 MHz ticks are too coarse: use s_memrealtime? -> clock64 = shader clock counter): latency of the cold
// load alone, of the 4 hot 1-KB stores alone, and of load + stores up to vmcnt(4) / vmcnt(0)
__global__ __launch_bounds__(64) void timing(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, long long* __restrict__ out,
                                            int iters, size_t cold_words) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  unsigned* h = hot + (tid >> 6) * 2048 + (tid & 63) * 4;
  long long tl = 0, ts = 0, t8 = 0, t0 = 0;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (int it = 0; it < iters; ++it) {
    const u4 val4 = {(unsigned)it, 1u, 2u, 3u};
    const unsigned* p = cold + ((tid * 16 + (size_t)(3 * it) * nthreads * 16) % cold_words);
    const unsigned* p2 = cold + ((tid * 16 + (size_t)(3 * it + 1) * nthreads * 16) % cold_words);
    unsigned v;
    long long a, b, c;
    // cold load alone
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\tglobal_load_dword %0, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v), "=&s"(a), "=&s"(b) : "v"(p) : "memory");
    tl += b - a;
    // 4 hot 1-KB stores alone
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "global_store_dwordx4 %2, %3, off\n\tglobal_store_dwordx4 %2, %3, off offset:1024\n\tglobal_store_dwordx4 %2, %3, off offset:2048\n\t"
                 "global_store_dwordx4 %2, %3, off offset:3072\n\t"
                 ""
                 "s_waitcnt vmcnt(0)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "v"(h), "v"(val4) : "memory");
    ts += b - a;
    // cold load, then the stores: time to vmcnt(4) and to vmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\tglobal_load_dword %0, %4, off\n\t"
                 "global_store_dwordx4 %5, %6, off\n\tglobal_store_dwordx4 %5, %6, off offset:1024\n\tglobal_store_dwordx4 %5, %6, off offset:2048\n\t"
                 "global_store_dwordx4 %5, %6, off offset:3072\n\t"
                 ""
                 "s_waitcnt vmcnt(4)\n\ts_memtime %2\n\ts_waitcnt vmcnt(0)\n\ts_memtime %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v), "=&s"(a), "=&s"(b), "=&s"(c) : "v"(p2), "v"(h), "v"(val4) : "memory");
    t8 += b - a; t0 += c - a;
  }
  if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 4] = tl; out[blockIdx.x * 4 + 1] = ts; out[blockIdx.x * 4 + 2] = t8; out[blockIdx.x * 4 + 3] = t0; }
}
int main() {
  const size_t cold_words = (size_t)1 << 30;       // 4 GB of cold data
  unsigned *cold, *hot, *res;
  const int blocks = 256, threads = 256, iters = 2000;
  if (hipMalloc(&cold, cold_words * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(cold, 0x5a, cold_words * 4);
  (void)hipMalloc(&hot, (size_t)blocks * threads * 32 * 4);
  (void)hipMalloc(&res, (size_t)blocks * threads * 2 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, cold, hot, res, iters, cold_words);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)blocks * threads * 2);
    (void)hipMemcpy(h.data(), res, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long bad = 0, seen = 0;
    for (size_t i = 0; i < h.size(); i += 2) { bad += h[i]; seen += h[i + 1]; }
    printf("rep %d: %llu trials, load value seen after vmcnt(0) in %llu, sampled EARLY (stores retired before the older load) in %llu\n",
           rep, (unsigned long long)blocks * threads * iters, seen, bad);
  }
  {
    const int tb = 256, ti = 200;
    long long* tout;
    (void)hipMalloc(&tout, tb * 4 * 8);
    hipLaunchKernelGGL(timing, dim3(tb), dim3(64), 0, 0, cold, hot, tout, ti, cold_words);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(tb * 4);
    (void)hipMemcpy(h.data(), tout, h.size() * 8, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < tb; ++b) for (int k = 0; k < 4; ++k) s[k] += (double)h[b * 4 + k];
    printf("s_memtime ticks (100 MHz) per trial, one wave per workgroup: cold load alone %.1f, 4 hot 1-KB stores alone %.1f, load+stores to vmcnt(4) %.1f, to vmcnt(0) %.1f\n",
           s[0] / tb / ti, s[1] / tb / ti, s[2] / tb / ti, s[3] / tb / ti);
  }
  return 0;
}
