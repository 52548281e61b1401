// Internal helpers shared by the libacx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/acx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

// in-library launch timer (bench.py's roofline leg): HIP events recorded on the caller's stream
// around every kernel launch of a kind, summed by acx_prof_collect.
enum { ACX_K_GEMM = 0, ACX_K_ATTN = 1, ACX_K_NORM = 2, ACX_K_OTHER = 3, ACX_K_COUNT = 4,
       ACX_K_GEMM_TN = 4 };   // acx_gemm_tn launches: reported under ACX_K_GEMM by acx_prof_collect, alone by acx_prof_gemm_tn
constexpr int ACX_PROF_MAX = 32768;

// Cross-workgroup hand-offs of the last-arriver reductions (colsum_fused, loss_fused, the K pieces of gemm_f32_sk_kernel and
// gemm_f32_w8_kernel): partial results are published by agent-scope write-through (sc1) stores, `s_waitcnt vmcnt(0)` (inline asm with
// a "memory" clobber), then a relaxed agent-scope arrival counter; the last arriver reads them back with agent-scope (sc1) loads.
// That is the guide's "handoff-flag / publish-large" form (MI355X_MICROARCH.md: valid forms; 3.0 us against 8.2 us per 64 KB
// publish) and it is what the product builds.  -DACX_HANDOFF_FENCES=1 ADDITIONALLY states the ordering in the HIP memory model: a
// release fence at agent scope in the arriving lane (buffer_wbl2 sc1: writes back the XCD L2's dirty lines) + the guide's asm wait
// before the counter, an acquire fence (buffer_inv sc1) in the last arriver.  A/B of the two builds on the training step:
// profiles/r06_handoff_fences.txt (the fenced build is correct and SLOWER: the write-back covers everything the preceding
// kernels of the stream left dirty in that L2, not just the few KB published here).
#ifndef ACX_HANDOFF_FENCES
#define ACX_HANDOFF_FENCES 0
#endif
#if ACX_HANDOFF_FENCES
#define ACX_HANDOFF_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define ACX_HANDOFF_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define ACX_HANDOFF_RELEASE() do { } while (0)
#define ACX_HANDOFF_ACQUIRE() do { } while (0)
#endif

// Development A/B switches.  The product library compiles them to their constant defaults; only a tools build with
// -DACX_DEBUG_SWITCHES (tools/ab_gemm.sh) reads them from the environment (ACX_<NAME>=0/1), once per process.
#ifdef ACX_DEBUG_SWITCHES
#include <stdlib.h>
static inline bool acx_dbg_switch_env(const char* env, bool dflt) {
  const char* v = getenv(env);
  return v ? atoi(v) != 0 : dflt;
}
#define ACX_DBG_SWITCH(name, dflt) ([] { static const bool v_ = acx_dbg_switch_env("ACX_" name, dflt); return v_; }())
#else
#define ACX_DBG_SWITCH(name, dflt) (dflt)
#endif

struct acx_ctx {
  int device;
  int multiprocessors;      // CUs of the device (persistent-kernel grid size)
  int opt_ring_min_tiles;   // ACX_OPT_RING_MIN_TILES
  int opt_sk_max_m;         // ACX_OPT_SK_MAX_M
  int opt_tn_p256_min_rows; // ACX_OPT_TN_P256_MIN_ROWS
  int opt_x6_cus;           // ACX_OPT_X6_CUS (0: all)
  int opt_x6_tail;          // ACX_OPT_X6_TAIL_SPLIT (0: off)
  int opt_x6_min_tiles;     // ACX_OPT_X6_MIN_TILES
  int opt_x6_strip;         // ACX_OPT_X6_STRIP_TAIL (1: by the cost model, 0: off, 2 / 3: always 128- / 64-column strips)
  char err[512];
  bool prof_on;
  bool prof_gemm_only;   // acx_prof_enable(ctx, 2): event pairs around the GEMM launches only
  int prof_n;          // recorded pairs
  int prof_created;    // event pairs created so far
  hipEvent_t* prof_ev; // [2 * ACX_PROF_MAX]
  unsigned char* prof_kind;
  double prof_gemm_flops;   // 2*M*N*K summed over acx_gemm / acx_gemm_tn launches while recording
  double prof_tn_flops, prof_tn_ms;   // the acx_gemm_tn share of it; its summed launch time (filled by acx_prof_collect)
  int prof_tn_count;
  void* comm;               // RCCL communicator (acx_comm_init), nullptr: none
  int comm_rank, comm_world;
};

struct AcxProfScope {
  acx_ctx* c;
  hipStream_t s;
  int slot;
  AcxProfScope(acx_ctx* ctx, int kind, hipStream_t stream) : c(ctx), s(stream), slot(-1) {
    if (!c || !c->prof_on || c->prof_n >= ACX_PROF_MAX) return;
    if (c->prof_gemm_only && kind != ACX_K_GEMM && kind != ACX_K_GEMM_TN) return;
    if (c->prof_n >= c->prof_created) {
      if (hipEventCreate(&c->prof_ev[2 * c->prof_created]) != hipSuccess) return;
      if (hipEventCreate(&c->prof_ev[2 * c->prof_created + 1]) != hipSuccess) return;
      c->prof_created++;
    }
    slot = c->prof_n++;
    c->prof_kind[slot] = (unsigned char)kind;
    (void)hipEventRecord(c->prof_ev[2 * slot], s);
  }
  ~AcxProfScope() {
    if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], s);
  }
};

// thread-local error slot for calls made with ctx == NULL
extern thread_local char acx_tls_err[512];

static inline int acx_fail(acx_ctx* ctx, int code, const char* fmt, const char* a = "", long b = 0,
                           long c = 0) {
  char* dst = ctx ? ctx->err : acx_tls_err;
  snprintf(dst, 512, fmt, a, b, c);
  return code;
}

static inline int acx_check_launch(acx_ctx* ctx, const char* name) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ACX_OK;
  snprintf(ctx ? ctx->err : acx_tls_err, 512, "%s: launch failed: %s", name, hipGetErrorString(e));
  return ACX_E_HIP;
}
#define ACX_CHECK_LAUNCH(ctx, name)              \
  do {                                           \
    int rc__ = acx_check_launch(ctx, name);      \
    if (rc__ != ACX_OK) return rc__;             \
  } while (0)

// acx_gemm.hip: true when acx_gemm runs `d` on the persistent strip-stream kernel (no partial last wave to split off)
bool acx_gemm_takes_strip_stream(const acx_gemm_desc* d);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// f32 -> bf16 round-to-nearest-even (same rounding torch uses for .to(torch.bfloat16)): gfx950 has the
// conversion in hardware (v_cvt_pk_bf16_f32), reached through the native __bf16 type
typedef __bf16 acx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float acx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16 f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {   // two values -> one packed dword
  const acx_f32x2 v = {lo, hi};
  const acx_bf16x2 r = __builtin_convertvector(v, acx_bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
// f32 -> fp16 (round-to-nearest-even, v_cvt_pk_f16_f32) for the TWO-plane fp16 split of the ACX_PREC_F16X3 mode: x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi); x - hi is exact in f32, hi + lo reproduces x to 2^-24 |x| while lo is a normal fp16 number (|x| above
// ~2^-3: below, lo's error is at most 2^-25 ABSOLUTE), |x| < 65504
typedef _Float16 acx_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2h2(float lo, float hi) {    // two values -> one packed dword of fp16
  const acx_f32x2 v = {lo, hi};
  const acx_f16x2 r = __builtin_convertvector(v, acx_f16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float h2f_lo(uint32_t p) { return (float)__builtin_bit_cast(acx_f16x2, p)[0]; }
__device__ __forceinline__ float h2f_hi(uint32_t p) { return (float)__builtin_bit_cast(acx_f16x2, p)[1]; }
// one packed pair of either plane format (FMT 0: bf16, the three-plane split; 1: fp16, the two-plane split) and its value back in f32
template <int FMT> __device__ __forceinline__ uint32_t acx_pk2(float a, float b) { if constexpr (FMT) return f2h2(a, b); else return f2bf2(a, b); }
template <int FMT> __device__ __forceinline__ float acx_unpk_lo(uint32_t p) { if constexpr (FMT) return h2f_lo(p); else return __uint_as_float(p << 16); }
template <int FMT> __device__ __forceinline__ float acx_unpk_hi(uint32_t p) { if constexpr (FMT) return h2f_hi(p); else return __uint_as_float(p & 0xffff0000u); }

// ---- BatchNorm arithmetic shared between the stand-alone kernels and the fused step kernels (acx_train.hip): ONE expression
// tree per quantity, so that hipcc's contraction choices -- and the bits -- are the same wherever it is inlined
// SyncBN combine (Chan et al.) of column c: gathered[r] = (mean[C], biased_var[C] * rows_r, rows_r) of rank r, ranks in order
__device__ __forceinline__ void acx_bn_combine_col(const float* __restrict__ g, int R, int C, int c, float& n_out, float& mean,
                                                   float& var_b, float& var_u) {
  const int ld = 2 * C + 1;
  float n = 0.f;
  for (int r = 0; r < R; ++r) n += g[(size_t)r * ld + 2 * C];
  float m = 0.f;
  for (int r = 0; r < R; ++r) m += g[(size_t)r * ld + c] * (g[(size_t)r * ld + 2 * C] / n);
  float m2 = 0.f;
  for (int r = 0; r < R; ++r) {
    const float dm = g[(size_t)r * ld + c] - m;
    m2 += g[(size_t)r * ld + C + c] + g[(size_t)r * ld + 2 * C] * (dm * dm);
  }
  n_out = n; mean = m; var_b = m2 / n; var_u = m2 / fmaxf(n - 1.f, 1.f);
}
// nn.BatchNorm1d running statistic: r = (1 - momentum) r + momentum x
__device__ __forceinline__ float acx_bn_running(float momentum, float om, float x, float r) { return momentum * x + om * r; }
// training BatchNorm output
__device__ __forceinline__ float acx_bn_apply(float raw, float mean, float var, float eps) { return (raw - mean) / sqrtf(var + eps); }
// f64 column sums of stage 1: (x, x^2) for the forward statistics (K2 = 0), (dl, dl * xhat) for the backward ones (K2 = 1)
template <int K2>
__device__ __forceinline__ void acx_bn_acc(double& s, double& qq, double v, double g) {
  if constexpr (K2 == 0) { s += v; qq += v * v; }
  else { s += g; qq += g * v; }
}
// stage 2: 256 threads; pair p = (kind, column) < 2*C1 is summed by G = 256 / (2*C1) threads, thread g of a pair taking blocks
// g, g + G, ... in order; the G partial sums meet in LDS and are added in g order -- a pure function of (nblocks, C1), so
// results are run-to-run identical.  (One thread per pair walking 512 blocks took 120 us: a chain of dependent L2 misses.)
// COHERENT: the partials were published by other workgroups of the SAME launch (agent-scope atomic loads: last-arriver form)
template <bool COHERENT = false>
__device__ __forceinline__ double bn_pair_total(const double* __restrict__ part, int nblocks, int CP, int C1, double* red) {
  const int P = 2 * C1, G = 256 / P;
  const int p = threadIdx.x % P, g = threadIdx.x / P;
  double v = 0.0;
  if (g < G) {
    const int k = p / C1, c = p - k * C1;
    const double* src = part + (size_t)k * CP + c;
    const size_t bs = (size_t)2 * CP;
    int b = g;
    for (; b + 7 * G < nblocks; b += 8 * G) {                         // eight independent loads in flight, added in block order
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (COHERENT) t8[u] = __hip_atomic_load(src + (size_t)(b + u * G) * bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else t8[u] = src[(size_t)(b + u * G) * bs];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t8[u];
    }
    for (; b < nblocks; b += G) {
      if constexpr (COHERENT) v += __hip_atomic_load(src + (size_t)b * bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v += src[(size_t)b * bs];
    }
  }
  red[threadIdx.x] = v;
  __syncthreads();
  double tot = 0.0;
  if ((int)threadIdx.x < P)
    for (int gg = 0; gg < G; ++gg) tot += red[gg * P + threadIdx.x];
  return tot;                                                          // valid for threadIdx.x < 2*C1: pair (k = t / C1, c = t % C1)
}

// ---- wave-per-row register layout shared by the row kernels
// lane owns elements: VPL%4==0 -> float4 groups at 4*lane + 256*i ; else scalar at lane + 64*i
template <int VPL>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int lane, float (&v)[VPL]) {
  if constexpr (VPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < VPL / 4; ++i) {
      const float4 t = *reinterpret_cast<const float4*>(p + 4 * lane + 256 * i);
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = p[lane + 64 * i];
  }
}
template <int VPL>
__device__ __forceinline__ int elem_index(int lane, int i) {
  if constexpr (VPL % 4 == 0) return 4 * lane + 256 * (i >> 2) + (i & 3);
  return lane + 64 * i;
}

