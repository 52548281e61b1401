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

