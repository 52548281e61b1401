// libacx collectives: thin RCCL calls on the CALLER's stream (host code only; no kernel of its own).
// SURVEY.md section 8(b) lists `acx_comm_init(ctx, rank, world, unique_id)` / `acx_allreduce(ctx, buf, count, dtype, stream)` among the
// entry points of the boundary.  The reference's exchange is Lightning DDP over NCCL (configs/trainer/ddp.yaml:1-9: strategy ddp,
// sync_batchnorm); the Python host of this repository drives the same exchange through torch.distributed ("nccl" IS RCCL on ROCm:
// anomalyclip_amd/parallel.py), and that stays the default.  These entry points give a C / C++ host -- or a HIP graph: a collective
// issued here on a capturing stream is recorded like any kernel -- the same collectives without torch: RCCL is opened at run time
// (dlopen: libacx.so carries no link-time dependency on it), the communicator belongs to the context, the caller owns every buffer.
#include "acx_internal.h"

#include <dlfcn.h>
#include <string.h>

namespace {
// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes; enums by value) -- declared here so that the build
// does not depend on the RCCL headers either
struct AcxNcclId { char internal[128]; };
typedef void* nccl_comm_t;
typedef int (*fn_get_unique_id)(AcxNcclId*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, AcxNcclId, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3, NCCL_INT64 = 4, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_BFLOAT16 = 9 };

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
  char why[256] = {0};
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return &r;
  tried = true;
  // a process that already carries an RCCL (PyTorch-ROCm loads its own) keeps using THAT one; otherwise the ROCm installation's
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.handle) break; }
  if (!r.handle) for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.handle) break; }
  if (!r.handle) { snprintf(r.why, sizeof(r.why), "librccl.so not found (%s)", dlerror()); return &r; }
  r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
  r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
  r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
  r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
  r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather) {
    snprintf(r.why, sizeof(r.why), "librccl.so lacks an expected symbol");
    r.handle = nullptr;
  }
  return &r;
}

int rccl_fail(acx_ctx* ctx, const char* what, int rc) {
  Rccl* r = rccl();
  const char* es = r->error_string ? r->error_string(rc) : "?";
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: RCCL error %d (%s)", what, rc, es);
  else snprintf(acx_tls_err, 512, "%s: RCCL error %d (%s)", what, rc, es);
  return ACX_E_RCCL;
}

int nccl_dtype(int dtype) {
  switch (dtype) {
    case ACX_F32: return NCCL_FLOAT32;
    case ACX_BF16: return NCCL_BFLOAT16;
    case ACX_F64: return NCCL_FLOAT64;
    case ACX_I64: return NCCL_INT64;
    default: return -1;
  }
}
}  // namespace

extern "C" int acx_comm_unique_id(void* id_out, size_t id_bytes) {
  if (!id_out || id_bytes < ACX_COMM_ID_BYTES) return acx_fail(nullptr, ACX_E_BADARG, "acx_comm_unique_id: needs ACX_COMM_ID_BYTES bytes%s");
  Rccl* r = rccl();
  if (!r->handle) { snprintf(acx_tls_err, 512, "acx_comm_unique_id: %s", r->why); return ACX_E_RCCL; }
  AcxNcclId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return rccl_fail(nullptr, "acx_comm_unique_id", rc);
  memcpy(id_out, &id, sizeof(id));
  return ACX_OK;
}

extern "C" int acx_comm_init(acx_ctx* ctx, int32_t rank, int32_t world, const void* unique_id) {
  if (!ctx || !unique_id) return acx_fail(ctx, ACX_E_BADARG, "acx_comm_init: null pointer%s");
  if (world < 1 || rank < 0 || rank >= world) return acx_fail(ctx, ACX_E_BADARG, "acx_comm_init: rank outside [0, world)%s");
  if (ctx->comm) return acx_fail(ctx, ACX_E_BADARG, "acx_comm_init: the context already owns a communicator (acx_comm_destroy first)%s");
  Rccl* r = rccl();
  if (!r->handle) return acx_fail(ctx, ACX_E_RCCL, "acx_comm_init: %s", r->why);
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (cur != ctx->device) (void)hipSetDevice(ctx->device);        // the communicator is bound to the context's device
  AcxNcclId id;
  memcpy(&id, unique_id, sizeof(id));
  nccl_comm_t comm = nullptr;
  const int rc = r->comm_init_rank(&comm, world, id, rank);
  if (cur != ctx->device && cur >= 0) (void)hipSetDevice(cur);
  if (rc) return rccl_fail(ctx, "acx_comm_init", rc);
  ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
  return ACX_OK;
}

extern "C" int acx_comm_destroy(acx_ctx* ctx) {
  if (!ctx) return acx_fail(ctx, ACX_E_BADARG, "acx_comm_destroy: no context%s");
  if (!ctx->comm) return ACX_OK;
  const int rc = rccl()->comm_destroy((nccl_comm_t)ctx->comm);
  ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 0;
  return rc ? rccl_fail(ctx, "acx_comm_destroy", rc) : ACX_OK;
}

extern "C" int acx_comm_info(acx_ctx* ctx, int32_t* rank, int32_t* world) {
  if (!ctx || !rank || !world) return acx_fail(ctx, ACX_E_BADARG, "acx_comm_info: null pointer%s");
  *rank = ctx->comm ? ctx->comm_rank : 0;
  *world = ctx->comm ? ctx->comm_world : 0;
  return ACX_OK;
}

extern "C" int acx_allreduce(acx_ctx* ctx, void* buf, int64_t count, int32_t dtype, int32_t op, void* stream) {
  if (!ctx || (!buf && count > 0)) return acx_fail(ctx, ACX_E_BADARG, "acx_allreduce: null pointer%s");
  if (!ctx->comm) return acx_fail(ctx, ACX_E_BADARG, "acx_allreduce: no communicator (acx_comm_init)%s");
  const int dt = nccl_dtype(dtype);
  const int rop = op == ACX_COMM_SUM ? NCCL_SUM : op == ACX_COMM_MAX ? NCCL_MAX : op == ACX_COMM_MIN ? NCCL_MIN : -1;
  if (dt < 0 || rop < 0 || count < 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_allreduce: dtype f32 / bf16 / f64 / i64, op sum / max / min%s");
  if (count == 0) return ACX_OK;
  const int rc = rccl()->all_reduce(buf, buf, (size_t)count, dt, rop, (nccl_comm_t)ctx->comm, (hipStream_t)stream);   // in place
  return rc ? rccl_fail(ctx, "acx_allreduce", rc) : ACX_OK;
}

extern "C" int acx_allgather(acx_ctx* ctx, const void* send, void* recv, int64_t count, int32_t dtype, void* stream) {
  if (!ctx || ((!send || !recv) && count > 0)) return acx_fail(ctx, ACX_E_BADARG, "acx_allgather: null pointer%s");
  if (!ctx->comm) return acx_fail(ctx, ACX_E_BADARG, "acx_allgather: no communicator (acx_comm_init)%s");
  const int dt = nccl_dtype(dtype);
  if (dt < 0 || count < 0) return acx_fail(ctx, ACX_E_UNSUPPORTED, "acx_allgather: dtype f32 / bf16 / f64 / i64%s");
  if (count == 0) return ACX_OK;
  const int rc = rccl()->all_gather(send, recv, (size_t)count, dt, (nccl_comm_t)ctx->comm, (hipStream_t)stream);
  return rc ? rccl_fail(ctx, "acx_allgather", rc) : ACX_OK;
}
