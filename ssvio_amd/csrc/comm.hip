// ssvio_amd/csrc/comm.hip -- RCCL inside the library ("RCCL all-reduce over xGMI of the pose
// blocks"): ssx_comm_* of include/ssx.h.  The reference is a single process (no counterpart); one process per GPU
// creates one communicator for its ctx's device, ssx_ba_solve then sums its exchange buffers with ncclAllReduce
// (f64, sum, in place) ENQUEUED ON THE CTX STREAM -- no host round trip, no callback into the host language.
//
// librccl is bound at run time (dlopen), not at link time: a single-GPU user never loads it, and a process that
// already holds a copy (PyTorch-ROCm bundles its own librccl.so.1) keeps exactly that one -- two RCCL instances in one
// process would not share communicators.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "ctx.hpp"

struct ssx_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  bool owned = true;          // created by ssx_comm_init (destroyed with the handle) or wrapped (caller's)
};

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char err[256] = {0};
};

RcclApi* rccl()
{
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) { snprintf(api.err, sizeof(api.err), "librccl.so.1 not found: %s", dlerror()); return; }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) {
      snprintf(api.err, sizeof(api.err), "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
      api.handle = nullptr;
    }
  });
  return &api;
}

const char* rccl_error(RcclApi* a, ncclResult_t r) { return a->GetErrorString ? a->GetErrorString(r) : "rccl error"; }

}  // namespace

// The collective of ssx_ba_solve when ssx_ba_options.comm is set: same contract as an ssx_allreduce_fn
int ssx_comm_allreduce_f64(void* user, double* buf_dev, size_t count, void* stream)
{
  ssx_comm* c = static_cast<ssx_comm*>(user);
  RcclApi* a = rccl();
  if (!c || !c->comm || !a->handle) return 1;
  return a->AllReduce(buf_dev, buf_dev, count, ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

extern "C" {

static_assert(sizeof(ssx_comm_id) == sizeof(ncclUniqueId), "ssx_comm_id carries an ncclUniqueId");

ssx_status ssx_comm_unique_id(ssx_ctx* ctx, ssx_comm_id* out)
{
  if (!ctx || !out) return SSX_ERR_INVALID_ARG;
  RcclApi* a = rccl();
  if (!a->handle) { ctx->set_error("ssx_comm: %s", a->err); return SSX_ERR_COMM; }
  ncclUniqueId id;
  const ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) { ctx->set_error("ssx_comm: ncclGetUniqueId -> %s", rccl_error(a, r)); return SSX_ERR_COMM; }
  memcpy(out->bytes, id.internal, sizeof(out->bytes));
  return SSX_OK;
}

ssx_status ssx_comm_init(ssx_ctx* ctx, const ssx_comm_id* id, int32_t rank, int32_t world_size, ssx_comm** out)
{
  if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return SSX_ERR_INVALID_ARG;
  *out = nullptr;
  RcclApi* a = rccl();
  if (!a->handle) { ctx->set_error("ssx_comm: %s", a->err); return SSX_ERR_COMM; }
  SSX_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ncclUniqueId nid;
  memcpy(nid.internal, id->bytes, sizeof(nid.internal));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = a->CommInitRank(&comm, world_size, nid, rank);      // collective: every rank calls it
  if (r != ncclSuccess) { ctx->set_error("ssx_comm: ncclCommInitRank(rank %d of %d) -> %s", rank, world_size, rccl_error(a, r)); return SSX_ERR_COMM; }
  ssx_comm* c = new ssx_comm();
  c->comm = comm; c->rank = rank; c->world = world_size; c->device = ctx->device; c->owned = true;
  *out = c;
  return SSX_OK;
}

ssx_status ssx_comm_wrap(ssx_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world_size, ssx_comm** out)
{
  if (!ctx || !nccl_comm || !out || world_size < 1 || rank < 0 || rank >= world_size) return SSX_ERR_INVALID_ARG;
  RcclApi* a = rccl();
  if (!a->handle) { ctx->set_error("ssx_comm: %s", a->err); return SSX_ERR_COMM; }
  ssx_comm* c = new ssx_comm();
  c->comm = static_cast<ncclComm_t>(nccl_comm); c->rank = rank; c->world = world_size; c->device = ctx->device; c->owned = false;
  *out = c;
  return SSX_OK;
}

void ssx_comm_destroy(ssx_comm* c)
{
  if (!c) return;
  RcclApi* a = rccl();
  if (c->owned && c->comm && a->handle) {
    (void)hipSetDevice(c->device);
    (void)a->CommDestroy(c->comm);
  }
  delete c;
}

ssx_status ssx_comm_info(const ssx_comm* c, int32_t* rank, int32_t* world_size)
{
  if (!c) return SSX_ERR_INVALID_ARG;
  if (rank) *rank = c->rank;
  if (world_size) *world_size = c->world;
  return SSX_OK;
}

// In-place f64 sum over the communicator, enqueued on the ctx stream (what ssx_ba_solve issues; exposed for tests and
// for callers that exchange their own buffers, e.g. the final pose gather of a sharded map).
ssx_status ssx_comm_allreduce_sum(ssx_ctx* ctx, ssx_comm* c, double* buf_dev, size_t count)
{
  if (!ctx || !c || (!buf_dev && count)) return SSX_ERR_INVALID_ARG;
  if (count == 0) return SSX_OK;
  if (ssx_comm_allreduce_f64(c, buf_dev, count, ctx->stream) != 0) { ctx->set_error("ssx_comm: ncclAllReduce failed"); return SSX_ERR_COMM; }
  return SSX_OK;
}

}  // extern "C"
