// The one collective of the path - the broadcast of the prepared inverse-sweep spectrum from rank 0 - done by the
// library itself over RCCL (xGMI inside a node), so that the host side needs no Python communication package.
// librccl is opened on first use (dlopen): a single-GPU process never loads it and the library has no link-time
// dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return IMP_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return fail(IMP_ERR_UNSUPPORTED, "librccl not found (%s)", dlerror());
  Rccl r;
  r.handle = h;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
  if (!r.CommCount || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.GetErrorString)
    return fail(IMP_ERR_UNSUPPORTED, "librccl lacks an expected entry point");
  g_rccl = r;
  return IMP_OK;
}

}  // namespace

struct imp_comm {
  imp_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
};

#define RCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return fail(IMP_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

extern "C" int imp_comm_unique_id(unsigned char id_out[128]) {
  if (!id_out) return fail(IMP_ERR_INVALID, "imp_comm_unique_id: null argument");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "the unique id travels as 128 bytes");
  std::memcpy(id_out, &id, sizeof(id));
  return IMP_OK;
}

extern "C" int imp_comm_create(imp_ctx* ctx, const unsigned char id[128], int rank, int nranks, imp_comm** out) {
  if (!ctx || !id || !out) return fail(IMP_ERR_INVALID, "imp_comm_create: null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(IMP_ERR_INVALID, "imp_comm_create: rank %d of %d", rank, nranks);
  int rc = rccl_load();
  if (rc) return rc;
  IMP_CTX_LOCK(ctx);
  if ((rc = ctx_bind(ctx))) return rc;
  imp_comm* c = new (std::nothrow) imp_comm();
  if (!c) return fail(IMP_ERR_ALLOC, "out of host memory");
  c->ctx = ctx;
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(IMP_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
  }
  *out = c;
  return IMP_OK;
}

// librccl loadable and complete?  No communicator, no device work: the ranks agree on this BEFORE any of them enters
// ncclCommInitRank, which has no timeout (a rank that failed to load the library would leave the others blocked there).
extern "C" int imp_comm_probe(void) { return rccl_load(); }

// ranks of the communicator as RCCL itself counts them (ncclCommCount)
extern "C" int imp_comm_nranks(imp_comm* c, int* nranks) {
  if (!c || !nranks) return fail(IMP_ERR_INVALID, "imp_comm_nranks: null argument");
  RCCL_TRY(g_rccl.CommCount(c->comm, nranks));
  return IMP_OK;
}

extern "C" void imp_comm_destroy(imp_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
}

extern "C" int imp_comm_broadcast(imp_comm* c, void* dptr, size_t bytes, int root) {
  if (!c || (!dptr && bytes)) return fail(IMP_ERR_INVALID, "imp_comm_broadcast: null argument");
  if (root < 0 || root >= c->nranks) return fail(IMP_ERR_INVALID, "imp_comm_broadcast: root %d of %d", root, c->nranks);
  if (!bytes) return IMP_OK;
  IMP_CTX_LOCK(c->ctx);
  int rc = ctx_bind(c->ctx);
  if (rc) return rc;
  RCCL_TRY(g_rccl.Broadcast(dptr, dptr, bytes, ncclUint8, root, c->comm, c->ctx->stream));
  HIP_TRY(hipStreamSynchronize(c->ctx->stream));
  return IMP_OK;
}

extern "C" int imp_plan_broadcast_spectrum(imp_plan* plan, imp_comm* c, int root, size_t* bytes_out) {
  if (!plan || !c) return fail(IMP_ERR_INVALID, "imp_plan_broadcast_spectrum: null argument");
  void* dptr = nullptr;
  size_t bytes = 0;
  int rc = imp_plan_spectrum(plan, &dptr, &bytes);
  if (rc) return rc;
  if (bytes_out) *bytes_out = bytes;
  return imp_comm_broadcast(c, dptr, bytes, root);
}
