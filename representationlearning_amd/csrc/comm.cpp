// Data-parallel exchange steps of the RSSFormer training step behind the C ABI: one RCCL
// communicator per (process, purpose) over the GPUs of one node, driven on the CALLER's hipStream so that every collective
// is just another node of the captured training step.  Replaces, on the reference side, what `ever`'s th_amp_ddp trainer
// gets from torch DistributedDataParallel + nn.SyncBatchNorm (configs/base/loveda.py:106-108, train.py:79,
// modules/ffn_block.py:222-234).
//
// RCCL is bound at run time (dlopen) rather than at link time: the process already carries the RCCL build that PyTorch
// ships, and a second copy linked in here would be a second, incompatible runtime.  The caller names the library
// (rccl_path) or passes NULL for the default search ("librccl.so.1", "librccl.so").  The resolved function table is
// written once (std::call_once) and read-only afterwards; communicators are caller-owned handles.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <new>
#include "../../include/rssf.h"

namespace rssf {
void set_error(const char* fmt, ...);
}

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
enum { kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(const void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_once;

void bind(const char* path) {
  const char* names[3] = {path, "librccl.so.1", "librccl.so"};
  for (int i = path ? 0 : 1; i < 3 && !g_rccl.handle; ++i) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "cannot load RCCL (%s)", dlerror());
    return;
  }
  bool ok = true;
  auto sym = [&](const char* n) {
    void* p = dlsym(g_rccl.handle, n);
    if (!p) { ok = false; snprintf(g_rccl.why, sizeof(g_rccl.why), "RCCL symbol %s missing", n); }
    return p;
  };
  g_rccl.GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
  g_rccl.CommCount = (int (*)(const void*, int*))sym("ncclCommCount");
  g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (!ok) g_rccl.handle = nullptr;
}

const Rccl* rccl(const char* path) {
  std::call_once(g_once, bind, path);
  if (!g_rccl.handle) {
    rssf::set_error("rssf_comm: %s", g_rccl.why);
    return nullptr;
  }
  return &g_rccl;
}

int fail(const Rccl* r, int rc, const char* what) {
  rssf::set_error("%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
  return RSSF_ERR_LAUNCH;
}

}  // namespace

struct rssf_comm {
  void* nccl;
  int rank, world;
};

extern "C" int rssf_comm_unique_id(void* id128, const char* rccl_path) {
  const Rccl* r = rccl(rccl_path);
  if (!r) return RSSF_ERR_UNSUPPORTED;
  if (!id128) { rssf::set_error("rssf_comm_unique_id: NULL buffer"); return RSSF_ERR_BAD_ARG; }
  UniqueId id;
  const int rc = r->GetUniqueId(&id);
  if (rc) return fail(r, rc, "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return RSSF_OK;
}

extern "C" int rssf_comm_init(rssf_comm** comm, int rank, int world, const void* id128, const char* rccl_path) {
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) {
    rssf::set_error("rssf_comm_init: bad arguments (rank %d, world %d)", rank, world);
    return RSSF_ERR_BAD_ARG;
  }
  const Rccl* r = rccl(rccl_path);
  if (!r) return RSSF_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* h = nullptr;
  const int rc = r->CommInitRank(&h, world, id, rank);
  if (rc) return fail(r, rc, "ncclCommInitRank");
  rssf_comm* c = new (std::nothrow) rssf_comm{h, rank, world};
  if (!c) { r->CommDestroy(h); rssf::set_error("rssf_comm_init: out of memory"); return RSSF_ERR_LAUNCH; }
  *comm = c;
  return RSSF_OK;
}

extern "C" int rssf_comm_rank(const rssf_comm* c) { return c ? c->rank : -1; }
extern "C" int rssf_comm_world(const rssf_comm* c) { return c ? c->world : 0; }

// what RCCL ITSELF says the communicator spans (ncclCommCount), not what the caller passed to rssf_comm_init: bench.py prints it
// so that a multi-GPU line can be checked against "RCCL saw N ranks"
extern "C" int rssf_comm_nranks(const rssf_comm* c) {
  if (!c || !c->nccl) return 0;
  const Rccl* r = rccl(nullptr);
  if (!r) return 0;
  int n = 0;
  const int rc = r->CommCount(c->nccl, &n);
  if (rc) { fail(r, rc, "ncclCommCount"); return -1; }
  return n;
}

extern "C" int rssf_allreduce_bucket(void* buf, int64_t count, int dtype, rssf_comm* c, void* stream) {
  if (!c || !buf || count <= 0) { rssf::set_error("rssf_allreduce_bucket: bad arguments"); return RSSF_ERR_BAD_ARG; }
  if (dtype != RSSF_F32 && dtype != RSSF_BF16) { rssf::set_error("rssf_allreduce_bucket: unsupported dtype %d", dtype); return RSSF_ERR_UNSUPPORTED; }
  const Rccl* r = rccl(nullptr);
  if (!r) return RSSF_ERR_UNSUPPORTED;
  const int rc = r->AllReduce(buf, buf, (size_t)count, dtype == RSSF_F32 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, c->nccl, (hipStream_t)stream);
  return rc ? fail(r, rc, "ncclAllReduce(bucket)") : RSSF_OK;
}

extern "C" int rssf_syncbn_exchange(float* stats, int64_t count, rssf_comm* c, void* stream) {
  if (!c || !stats || count <= 0) { rssf::set_error("rssf_syncbn_exchange: bad arguments"); return RSSF_ERR_BAD_ARG; }
  const Rccl* r = rccl(nullptr);
  if (!r) return RSSF_ERR_UNSUPPORTED;
  const int rc = r->AllReduce(stats, stats, (size_t)count, kNcclFloat32, kNcclSum, c->nccl, (hipStream_t)stream);
  return rc ? fail(r, rc, "ncclAllReduce(syncbn)") : RSSF_OK;
}

extern "C" int rssf_comm_destroy(rssf_comm* c) {
  if (!c) return RSSF_OK;
  const Rccl* r = rccl(nullptr);
  int rc = 0;
  if (r && c->nccl) rc = r->CommDestroy(c->nccl);
  delete c;
  return (r && rc) ? fail(r, rc, "ncclCommDestroy") : RSSF_OK;
}
