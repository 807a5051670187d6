// Collective entry points of the C ABI (npp_comm_init / npp_allreduce_bucket / npp_syncbn_exchange): RCCL over
// xGMI, one communicator per process (one process per GPU), enqueued on the stream the caller passes -- no helper stream and no
// host synchronisation inside the library, so the calls can sit inside a hipGraph capture on whatever stream the step uses.
// Replaces what torch.distributed's ProcessGroupNCCL does for the reference's DistributedDataParallel gradient all-reduce
// (augment_lip_sync.py:206-208) and SyncBatchNorm statistics exchange (augment_lip_sync.py:191).
// librccl is opened at run time (dlopen) so that libnpp_hip.so loads on a box without it; the single-GPU path never needs it.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 0;
  std::mutex mu;
} g;

bool load_rccl() {
  if (g.h) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g.h) break;
  }
  if (!g.h) { npp_set_error("npp_comm: cannot open librccl.so (%s)", dlerror()); return false; }
  g.GetUniqueId = reinterpret_cast<decltype(g.GetUniqueId)>(dlsym(g.h, "ncclGetUniqueId"));
  g.CommInitRank = reinterpret_cast<decltype(g.CommInitRank)>(dlsym(g.h, "ncclCommInitRank"));
  g.CommDestroy = reinterpret_cast<decltype(g.CommDestroy)>(dlsym(g.h, "ncclCommDestroy"));
  g.CommCount = reinterpret_cast<decltype(g.CommCount)>(dlsym(g.h, "ncclCommCount"));
  g.AllReduce = reinterpret_cast<decltype(g.AllReduce)>(dlsym(g.h, "ncclAllReduce"));
  g.GetErrorString = reinterpret_cast<decltype(g.GetErrorString)>(dlsym(g.h, "ncclGetErrorString"));
  if (!g.GetUniqueId || !g.CommInitRank || !g.CommDestroy || !g.AllReduce || !g.GetErrorString) {
    npp_set_error("npp_comm: librccl.so lacks a required symbol");
    return false;
  }
  return true;
}

int rccl_fail(const char* what, ncclResult_t r) {
  npp_set_error("%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "RCCL error");
  return NPP_E_RCCL;
}

}  // namespace

extern "C" int npp_comm_unique_id(void* id128) {
  NPP_REQUIRE(id128, NPP_E_NULL, "npp_comm_unique_id: null pointer");
  std::lock_guard<std::mutex> lk(g.mu);
  if (!load_rccl()) return NPP_E_RCCL;
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  ncclUniqueId id;
  const ncclResult_t r = g.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id128, &id, sizeof(id));
  return NPP_OK;
}

extern "C" int npp_comm_init(const void* id128, int rank, int world) {
  NPP_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, NPP_E_SHAPE, "npp_comm_init: bad rank %d of %d", rank, world);
  std::lock_guard<std::mutex> lk(g.mu);
  if (!load_rccl()) return NPP_E_RCCL;
  NPP_REQUIRE(g.comm == nullptr, NPP_E_UNSUPPORTED, "npp_comm_init: a communicator exists already (npp_comm_destroy first)");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = g.CommInitRank(&g.comm, world, id, rank);      // uses the calling thread's current device
  if (r != ncclSuccess) { g.comm = nullptr; return rccl_fail("ncclCommInitRank", r); }
  g.rank = rank; g.world = world;
  return NPP_OK;
}

// the number of ranks the COMMUNICATOR reports (ncclCommCount), not the value it was created with; 0 without a communicator
extern "C" int npp_comm_world(void) {
  if (!g.comm) return 0;
  int n = 0;
  if (g.CommCount && g.CommCount(g.comm, &n) == ncclSuccess) return n;
  return g.world;
}

extern "C" int npp_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.comm) {
    const ncclResult_t r = g.CommDestroy(g.comm);
    g.comm = nullptr; g.world = 0;
    if (r != ncclSuccess) return rccl_fail("ncclCommDestroy", r);
  }
  return NPP_OK;
}

// in-place all-reduce of one gradient bucket: dtype NPP_F32 (or NPP_BF16); average != 0 divides by the world size in the collective
extern "C" int npp_allreduce_bucket(void* buf, int64_t count, int dtype, int average, void* stream) {
  NPP_REQUIRE(buf && count > 0, NPP_E_NULL, "npp_allreduce_bucket: null / empty buffer");
  NPP_REQUIRE(g.comm, NPP_E_RCCL, "npp_allreduce_bucket: no communicator (npp_comm_init)");
  NPP_REQUIRE(dtype == NPP_F32 || dtype == NPP_BF16, NPP_E_DTYPE, "npp_allreduce_bucket: dtype must be f32 or bf16");
  const ncclResult_t r = g.AllReduce(buf, buf, (size_t)count, dtype == NPP_F32 ? ncclFloat32 : ncclBfloat16,
                                     average ? ncclAvg : ncclSum, g.comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
  return NPP_OK;
}

// SyncBatchNorm statistics: in-place SUM of `count` f64 partial sums ([replica][sum | sumsq] vectors of every BatchNorm produced
// since the last exchange, back to back in one buffer; the finalize kernel then divides by the GLOBAL count)
extern "C" int npp_syncbn_exchange(double* stats, int64_t count, void* stream) {
  NPP_REQUIRE(stats && count > 0, NPP_E_NULL, "npp_syncbn_exchange: null / empty buffer");
  NPP_REQUIRE(g.comm, NPP_E_RCCL, "npp_syncbn_exchange: no communicator (npp_comm_init)");
  const ncclResult_t r = g.AllReduce(stats, stats, (size_t)count, ncclFloat64, ncclSum, g.comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce(f64)", r);
  return NPP_OK;
}
