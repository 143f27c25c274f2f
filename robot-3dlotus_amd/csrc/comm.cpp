// lotus-hip: the collectives of the data-parallel step, issued straight into the CALLER'S HIP stream through RCCL
// (genrobo3d/train/utils/distributed.py:196-205 wraps the model in DistributedDataParallel, train_simple_policy.py:116-117
// converts to SyncBatchNorm: both end in NCCL all-reduces issued by ProcessGroupNCCL).  Why not torch.distributed for these:
// a blocking ProcessGroupNCCL collective costs the stream it is issued on ~11 us beyond the collective itself (work object,
// end event = a marker packet in the queue; 26 us through its own stream) — measured with tools/dbg/msg_cost.py — and the
// step has 36 latency-bound statistics messages on its critical stream.  ncclAllReduce on a communicator of our own is ONE
// kernel in the stream the producer and the consumer of the message run on: no event, no second stream, no hardware queue.
//
// RCCL is not a link-time dependency (the library must load on boxes without it): lotus_comm_load() dlopen()s the librccl the
// process already uses (the Python host passes the path of the one torch loaded, so both share one runtime instance).
// torch.distributed stays the bootstrap: it carries the 128-byte unique id from rank 0 to the others.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lotus_hip.h"

void lotus_set_error(const char* fmt, ...);
#define LOTUS_CHECK_ARG(cond, ...)  \
  do {                              \
    if (!(cond)) {                  \
      lotus_set_error(__VA_ARGS__); \
      return LOTUS_E_ARG;           \
    }                               \
  } while (0)

namespace {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
// ncclDataType_t / ncclRedOp_t values of nccl.h (stable across NCCL 2.x and RCCL)
enum { kNcclInt32 = 2, kNcclFloat32 = 7, kNcclFloat64 = 8 };
enum { kNcclSum = 0, kNcclMax = 2, kNcclAvg = 4 };

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
} R;

const char* err_str(ncclResult_t r) { return R.GetErrorString ? R.GetErrorString(r) : "?"; }
}  // namespace

extern "C" {

int lotus_comm_load(const char* path) {
  if (R.handle) return LOTUS_OK;
  const char* names[] = {path, "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    lotus_set_error("lotus_comm_load: cannot open RCCL (%s)", dlerror());
    return LOTUS_E_UNSUPPORTED;
  }
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
    lotus_set_error("lotus_comm_load: the library lacks the NCCL entry points");
    dlclose(h);
    return LOTUS_E_UNSUPPORTED;
  }
  R = r;
  return LOTUS_OK;
}

int lotus_comm_version(void) {
  int v = 0;
  if (R.GetVersion) (void)R.GetVersion(&v);
  return v;
}

int lotus_comm_unique_id(void* id128_host) {
  LOTUS_CHECK_ARG(id128_host, "lotus_comm_unique_id: null buffer");
  if (!R.handle) {
    lotus_set_error("lotus_comm_unique_id: lotus_comm_load() first");
    return LOTUS_E_UNSUPPORTED;
  }
  ncclUniqueId id;
  ncclResult_t r = R.GetUniqueId(&id);
  if (r != 0) {
    lotus_set_error("ncclGetUniqueId: %s", err_str(r));
    return LOTUS_E_LAUNCH;
  }
  memcpy(id128_host, &id, sizeof(id));
  return LOTUS_OK;
}

unsigned long long lotus_comm_create(const void* id128_host, int nranks, int rank) {
  if (!R.handle || !id128_host || nranks < 1 || rank < 0 || rank >= nranks) {
    lotus_set_error("lotus_comm_create: bad arguments (or lotus_comm_load() not called)");
    return 0;
  }
  ncclUniqueId id;
  memcpy(&id, id128_host, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t r = R.CommInitRank(&c, nranks, id, rank);
  if (r != 0 || !c) {
    lotus_set_error("ncclCommInitRank(%d of %d): %s", rank, nranks, err_str(r));
    return 0;
  }
  return (unsigned long long)(uintptr_t)c;
}

int lotus_comm_allreduce(unsigned long long comm, void* buf, size_t count, int dtype, int op, void* stream) {
  LOTUS_CHECK_ARG(comm && buf && dtype >= 0 && dtype <= 2 && op >= 0 && op <= 2, "lotus_comm_allreduce: bad arguments");
  if (count == 0) return LOTUS_OK;
  static const int dt[3] = {kNcclFloat32, kNcclFloat64, kNcclInt32};
  static const int ops[3] = {kNcclSum, kNcclMax, kNcclAvg};
  ncclResult_t r = R.AllReduce(buf, buf, count, dt[dtype], ops[op], (ncclComm_t)(uintptr_t)comm, (hipStream_t)stream);
  if (r != 0) {
    lotus_set_error("ncclAllReduce: %s", err_str(r));
    return LOTUS_E_LAUNCH;
  }
  return LOTUS_OK;
}

int lotus_comm_destroy(unsigned long long comm) {
  if (comm && R.CommDestroy) (void)R.CommDestroy((ncclComm_t)(uintptr_t)comm);
  return LOTUS_OK;
}

}  // extern "C"
