// Collectives of the data-parallel path as C-ABI entry points over RCCL (xGMI), for callers that bind libpvrl_hip.so
// directly: the gradient all-reduce that replaces DistributedDataParallel's reducer
// (reference lib/models/build.py:49-53) and the all-gather of lib/utils/distributed.py:13-50.  The Python host side of this
// repo reaches the same RCCL through torch.distributed (backend "nccl"); these entry points are the equivalent for a host
// without torch.  RCCL is bound at run time (dlopen, preferring a copy already loaded into the process -- PyTorch ships its
// own librccl.so and two copies in one process must not be mixed), so libpvrl_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include "../../include/pvrl.h"
#include "common.h"

namespace {
typedef struct { char internal[128]; } rcclUniqueId;          // ncclUniqueId (nccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm_t;
struct Rccl {
  int (*GetUniqueId)(rcclUniqueId*);
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
  int (*CommDestroy)(rcclComm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  bool ok;
};
Rccl* rccl() {
  static Rccl r = [] {
    Rccl x;
    memset(&x, 0, sizeof(x));
    void* h = nullptr;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);            // the copy the process already uses (e.g. PyTorch's)
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.GetUniqueId = (int (*)(rcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    x.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    x.CommDestroy = (int (*)(rcclComm_t))dlsym(h, "ncclCommDestroy");
    x.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    x.AllGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    x.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(h, "ncclReduceScatter");
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.ReduceScatter;
    return x;
  }();
  return r.ok ? &r : nullptr;
}
constexpr int kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t values of nccl.h
constexpr int PVRL_ECOMM = -3;
}  // namespace

extern "C" int pvrl_comm_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r || !id128) return r ? PVRL_EINVAL : PVRL_ECOMM;
  rcclUniqueId id;
  if (r->GetUniqueId(&id) != 0) return PVRL_ECOMM;
  memcpy(id128, &id, sizeof(id));
  return PVRL_OK;
}

extern "C" int pvrl_comm_init(void** comm, int world, int rank, const void* id128) {
  Rccl* r = rccl();
  if (!r) return PVRL_ECOMM;
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return PVRL_EINVAL;
  rcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  rcclComm_t c = nullptr;
  if (r->CommInitRank(&c, world, id, rank) != 0) return PVRL_ECOMM;
  *comm = c;
  return PVRL_OK;
}

extern "C" int pvrl_comm_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
  Rccl* r = rccl();
  if (!r) return PVRL_ECOMM;
  if (!comm || (!buf && n > 0) || n < 0) return PVRL_EINVAL;
  if (n == 0) return PVRL_OK;
  return r->AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream) == 0 ? PVRL_OK : PVRL_ECOMM;
}

extern "C" int pvrl_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  Rccl* r = rccl();
  if (!r) return PVRL_ECOMM;
  if (!comm || !send || !recv || bytes_per_rank < 0) return PVRL_EINVAL;
  if (bytes_per_rank == 0) return PVRL_OK;
  return r->AllGather(send, recv, (size_t)bytes_per_rank, kNcclInt8, comm, (hipStream_t)stream) == 0 ? PVRL_OK : PVRL_ECOMM;
}

// Sum-reduce `world` equal shards: rank r receives the sum over ranks of send[r * n_per_rank, (r + 1) * n_per_rank) in recv.
// With pvrl_comm_allgather of the reduced shards this is the two-step gradient all-reduce of SURVEY section 5 / 8e for a
// fully-connected xGMI node: every rank talks to its 7 peers directly (payload (W - 1) / W of the buffer per step over 7
// links) instead of pushing the whole buffer around a ring; recv may alias the caller's own shard of send (in place).
extern "C" int pvrl_comm_reducescatter_f32(void* comm, const float* send, float* recv, int64_t n_per_rank, void* stream) {
  Rccl* r = rccl();
  if (!r) return PVRL_ECOMM;
  if (!comm || ((!send || !recv) && n_per_rank > 0) || n_per_rank < 0) return PVRL_EINVAL;
  if (n_per_rank == 0) return PVRL_OK;
  return r->ReduceScatter(send, recv, (size_t)n_per_rank, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream) == 0 ? PVRL_OK : PVRL_ECOMM;
}

extern "C" int pvrl_comm_destroy(void* comm) {
  Rccl* r = rccl();
  if (!r) return PVRL_ECOMM;
  if (!comm) return PVRL_OK;
  return r->CommDestroy(comm) == 0 ? PVRL_OK : PVRL_ECOMM;
}
