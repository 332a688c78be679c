// RCCL behind the C ABI: dl3p_comm_{unique_id,init,allreduce,syncbn_allreduce,destroy}, for hosts
// that are not Python.  One process per GPU (reference: train.py:143-158, tf.distribute.MirroredStrategy -- gradient
// all-reduce + SyncBatchNormalization's statistics all-reduce, both sums).  The Python facade keeps torch.distributed
// (backend 'nccl' IS RCCL) because its collectives are captured into the step's hipGraph next to the kernels; these
// entry points give a C / C++ / Go host the same two reductions on the same flat buffers:
//   gradient buckets: contiguous fp32 slices of the flat gradient buffer, in place;
//   SyncBN:           the fp64 (sum, sum^2) / (sum g', sum g' xhat) staging vector, in place.
// librccl is bound at run time (dlopen) so that libdl3p.so itself has no link dependency on it: a process that already
// carries an RCCL (torch's bundled copy has the same soname) gets that one.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
struct UniqueId { char internal[128]; };     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                          // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
const int kFloat32 = 7, kFloat64 = 8, kSum = 0;     // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {getenv("DL3P_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (r.handle) {
      r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
      r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
      r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
      r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
      r.error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
    }
  }
  const bool ok = r.handle && r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy;
  return ok ? &r : nullptr;
}

struct Dl3pComm { Comm comm; int rank, world; };

int check_rc(const char* who, int rc) {
  if (rc == 0) return DL3P_OK;
  Rccl* r = rccl();
  dl3p_set_error("%s: RCCL error %d (%s)", who, rc, (r && r->error_string) ? r->error_string(rc) : "?");
  return DL3P_ELAUNCH;
}
}  // namespace

extern "C" int dl3p_comm_unique_id(void* id128) {
  DL3P_CHECK_ARG(id128 != nullptr, "dl3p_comm_unique_id: null buffer");
  Rccl* r = rccl();
  DL3P_CHECK_ARG(r != nullptr, "dl3p_comm_unique_id: librccl not found (set DL3P_RCCL_LIB)");
  UniqueId id;
  int rc = check_rc("dl3p_comm_unique_id", r->get_unique_id(&id));
  if (rc) return rc;
  memcpy(id128, &id, sizeof(id));
  return DL3P_OK;
}

extern "C" int dl3p_comm_init(void** comm_out, int rank, int world_size, const void* id128) {
  DL3P_CHECK_ARG(comm_out && id128 && world_size > 0 && rank >= 0 && rank < world_size, "dl3p_comm_init: bad arguments");
  Rccl* r = rccl();
  DL3P_CHECK_ARG(r != nullptr, "dl3p_comm_init: librccl not found (set DL3P_RCCL_LIB)");
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  int rc = check_rc("dl3p_comm_init", r->comm_init_rank(&c, world_size, id, rank));
  if (rc) return rc;
  Dl3pComm* h = new Dl3pComm{c, rank, world_size};
  *comm_out = h;
  return DL3P_OK;
}

static int comm_allreduce(const char* who, void* comm, void* buf, size_t count, int dtype, void* stream) {
  DL3P_CHECK_ARG(comm && buf && count > 0, "%s: bad arguments", who);
  Rccl* r = rccl();
  DL3P_CHECK_ARG(r != nullptr, "%s: librccl not found", who);
  Dl3pComm* h = static_cast<Dl3pComm*>(comm);
  return check_rc(who, r->all_reduce(buf, buf, count, dtype, kSum, h->comm, (hipStream_t)stream));
}

extern "C" int dl3p_comm_allreduce(void* comm, float* buf, size_t count, void* stream) {
  return comm_allreduce("dl3p_comm_allreduce", comm, buf, count, kFloat32, stream);
}

extern "C" int dl3p_comm_syncbn_allreduce(void* comm, double* sums, size_t count, void* stream) {
  return comm_allreduce("dl3p_comm_syncbn_allreduce", comm, sums, count, kFloat64, stream);
}

extern "C" int dl3p_comm_destroy(void* comm) {
  DL3P_CHECK_ARG(comm != nullptr, "dl3p_comm_destroy: null communicator");
  Rccl* r = rccl();
  DL3P_CHECK_ARG(r != nullptr, "dl3p_comm_destroy: librccl not found");
  Dl3pComm* h = static_cast<Dl3pComm*>(comm);
  int rc = check_rc("dl3p_comm_destroy", r->comm_destroy(h->comm));
  delete h;
  return rc;
}
