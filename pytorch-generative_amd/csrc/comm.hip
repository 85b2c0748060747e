// comm.hip — the exchange step of data-parallel training: ONE RCCL all-reduce of the flat gradient
// buffer over xGMI, as an ordinary stream operation of the C-ABI (SURVEY.md §8(b) `pg_comm_*`, §8(e)).
//
// Replaces DistributedDataParallel's bucketed all-reduce hooks (reference trainer.py:78-82, process
// group set up in train.py:27-37). Because the collective is enqueued on the stream it is given —
// no host synchronisation, no allocation — it can be captured INSIDE the hipGraph of the training
// step, between the backward kernels and the norm / Adam kernels: one graph launch per step instead
// of two graphs around an eager collective (graph.py).
//
// librccl is bound at run time with dlopen / dlsym: a process that already carries an RCCL (PyTorch
// loads its own copy of librccl.so.1 with libtorch_hip) keeps exactly one copy, and a single-GPU
// process that never calls pg_comm_init never touches RCCL at all. The communicator is
// per-process state (one process per GPU, as train.py:43 spawns them): created once by pg_comm_init,
// used from whichever thread enqueues the step, destroyed by pg_comm_destroy; the entry points are
// serialised by a mutex, the collective itself is asynchronous.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* origin = "";
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0;

// g_mu held
int bind_rccl() {
  if (g_rccl.handle) return 0;
  static const char* const names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  const char* origin = "";
  for (const char* n : {names[0], names[1]}) {  // a copy the process already carries (PyTorch's)
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) { origin = "already loaded"; break; }
  }
  for (int i = 0; !h && i < 3; ++i) {
    h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (h) origin = names[i];
  }
  PG_REQUIRE(h, PG_EINVAL, "pg_comm: librccl not found (%s)", dlerror());
#define PG_SYM(field, name)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));         \
  PG_REQUIRE(g_rccl.field, PG_EINVAL, "pg_comm: librccl lacks %s", name)
  PG_SYM(GetUniqueId, "ncclGetUniqueId");
  PG_SYM(CommInitRank, "ncclCommInitRank");
  PG_SYM(CommDestroy, "ncclCommDestroy");
  PG_SYM(AllReduce, "ncclAllReduce");
  PG_SYM(Broadcast, "ncclBroadcast");
  PG_SYM(GetErrorString, "ncclGetErrorString");
  PG_SYM(GetVersion, "ncclGetVersion");
#undef PG_SYM
  g_rccl.origin = origin;
  g_rccl.handle = h;
  return 0;
}

#define PG_NCCL(call, what)                                                                \
  do {                                                                                     \
    const ncclResult_t r__ = (call);                                                       \
    if (r__ != ncclSuccess) {                                                              \
      pg_set_error("%s: RCCL error %d (%s)", what, (int)r__, g_rccl.GetErrorString(r__));  \
      return 1000 + (int)r__;                                                              \
    }                                                                                      \
  } while (0)

int nccl_dtype(int dtype, ncclDataType_t& t) {
  switch (dtype) {
    case PG_DTYPE_F32: t = ncclFloat32; return 0;
    default: pg_set_error("pg_comm: dtype %d not supported (fp32 only)", dtype); return PG_EINVAL;
  }
}

}  // namespace

PG_EXPORT int pg_comm_unique_id(char id[PG_COMM_ID_BYTES]) {
  std::lock_guard<std::mutex> lk(g_mu);
  PG_REQUIRE(id, PG_EINVAL, "pg_comm_unique_id: null id");
  if (int rc = bind_rccl()) return rc;
  static_assert(sizeof(ncclUniqueId) == PG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  PG_NCCL(g_rccl.GetUniqueId(&u), "pg_comm_unique_id");
  memcpy(id, u.internal, PG_COMM_ID_BYTES);
  return 0;
}

PG_EXPORT int pg_comm_init(int rank, int world, const char id[PG_COMM_ID_BYTES]) {
  std::lock_guard<std::mutex> lk(g_mu);
  PG_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, PG_EINVAL, "pg_comm_init: bad rank %d / world %d", rank, world);
  PG_REQUIRE(!g_comm, PG_EINVAL, "pg_comm_init: a communicator already exists (pg_comm_destroy first)");
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId u;
  memcpy(u.internal, id, PG_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;  // published only on success: a refused id must leave no communicator behind
  PG_NCCL(g_rccl.CommInitRank(&comm, world, u, rank), "pg_comm_init");  // on the calling thread's current device
  g_comm = comm;
  g_rank = rank;
  g_world = world;
  return 0;
}

PG_EXPORT int pg_comm_world(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_comm ? g_world : 0;
}

PG_EXPORT int pg_comm_rccl_version(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (bind_rccl()) return 0;
  int v = 0;
  return g_rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

PG_EXPORT int pg_allreduce_sum(void* buf, size_t n, int dtype, void* stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  PG_REQUIRE(g_comm, PG_EINVAL, "pg_allreduce_sum: no communicator (pg_comm_init first)");
  PG_REQUIRE(buf || n == 0, PG_EINVAL, "pg_allreduce_sum: null buffer");
  ncclDataType_t t;
  if (int rc = nccl_dtype(dtype, t)) return rc;
  if (n == 0) return 0;
  PG_NCCL(g_rccl.AllReduce(buf, buf, n, t, ncclSum, g_comm, (hipStream_t)stream), "pg_allreduce_sum");
  return 0;
}

PG_EXPORT int pg_broadcast(void* buf, size_t n, int dtype, int root, void* stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  PG_REQUIRE(g_comm, PG_EINVAL, "pg_broadcast: no communicator (pg_comm_init first)");
  PG_REQUIRE(root >= 0 && root < g_world, PG_EINVAL, "pg_broadcast: root %d outside world %d", root, g_world);
  ncclDataType_t t;
  if (int rc = nccl_dtype(dtype, t)) return rc;
  if (n == 0) return 0;
  PG_NCCL(g_rccl.Broadcast(buf, buf, n, t, root, g_comm, (hipStream_t)stream), "pg_broadcast");
  return 0;
}

PG_EXPORT int pg_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm) return 0;
  const ncclResult_t r = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  if (r != ncclSuccess) {
    pg_set_error("pg_comm_destroy: RCCL error %d (%s)", (int)r, g_rccl.GetErrorString(r));
    return 1000 + (int)r;
  }
  return 0;
}
