// R-layer of the C ABI: the collectives of the row-sharded step over RCCL.
// The reference has no communication of any kind; these are the exchanges sharded.py performs,
// exposed without torch types so that any binder of tt_hotpath.h has the multi-GPU path:
//   all-to-all   (routed lookups: ids, rows, row gradients)          R1
//   all-gather   (item embeddings), reduce-scatter (partial dI)      R2
//   all-reduce   (dense gradients; MAX of value weights / bucket sizes; SUM of the loss)   R3
// RCCL is bound at run time (dlopen of librccl.so.1 -- inside a torch process that is the copy torch
// already loaded), so libtt_hotpath.so keeps loading on a box without RCCL and carries no link-time
// dependency.  The only long-lived native state of the library is the communicator handle.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: nothing here links against RCCL
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

namespace tt {
namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
char g_load_error[384] = "";

template <typename F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  if (!fn) snprintf(g_load_error, sizeof(g_load_error), "RCCL does not export %s", name);
  return fn != nullptr;
}

void load_rccl() {
  const char* names[] = {getenv("TT_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
    snprintf(g_load_error, sizeof(g_load_error), "dlopen(%s): %s", n, dlerror());
  }
  if (!g_rccl.handle) return;
  Rccl& r = g_rccl;
  const bool ok = bind(r.GetUniqueId, "ncclGetUniqueId") && bind(r.CommInitRank, "ncclCommInitRank") &&
                  bind(r.CommDestroy, "ncclCommDestroy") && bind(r.CommCount, "ncclCommCount") &&
                  bind(r.GetErrorString, "ncclGetErrorString") && bind(r.AllGather, "ncclAllGather") &&
                  bind(r.ReduceScatter, "ncclReduceScatter") && bind(r.AllReduce, "ncclAllReduce") &&
                  bind(r.AllToAll, "ncclAllToAll") && bind(r.Broadcast, "ncclBroadcast");
  if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
}

int need_rccl(const char* who) {
  std::call_once(g_once, load_rccl);
  if (g_rccl.handle) return 0;
  set_error("%s: RCCL is not available (%s)", who, g_load_error);
  return TT_E_UNSUPPORTED;
}

// ncclResult_t -> the ABI's return convention: 0 ok, otherwise -(100 + code) so it cannot be mistaken
// for a hipError_t (> 0) or a TT_E_* (-1 .. -3)
int rc_of(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return 0;
  set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return -(100 + (int)r);
}

struct Comm {
  ncclComm_t comm;
  int rank, world;
};

bool dtype_of(int dtype, ncclDataType_t& out, int64_t& size) {
  switch (dtype) {
    case TT_COMM_F32: out = ncclFloat32; size = 4; return true;
    case TT_COMM_I32: out = ncclInt32; size = 4; return true;
    case TT_COMM_I64: out = ncclInt64; size = 8; return true;
    case TT_COMM_U8: out = ncclUint8; size = 1; return true;
  }
  return false;
}

}  // namespace
}  // namespace tt

using namespace tt;

extern "C" int tt_comm_unique_id(void* id_out) {
  if (!id_out) return fail_arg("tt_comm_unique_id: null pointer");
  int rc = need_rccl("tt_comm_unique_id");
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == TT_COMM_ID_BYTES, "TT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  ncclUniqueId id;
  if ((rc = rc_of(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"))) return rc;
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int tt_comm_init(const void* id, int32_t rank, int32_t world, tt_comm_t* out) {
  if (!id || !out) return fail_arg("tt_comm_init: null pointer");
  if (world <= 0 || rank < 0 || rank >= world) return fail_arg("tt_comm_init: rank / world");
  int rc = need_rccl("tt_comm_init");
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  Comm* c = new Comm{nullptr, rank, world};
  if ((rc = rc_of(g_rccl.CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank"))) { delete c; return rc; }
  *out = reinterpret_cast<tt_comm_t>(c);
  return 0;
}

extern "C" int tt_comm_destroy(tt_comm_t comm) {
  if (!comm) return 0;
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int rc = rc_of(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

extern "C" int tt_comm_size(tt_comm_t comm, int32_t* rank_out, int32_t* world_out) {
  if (!comm || !rank_out || !world_out) return fail_arg("tt_comm_size: null pointer");
  Comm* c = reinterpret_cast<Comm*>(comm);
  int n = 0;
  const int rc = rc_of(g_rccl.CommCount(c->comm, &n), "ncclCommCount");  // what RCCL itself reports
  if (rc) return rc;
  *rank_out = c->rank;
  *world_out = n;
  return 0;
}

extern "C" int tt_comm_allgather(tt_comm_t comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                                 tt_stream_t stream) {
  if (!comm || (count_per_rank != 0 && (!send || !recv))) return fail_arg("tt_comm_allgather: null pointer");
  ncclDataType_t dt;
  int64_t sz;
  if (count_per_rank < 0 || !dtype_of(dtype, dt, sz)) return fail_arg("tt_comm_allgather: count / dtype");
  if (count_per_rank == 0) return 0;  // an empty exchange is a no-op, as in torch.distributed
  Comm* c = reinterpret_cast<Comm*>(comm);
  return rc_of(g_rccl.AllGather(send, recv, (size_t)count_per_rank, dt, c->comm, S(stream)), "ncclAllGather");
}

extern "C" int tt_comm_reduce_scatter(tt_comm_t comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                                      int op, tt_stream_t stream) {
  if (!comm || (count_per_rank != 0 && (!send || !recv))) return fail_arg("tt_comm_reduce_scatter: null pointer");
  ncclDataType_t dt;
  int64_t sz;
  if (count_per_rank < 0 || !dtype_of(dtype, dt, sz) || (op != TT_COMM_SUM && op != TT_COMM_MAX))
    return fail_arg("tt_comm_reduce_scatter: count / dtype / op");
  if (count_per_rank == 0) return 0;
  Comm* c = reinterpret_cast<Comm*>(comm);
  return rc_of(g_rccl.ReduceScatter(send, recv, (size_t)count_per_rank, dt, op == TT_COMM_MAX ? ncclMax : ncclSum, c->comm,
                                    S(stream)), "ncclReduceScatter");
}

extern "C" int tt_comm_allreduce(tt_comm_t comm, const void* send, void* recv, int64_t count, int dtype, int op,
                                 tt_stream_t stream) {
  if (!comm || (count != 0 && (!send || !recv))) return fail_arg("tt_comm_allreduce: null pointer");
  ncclDataType_t dt;
  int64_t sz;
  if (count == 0 && dtype_of(dtype, dt, sz) && (op == TT_COMM_SUM || op == TT_COMM_MAX)) return 0;
  if (count <= 0 || !dtype_of(dtype, dt, sz) || (op != TT_COMM_SUM && op != TT_COMM_MAX))
    return fail_arg("tt_comm_allreduce: count / dtype / op");
  Comm* c = reinterpret_cast<Comm*>(comm);
  return rc_of(g_rccl.AllReduce(send, recv, (size_t)count, dt, op == TT_COMM_MAX ? ncclMax : ncclSum, c->comm, S(stream)),
               "ncclAllReduce");
}

extern "C" int tt_comm_alltoall(tt_comm_t comm, const void* send, void* recv, int64_t count_per_peer, int dtype,
                                tt_stream_t stream) {
  if (!comm || (count_per_peer != 0 && (!send || !recv))) return fail_arg("tt_comm_alltoall: null pointer");
  if (send == recv) return fail_arg("tt_comm_alltoall: in-place exchange is not supported");
  ncclDataType_t dt;
  int64_t sz;
  if (count_per_peer < 0 || !dtype_of(dtype, dt, sz)) return fail_arg("tt_comm_alltoall: count / dtype");
  if (count_per_peer == 0) return 0;
  Comm* c = reinterpret_cast<Comm*>(comm);
  return rc_of(g_rccl.AllToAll(send, recv, (size_t)count_per_peer, dt, c->comm, S(stream)), "ncclAllToAll");
}

extern "C" int tt_comm_broadcast(tt_comm_t comm, void* buf, int64_t count, int dtype, int32_t root, tt_stream_t stream) {
  if (!comm || (count != 0 && !buf)) return fail_arg("tt_comm_broadcast: null pointer");
  ncclDataType_t dt;
  int64_t sz;
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (count < 0 || !dtype_of(dtype, dt, sz) || root < 0 || root >= c->world) return fail_arg("tt_comm_broadcast: count / dtype / root");
  if (count == 0) return 0;
  return rc_of(g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, c->comm, S(stream)), "ncclBroadcast");
}
