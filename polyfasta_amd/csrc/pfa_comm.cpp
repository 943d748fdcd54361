// pfa_comm.cpp — the multi-GPU side of libpfa_hip.so: one process per GPU, RCCL over xGMI.
//
// The scan shards into independent alignments (the reference's file loop, PolyFastA.py:104), so
// no collective touches the data path.  What travels is the final gather of the fixed-size
// integer records (pfa_result) plus a barrier and a max-reduction for timing.  RCCL is bound
// at run time (dlopen of librccl.so.1 on the first pfa_comm_* call): a single-GPU process never
// loads it.  The 128-byte unique id is made by rank 0 and carried to the other ranks by the
// host language (polyfasta_amd/dist.py: a file next to the launcher's rendezvous).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "pfa_common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
  Rccl api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  uint8_t* d_stage = nullptr;   // device staging of host-buffer collectives, grown on demand
  size_t stage_bytes = 0;
  uint8_t* d_small = nullptr;   // 64 bytes: barrier word, reduction operands
  bool ready = false;
};

Comm g_comm;

int bind_rccl(Rccl& r) {
  if (r.handle) return PFA_OK;
  const char* names[] = {getenv("PFA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    pfa::set_error("cannot load librccl (%s): multi-GPU runs need RCCL", dlerror());
    return PFA_ERR_NO_DEVICE;
  }
#define PFA_SYM(field, name)                                                   \
  do {                                                                         \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));      \
    if (!r.field) {                                                            \
      pfa::set_error("librccl lacks %s", name);                                \
      dlclose(r.handle);                                                       \
      r = Rccl();                                                              \
      return PFA_ERR_NO_DEVICE;                                                \
    }                                                                          \
  } while (0)
  PFA_SYM(GetUniqueId, "ncclGetUniqueId");
  PFA_SYM(CommInitRank, "ncclCommInitRank");
  PFA_SYM(CommDestroy, "ncclCommDestroy");
  PFA_SYM(AllGather, "ncclAllGather");
  PFA_SYM(AllReduce, "ncclAllReduce");
  PFA_SYM(Send, "ncclSend");
  PFA_SYM(Recv, "ncclRecv");
  PFA_SYM(GroupStart, "ncclGroupStart");
  PFA_SYM(GroupEnd, "ncclGroupEnd");
  PFA_SYM(GetErrorString, "ncclGetErrorString");
#undef PFA_SYM
  return PFA_OK;
}

#define PFA_NCCL(call)                                                                            \
  do {                                                                                            \
    ncclResult_t r__ = (call);                                                                    \
    if (r__ != ncclSuccess) {                                                                     \
      pfa::set_error("%s failed: %s (%s:%d)", #call, g_comm.api.GetErrorString(r__), __FILE__, __LINE__); \
      return PFA_ERR_HIP;                                                                         \
    }                                                                                             \
  } while (0)

#define PFA_NEED_COMM()                                                          \
  do {                                                                           \
    PFA_NEED_INIT();                                                             \
    if (!g_comm.ready) {                                                         \
      pfa::set_error("pfa_comm_init() has not been called successfully");       \
      return PFA_ERR_NOT_INIT;                                                   \
    }                                                                            \
  } while (0)

int need_stage(size_t bytes) {
  if (g_comm.stage_bytes >= bytes) return PFA_OK;
  if (g_comm.d_stage) PFA_HIP(hipFree(g_comm.d_stage));
  g_comm.d_stage = nullptr;
  g_comm.stage_bytes = 0;
  const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
  PFA_HIP(hipMalloc((void**)&g_comm.d_stage, want));
  g_comm.stage_bytes = want;
  return PFA_OK;
}

}  // namespace

extern "C" {

int pfa_comm_unique_id(uint8_t* id_out) {
  if (!id_out) { pfa::set_error("pfa_comm_unique_id: id_out is NULL"); return PFA_ERR_ARG; }
  static_assert(sizeof(ncclUniqueId) == PFA_COMM_ID_BYTES, "unique id size");
  int rc = bind_rccl(g_comm.api);
  if (rc != PFA_OK) return rc;
  ncclUniqueId id;
  PFA_NCCL(g_comm.api.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PFA_OK;
}

int pfa_comm_init(int rank, int world, const uint8_t* id) {
  PFA_NEED_INIT();
  if (!id || world < 1 || rank < 0 || rank >= world) { pfa::set_error("pfa_comm_init: bad rank %d / world %d / id", rank, world); return PFA_ERR_ARG; }
  if (g_comm.ready) { pfa::set_error("pfa_comm_init: a communicator exists already"); return PFA_ERR_ARG; }
  int rc = bind_rccl(g_comm.api);
  if (rc != PFA_OK) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  PFA_HIP(hipSetDevice(pfa::ctx().device));
  PFA_NCCL(g_comm.api.CommInitRank(&g_comm.comm, world, uid, rank));
  PFA_HIP(hipMalloc((void**)&g_comm.d_small, 64));
  PFA_HIP(hipMemset(g_comm.d_small, 0, 64));
  g_comm.rank = rank;
  g_comm.world = world;
  g_comm.ready = true;
  return PFA_OK;
}

int pfa_comm_rank(void) { return g_comm.ready ? g_comm.rank : 0; }
int pfa_comm_world(void) { return g_comm.ready ? g_comm.world : 1; }

int pfa_comm_destroy(void) {
  if (!g_comm.ready) return PFA_OK;
  if (pfa::ctx().ready) (void)hipStreamSynchronize(pfa::ctx().stream);
  (void)g_comm.api.CommDestroy(g_comm.comm);
  (void)hipFree(g_comm.d_stage);
  (void)hipFree(g_comm.d_small);
  const Rccl keep = g_comm.api;   // the library stays loaded
  g_comm = Comm();
  g_comm.api = keep;
  return PFA_OK;
}

int pfa_comm_barrier(void) {
  PFA_NEED_COMM();
  hipStream_t st = pfa::ctx().stream;
  PFA_NCCL(g_comm.api.AllReduce(g_comm.d_small, g_comm.d_small + 32, 1, ncclInt32, ncclSum, g_comm.comm, st));
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_comm_all_reduce_max(double* x) {
  PFA_NEED_COMM();
  if (!x) { pfa::set_error("pfa_comm_all_reduce_max: x is NULL"); return PFA_ERR_ARG; }
  hipStream_t st = pfa::ctx().stream;
  PFA_HIP(hipMemcpyAsync(g_comm.d_small + 8, x, 8, hipMemcpyHostToDevice, st));
  PFA_NCCL(g_comm.api.AllReduce(g_comm.d_small + 8, g_comm.d_small + 16, 1, ncclFloat64, ncclMax, g_comm.comm, st));
  PFA_HIP(hipMemcpyAsync(x, g_comm.d_small + 16, 8, hipMemcpyDeviceToHost, st));
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_comm_all_gather(const void* send_host, void* recv_host, size_t bytes_per_rank) {
  PFA_NEED_COMM();
  if (bytes_per_rank == 0) return PFA_OK;
  if (!send_host || !recv_host) { pfa::set_error("pfa_comm_all_gather: NULL buffer"); return PFA_ERR_ARG; }
  const size_t w = (size_t)g_comm.world;
  const size_t send_span = (bytes_per_rank + 255u) & ~size_t(255);   // the receive buffer starts on a 256-byte boundary
  int rc = need_stage(send_span + w * bytes_per_rank);
  if (rc != PFA_OK) return rc;
  hipStream_t st = pfa::ctx().stream;
  uint8_t* d_send = g_comm.d_stage;
  uint8_t* d_recv = g_comm.d_stage + send_span;
  PFA_HIP(hipMemcpyAsync(d_send, send_host, bytes_per_rank, hipMemcpyHostToDevice, st));
  PFA_NCCL(g_comm.api.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, g_comm.comm, st));
  PFA_HIP(hipMemcpyAsync(recv_host, d_recv, w * bytes_per_rank, hipMemcpyDeviceToHost, st));
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_comm_gatherv(const void* send_host, uint64_t send_bytes, void* recv_host, const uint64_t* counts, int root) {
  PFA_NEED_COMM();
  const int w = g_comm.world, me = g_comm.rank;
  if (!counts || root < 0 || root >= w || (send_bytes && !send_host)) { pfa::set_error("pfa_comm_gatherv: bad argument"); return PFA_ERR_ARG; }
  if (counts[me] != send_bytes) { pfa::set_error("pfa_comm_gatherv: counts[%d] = %llu but %llu bytes are sent", me, (unsigned long long)counts[me], (unsigned long long)send_bytes); return PFA_ERR_ARG; }
  uint64_t total = 0;
  for (int r = 0; r < w; ++r) total += counts[r];
  if (me == root && total && !recv_host) { pfa::set_error("pfa_comm_gatherv: recv_host is NULL on the root"); return PFA_ERR_ARG; }
  // staging: [ my bytes | (root) everybody's bytes, rank after rank ], both on 256-byte boundaries
  const size_t send_span = ((size_t)send_bytes + 255u) & ~size_t(255);
  int rc = need_stage(send_span + (me == root ? (size_t)total : 0) + 256u);
  if (rc != PFA_OK) return rc;
  hipStream_t st = pfa::ctx().stream;
  uint8_t* const d_send = g_comm.d_stage;
  uint8_t* const d_recv = g_comm.d_stage + send_span;
  if (send_bytes) PFA_HIP(hipMemcpyAsync(d_send, send_host, (size_t)send_bytes, hipMemcpyHostToDevice, st));
  PFA_NCCL(g_comm.api.GroupStart());
  // inside the group an error does not return: the group is closed first (a library must not leave RCCL with an open group)
  ncclResult_t first_err = ncclSuccess;
  const char* what = "";
  if (me == root) {
    uint64_t off = 0;
    for (int r = 0; r < w && first_err == ncclSuccess; ++r) {
      if (counts[r] && r != me) {
        first_err = g_comm.api.Recv(d_recv + off, (size_t)counts[r], ncclUint8, r, g_comm.comm, st);
        what = "ncclRecv";
      }
      off += counts[r];
    }
  } else if (send_bytes) {
    first_err = g_comm.api.Send(d_send, (size_t)send_bytes, ncclUint8, root, g_comm.comm, st);
    what = "ncclSend";
  }
  const ncclResult_t end_err = g_comm.api.GroupEnd();
  if (first_err != ncclSuccess || end_err != ncclSuccess) {
    const bool inner = first_err != ncclSuccess;
    pfa::set_error("pfa_comm_gatherv: %s failed: %s", inner ? what : "ncclGroupEnd", g_comm.api.GetErrorString(inner ? first_err : end_err));
    return PFA_ERR_HIP;
  }
  if (me == root) {
    uint64_t off = 0;
    for (int r = 0; r < me; ++r) off += counts[r];
    if (send_bytes) PFA_HIP(hipMemcpyAsync(d_recv + off, d_send, (size_t)send_bytes, hipMemcpyDeviceToDevice, st));   // the root's own part
    if (total) PFA_HIP(hipMemcpyAsync(recv_host, d_recv, (size_t)total, hipMemcpyDeviceToHost, st));
  }
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

int pfa_comm_all_gather_results(pfa_plan* plan, pfa_result* out) {
  PFA_NEED_COMM();
  if (!plan || !out) { pfa::set_error("pfa_comm_all_gather_results: NULL argument"); return PFA_ERR_ARG; }
  void* d_res = nullptr;
  size_t bytes = 0;
  int rc = pfa_plan_results_dev(plan, &d_res, &bytes);
  if (rc != PFA_OK) return rc;
  const size_t w = (size_t)g_comm.world;
  rc = need_stage(w * bytes);
  if (rc != PFA_OK) return rc;
  hipStream_t st = pfa::ctx().stream;   // the scan launches of the plan are ahead on the same stream
  PFA_NCCL(g_comm.api.AllGather(d_res, g_comm.d_stage, bytes, ncclUint8, g_comm.comm, st));
  PFA_HIP(hipMemcpyAsync(out, g_comm.d_stage, w * bytes, hipMemcpyDeviceToHost, st));
  PFA_HIP(hipStreamSynchronize(st));
  return PFA_OK;
}

}  // extern "C"
