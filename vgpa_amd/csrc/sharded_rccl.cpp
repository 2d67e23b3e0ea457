// RCCL behind the vgpa_comm table of the row-sharded recursion (large_d.hip, include/vgpa_hip.h).
// librccl is dlopen'ed on first use -- a copy the process already holds (torch's) is reused -- so that libvgpa_hip.so has
// no link-time dependency on it and single-GPU users never load it.  The unique id travels through the C ABI as 128 bytes.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "vgpa_internal.h"

namespace {

struct UniqueId { char internal[VGPA_RCCL_UNIQUE_ID_BYTES]; };
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*comm_destroy_t)(void*);
typedef int (*comm_abort_t)(void*);
typedef int (*all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*send_t)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*recv_t)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*group_t)();
typedef int (*comm_count_t)(void*, int*);
typedef int (*comm_split_t)(void*, int, int, void**, void*);
constexpr int kNcclFloat64 = 8;          // ncclDataType_t (rccl.h)

struct Api {
  void* so = nullptr;
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  comm_abort_t comm_abort = nullptr;      // optional
  all_gather_t all_gather = nullptr;
  send_t send = nullptr;
  recv_t recv = nullptr;
  group_t group_start = nullptr, group_end = nullptr;
  comm_count_t comm_count = nullptr;      // optional
  comm_split_t comm_split = nullptr;      // optional (ncclCommSplit): the second communicator of the table
  bool tried = false, ok = false;
};
Api g_api;
std::mutex g_mu;

bool load_api() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.tried) return g_api.ok;
  g_api.tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_api.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (g_api.so) break;
  }
  for (const char* n : names) {
    if (g_api.so) break;
    g_api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!g_api.so) return false;
  g_api.get_unique_id = (get_unique_id_t)dlsym(g_api.so, "ncclGetUniqueId");
  g_api.comm_init_rank = (comm_init_rank_t)dlsym(g_api.so, "ncclCommInitRank");
  g_api.comm_destroy = (comm_destroy_t)dlsym(g_api.so, "ncclCommDestroy");
  g_api.comm_abort = (comm_abort_t)dlsym(g_api.so, "ncclCommAbort");
  g_api.all_gather = (all_gather_t)dlsym(g_api.so, "ncclAllGather");
  g_api.send = (send_t)dlsym(g_api.so, "ncclSend");
  g_api.recv = (recv_t)dlsym(g_api.so, "ncclRecv");
  g_api.group_start = (group_t)dlsym(g_api.so, "ncclGroupStart");
  g_api.group_end = (group_t)dlsym(g_api.so, "ncclGroupEnd");
  g_api.comm_count = (comm_count_t)dlsym(g_api.so, "ncclCommCount");
  g_api.comm_split = (comm_split_t)dlsym(g_api.so, "ncclCommSplit");
  g_api.ok = g_api.get_unique_id && g_api.comm_init_rank && g_api.comm_destroy && g_api.all_gather && g_api.send && g_api.recv &&
             g_api.group_start && g_api.group_end;
  return g_api.ok;
}

// TWO communicators per table: the stage's all-to-all / all-gathers run on the shard's compute stream, the point-to-point groups of the
// pipelined gather on its communication stream, concurrently.  RCCL orders the operations of ONE communicator across streams with
// events of its own; a communicator per stream is the documented pattern (comm2 = ncclCommSplit of comm with one colour: same ranks,
// no second unique id to distribute).  Without ncclCommSplit in the library both roles share `comm`.
struct RcclComm {
  void* comm = nullptr;
  void* comm2 = nullptr;       // point-to-point traffic of the communication stream (nullptr: `comm`)
  int rank = 0, world = 1;
  void* p2p() const { return comm2 ? comm2 : comm; }
};

int rccl_all_gather(void* user, const double* send, double* recv, uint64_t count, void* stream) {
  RcclComm* c = static_cast<RcclComm*>(user);
  if (!c->comm) return -1;
  return g_api.all_gather(send, recv, (size_t)count, kNcclFloat64, c->comm, (hipStream_t)stream);
}

// chunk q of send -> rank q: grouped point-to-point calls, one xGMI link per peer
int rccl_all_to_all(void* user, const double* send, double* recv, uint64_t count, void* stream) {
  RcclComm* c = static_cast<RcclComm*>(user);
  if (!c->comm) return -1;
  int rc = g_api.group_start();
  for (int q = 0; q < c->world && rc == 0; q++) {
    rc = g_api.send(send + (size_t)q * count, (size_t)count, kNcclFloat64, q, c->comm, (hipStream_t)stream);
    if (rc == 0) rc = g_api.recv(recv + (size_t)q * count, (size_t)count, kNcclFloat64, q, c->comm, (hipStream_t)stream);
  }
  const int rc2 = g_api.group_end();
  return rc ? rc : rc2;
}

// point-to-point pair of the pipelined gather (always inside rccl_group_begin / rccl_group_end: one launch per sub-block)
int rccl_send(void* user, const double* buf, uint64_t count, int peer, void* stream) {
  RcclComm* c = static_cast<RcclComm*>(user);
  return c->comm ? g_api.send(buf, (size_t)count, kNcclFloat64, peer, c->p2p(), (hipStream_t)stream) : -1;
}
int rccl_recv(void* user, double* buf, uint64_t count, int peer, void* stream) {
  RcclComm* c = static_cast<RcclComm*>(user);
  return c->comm ? g_api.recv(buf, (size_t)count, kNcclFloat64, peer, c->p2p(), (hipStream_t)stream) : -1;
}
// after a failure: ncclCommAbort frees the communicator and fails the operations this rank still has in flight, so that its
// streams drain; the peers see their own collectives fail or time out (vgpa_shard's bounded waits) and abort in turn
int rccl_abort(void* user) {
  RcclComm* c = static_cast<RcclComm*>(user);
  if (!c->comm) return 0;
  void* comm = c->comm;
  void* comm2 = c->comm2;
  c->comm = nullptr; c->comm2 = nullptr;
  int rc2 = 0;
  if (comm2) rc2 = g_api.comm_abort ? g_api.comm_abort(comm2) : g_api.comm_destroy(comm2);
  const int rc = g_api.comm_abort ? g_api.comm_abort(comm) : g_api.comm_destroy(comm);
  return rc ? rc : rc2;
}

int rccl_group_begin(void*) { return g_api.group_start(); }
int rccl_group_end(void*) { return g_api.group_end(); }

}  // namespace

extern "C" {

int vgpa_rccl_unique_id(void* out) {
  if (!out) return VGPA_ERR_ARG;
  if (!load_api()) return VGPA_ERR_UNSUPPORTED;
  UniqueId id;
  if (g_api.get_unique_id(&id) != 0) return VGPA_ERR_DEVICE;
  std::memcpy(out, id.internal, VGPA_RCCL_UNIQUE_ID_BYTES);
  return VGPA_OK;
}

int vgpa_rccl_comm_create(vgpa_comm* out, const void* id_bytes, int rank, int world, int device) {
  if (!out || !id_bytes || world < 1 || rank < 0 || rank >= world) return VGPA_ERR_ARG;
  if (!load_api()) return VGPA_ERR_UNSUPPORTED;
  if (hipSetDevice(device) != hipSuccess) return VGPA_ERR_DEVICE;
  UniqueId id;
  std::memcpy(id.internal, id_bytes, VGPA_RCCL_UNIQUE_ID_BYTES);
  RcclComm* c = new RcclComm();
  c->rank = rank; c->world = world;
  if (g_api.comm_init_rank(&c->comm, world, id, rank) != 0) { delete c; return VGPA_ERR_DEVICE; }
  // (collective: every rank of the table splits; a failure leaves the one communicator for both streams)
  if (g_api.comm_split && g_api.comm_split(c->comm, 0, rank, &c->comm2, nullptr) != 0) c->comm2 = nullptr;
  out->user = c;
  out->all_gather = rccl_all_gather;
  out->all_to_all = rccl_all_to_all;
  out->group_begin = rccl_group_begin;
  out->group_end = rccl_group_end;
  out->send = rccl_send;
  out->recv = rccl_recv;
  out->abort = rccl_abort;
  return VGPA_OK;
}

// ranks of the communicator as the LIBRARY counts them (ncclCommCount) -- what bench.py reports as `rccl_ranks`
int vgpa_rccl_comm_count(const vgpa_comm* comm, int* count) {
  if (!comm || !comm->user || !count) return VGPA_ERR_ARG;
  if (comm->all_gather != rccl_all_gather) return VGPA_ERR_ARG;       // not a table vgpa_rccl_comm_create filled
  RcclComm* c = static_cast<RcclComm*>(comm->user);
  if (!c->comm) return VGPA_ERR_COMM;
  if (!g_api.comm_count) return VGPA_ERR_UNSUPPORTED;
  return g_api.comm_count(c->comm, count) == 0 ? VGPA_OK : VGPA_ERR_DEVICE;
}

// communicators behind the table: 2 (one per stream of the shard), or 1 when librccl has no ncclCommSplit
int vgpa_rccl_comm_streams(const vgpa_comm* comm, int* count) {
  if (!comm || !comm->user || !count) return VGPA_ERR_ARG;
  if (comm->all_gather != rccl_all_gather) return VGPA_ERR_ARG;
  RcclComm* c = static_cast<RcclComm*>(comm->user);
  if (!c->comm) return VGPA_ERR_COMM;
  *count = c->comm2 ? 2 : 1;
  return VGPA_OK;
}

void vgpa_rccl_comm_destroy(vgpa_comm* comm) {
  if (!comm || !comm->user) return;
  RcclComm* c = static_cast<RcclComm*>(comm->user);
  if (c->comm2) (void)g_api.comm_destroy(c->comm2);
  if (c->comm) (void)g_api.comm_destroy(c->comm);
  delete c;
  comm->user = nullptr;
}

}  // extern "C"
