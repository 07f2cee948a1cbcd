// dcrx_rccl.cpp — the multi-GPU entries of the C ABI (include/dcrx.h, "multi-GPU"): RCCL over xGMI bound directly, no
// framework in between.  The reference is single-process and has nothing here (README.md:370-376: "submit many jobs"); what is
// built is SURVEY.md 8(e): reads sharded in contiguous ranges, one exchange at the end of a step — the ranks' counts
// (ncclAllGather), their tuple messages to rank 0 in exact sizes (grouped ncclSend / ncclRecv), the counters summed
// (ncclAllReduce of uint64[32]).
//
// librccl is opened when the first communicator is asked for (dlopen; "librccl.so.1" by soname, so that a process that already
// holds an RCCL — PyTorch's — shares it, with the HIP runtime that came with it): a single-GPU caller never loads it, and
// libdcrx.so itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dcrx.h"

namespace dcrx {
int set_err(int code, const char *msg);
}
using dcrx::set_err;

namespace {

struct Rccl {
  void *handle = nullptr;
  std::string why;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

Rccl *rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (R.handle) break;
      R.why = dlerror() ? dlerror() : "dlopen failed";
    }
    if (!R.handle) return;
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
      void *p = dlsym(R.handle, name);
      if (!p) { ok = false; R.why = std::string("librccl has no ") + name; }
      return p;
    };
#define DCRX_RCCL_SYM(field, name) R.field = reinterpret_cast<decltype(R.field)>(sym(name))
    DCRX_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); DCRX_RCCL_SYM(CommInitRank, "ncclCommInitRank"); DCRX_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    DCRX_RCCL_SYM(CommDestroy, "ncclCommDestroy"); DCRX_RCCL_SYM(AllReduce, "ncclAllReduce"); DCRX_RCCL_SYM(AllGather, "ncclAllGather");
    DCRX_RCCL_SYM(Send, "ncclSend"); DCRX_RCCL_SYM(Recv, "ncclRecv"); DCRX_RCCL_SYM(GroupStart, "ncclGroupStart"); DCRX_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    DCRX_RCCL_SYM(GetErrorString, "ncclGetErrorString"); DCRX_RCCL_SYM(GetVersion, "ncclGetVersion");
#undef DCRX_RCCL_SYM
    if (!ok) { dlclose(R.handle); R.handle = nullptr; }
  });
  return R.handle ? &R : nullptr;
}

int rccl_missing() {
  static std::string msg;
  Rccl *r = rccl();
  (void)r;
  msg = "RCCL is not available (librccl.so.1 could not be opened, or lacks an entry this library binds)";
  return set_err(DCRX_E_UNSUPPORTED, msg.c_str());
}

int nccl_err(Rccl *R, ncclResult_t e, const char *what) {
  std::string m = std::string(what) + ": " + (R && R->GetErrorString ? R->GetErrorString(e) : "RCCL error") + " (" + std::to_string((int)e) + ")";
  return set_err(DCRX_E_HIP, m.c_str());
}
int hip_fail(hipError_t e, const char *what) {
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return set_err(DCRX_E_HIP, m.c_str());
}
#define NCCL_TRY(R, call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return nccl_err(R, e_, #call); } while (0)
#define HIP_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_, #call); } while (0)

}  // namespace

struct dcrx_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
  uint64_t *d_scratch = nullptr;      // [2 + world]: a count, a barrier word, the ranks' counts
  uint64_t *h_scratch = nullptr;      // pinned, [2 + world]
};

namespace {
int finish_create(Rccl *R, dcrx_comm *c) {
  HIP_TRY(hipMalloc(&c->d_scratch, (size_t)(2 + c->world) * 8));
  HIP_TRY(hipMemset(c->d_scratch, 0, (size_t)(2 + c->world) * 8));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_scratch), (size_t)(2 + c->world) * 8, hipHostMallocDefault));
  (void)R;
  return DCRX_OK;
}
}  // namespace

extern "C" {

int dcrx_comm_available(void) { return rccl() ? 1 : 0; }

int dcrx_comm_unique_id(uint8_t *id) {
  if (!id) return set_err(DCRX_E_INVALID, "null id");
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  static_assert(DCRX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as RCCL made it");
  ncclUniqueId u;
  NCCL_TRY(R, R->GetUniqueId(&u));
  std::memcpy(id, u.internal, DCRX_COMM_ID_BYTES);
  return DCRX_OK;
}

int dcrx_comm_create(const uint8_t *id, int world, int rank, dcrx_comm_t **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return set_err(DCRX_E_INVALID, "dcrx_comm_create: bad argument");
  *out = nullptr;
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  dcrx_comm *c = new (std::nothrow) dcrx_comm;
  if (!c) return set_err(DCRX_E_NOMEM, "dcrx_comm_create");
  c->world = world; c->rank = rank;
  hipError_t he = hipGetDevice(&c->device);
  if (he != hipSuccess) { delete c; return hip_fail(he, "hipGetDevice"); }
  ncclUniqueId u;
  std::memcpy(u.internal, id, DCRX_COMM_ID_BYTES);
  ncclResult_t e = R->CommInitRank(&c->comm, world, u, rank);
  if (e != ncclSuccess) { delete c; return nccl_err(R, e, "ncclCommInitRank"); }
  int rc = finish_create(R, c);
  if (rc) { dcrx_comm_destroy(c); return rc; }
  *out = c;
  return DCRX_OK;
}

int dcrx_comm_create_all(int world, const int *devices, dcrx_comm_t **out) {
  if (!out || world < 1) return set_err(DCRX_E_INVALID, "dcrx_comm_create_all: bad argument");
  for (int r = 0; r < world; r++) out[r] = nullptr;
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  std::vector<ncclComm_t> comms((size_t)world);
  std::vector<int> devs((size_t)world);
  for (int r = 0; r < world; r++) devs[(size_t)r] = devices ? devices[r] : r;
  NCCL_TRY(R, R->CommInitAll(comms.data(), world, devs.data()));
  int before = 0;
  (void)hipGetDevice(&before);
  int rc = DCRX_OK;
  for (int r = 0; r < world && rc == DCRX_OK; r++) {
    dcrx_comm *c = new (std::nothrow) dcrx_comm;
    if (!c) { rc = set_err(DCRX_E_NOMEM, "dcrx_comm_create_all"); break; }
    c->comm = comms[(size_t)r]; c->world = world; c->rank = r; c->device = devs[(size_t)r];
    out[r] = c;
    if (hipSetDevice(c->device) != hipSuccess) { rc = set_err(DCRX_E_HIP, "hipSetDevice"); break; }
    rc = finish_create(R, c);
  }
  (void)hipSetDevice(before);
  if (rc) for (int r = 0; r < world; r++) { if (out[r]) dcrx_comm_destroy(out[r]); else if (comms[(size_t)r]) R->CommDestroy(comms[(size_t)r]); out[r] = nullptr; }
  return rc;
}

void dcrx_comm_destroy(dcrx_comm_t *c) {
  if (!c) return;
  Rccl *R = rccl();
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->h_scratch) (void)hipHostFree(c->h_scratch);
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  delete c;
}

int dcrx_comm_info(const dcrx_comm_t *c, int *world, int *rank, int *device) {
  if (!c) return set_err(DCRX_E_INVALID, "null communicator");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  return DCRX_OK;
}

int dcrx_comm_allreduce_u64(dcrx_comm_t *c, const uint64_t *d_in, uint64_t *d_out, uint64_t n, int op, void *stream) {
  if (!c || !d_in || !d_out || op < 0 || op > 1) return set_err(DCRX_E_INVALID, "dcrx_comm_allreduce_u64: bad argument");
  Rccl *R = rccl();
  NCCL_TRY(R, R->AllReduce(d_in, d_out, (size_t)n, ncclUint64, op == DCRX_COMM_SUM ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
  return DCRX_OK;
}

int dcrx_comm_allreduce_f64(dcrx_comm_t *c, const double *d_in, double *d_out, uint64_t n, int op, void *stream) {
  if (!c || !d_in || !d_out || op < 0 || op > 1) return set_err(DCRX_E_INVALID, "dcrx_comm_allreduce_f64: bad argument");
  Rccl *R = rccl();
  NCCL_TRY(R, R->AllReduce(d_in, d_out, (size_t)n, ncclFloat64, op == DCRX_COMM_SUM ? ncclSum : ncclMax, c->comm, (hipStream_t)stream));
  return DCRX_OK;
}

int dcrx_comm_allgather(dcrx_comm_t *c, const void *d_in, void *d_out, uint64_t bytes_per_rank, void *stream) {
  if (!c || !d_in || !d_out) return set_err(DCRX_E_INVALID, "dcrx_comm_allgather: bad argument");
  Rccl *R = rccl();
  NCCL_TRY(R, R->AllGather(d_in, d_out, (size_t)bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream));
  return DCRX_OK;
}

int dcrx_comm_gather_v(dcrx_comm_t *c, const void *d_send, uint64_t send_bytes, void *const *d_recv, const uint64_t *recv_bytes, int root,
                       void *stream) {
  if (!c || root < 0 || root >= c->world) return set_err(DCRX_E_INVALID, "dcrx_comm_gather_v: bad argument");
  if (c->rank == root && c->world > 1 && (!d_recv || !recv_bytes)) return set_err(DCRX_E_INVALID, "dcrx_comm_gather_v: the root needs its receive buffers");
  if (c->rank != root && send_bytes && !d_send) return set_err(DCRX_E_INVALID, "dcrx_comm_gather_v: null message");
  if (c->world == 1) return DCRX_OK;      // (the root's own message stays where it is)
  Rccl *R = rccl();
  // one group: on RCCL one launch for all peers' receives, not one per peer
  NCCL_TRY(R, R->GroupStart());
  ncclResult_t e = ncclSuccess;
  if (c->rank == root) {
    for (int r = 0; r < c->world && e == ncclSuccess; r++)
      if (r != root && recv_bytes[r]) e = R->Recv(d_recv[r], (size_t)recv_bytes[r], ncclUint8, r, c->comm, (hipStream_t)stream);
  } else if (send_bytes) {
    e = R->Send(d_send, (size_t)send_bytes, ncclUint8, root, c->comm, (hipStream_t)stream);
  }
  ncclResult_t e2 = R->GroupEnd();
  if (e != ncclSuccess) return nccl_err(R, e, "ncclSend / ncclRecv");
  if (e2 != ncclSuccess) return nccl_err(R, e2, "ncclGroupEnd");
  return DCRX_OK;
}

int dcrx_comm_barrier(dcrx_comm_t *c, void *stream) {
  if (!c) return set_err(DCRX_E_INVALID, "null communicator");
  Rccl *R = rccl();
  NCCL_TRY(R, R->AllReduce(c->d_scratch + 1, c->d_scratch + 1, 1, ncclUint64, ncclSum, c->comm, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return DCRX_OK;
}

// ---- host-side exchanges (control plane: sizes, error flags, a Counter's keys): staged through device memory, synchronous ----
int dcrx_comm_allgather_host(dcrx_comm_t *c, const void *h_in, void *h_out, uint64_t bytes_per_rank) {
  if (!c || (bytes_per_rank && (!h_in || !h_out))) return set_err(DCRX_E_INVALID, "dcrx_comm_allgather_host: bad argument");
  if (!bytes_per_rank) return DCRX_OK;
  if (c->world == 1) { std::memcpy(h_out, h_in, (size_t)bytes_per_rank); return DCRX_OK; }
  Rccl *R = rccl();
  void *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_in, (size_t)bytes_per_rank));
  hipError_t he = hipMalloc(&d_out, (size_t)bytes_per_rank * (size_t)c->world);
  if (he != hipSuccess) { (void)hipFree(d_in); return hip_fail(he, "hipMalloc"); }
  int rc = DCRX_OK;
  he = hipMemcpy(d_in, h_in, (size_t)bytes_per_rank, hipMemcpyHostToDevice);
  if (he != hipSuccess) rc = hip_fail(he, "hipMemcpy");
  if (!rc) { ncclResult_t e = R->AllGather(d_in, d_out, (size_t)bytes_per_rank, ncclUint8, c->comm, nullptr); if (e != ncclSuccess) rc = nccl_err(R, e, "ncclAllGather"); }
  if (!rc) { he = hipStreamSynchronize(nullptr); if (he != hipSuccess) rc = hip_fail(he, "hipStreamSynchronize"); }
  if (!rc) { he = hipMemcpy(h_out, d_out, (size_t)bytes_per_rank * (size_t)c->world, hipMemcpyDeviceToHost); if (he != hipSuccess) rc = hip_fail(he, "hipMemcpy"); }
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

int dcrx_comm_allreduce_host_u64(dcrx_comm_t *c, uint64_t *h_inout, uint64_t n, int op) {
  if (!c || (n && !h_inout) || op < 0 || op > 1) return set_err(DCRX_E_INVALID, "dcrx_comm_allreduce_host_u64: bad argument");
  if (!n || c->world == 1) return DCRX_OK;
  Rccl *R = rccl();
  uint64_t *d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), (size_t)n * 8));
  int rc = DCRX_OK;
  hipError_t he = hipMemcpy(d, h_inout, (size_t)n * 8, hipMemcpyHostToDevice);
  if (he != hipSuccess) rc = hip_fail(he, "hipMemcpy");
  if (!rc) { ncclResult_t e = R->AllReduce(d, d, (size_t)n, ncclUint64, op == DCRX_COMM_SUM ? ncclSum : ncclMax, c->comm, nullptr); if (e != ncclSuccess) rc = nccl_err(R, e, "ncclAllReduce"); }
  if (!rc) { he = hipStreamSynchronize(nullptr); if (he != hipSuccess) rc = hip_fail(he, "hipStreamSynchronize"); }
  if (!rc) { he = hipMemcpy(h_inout, d, (size_t)n * 8, hipMemcpyDeviceToHost); if (he != hipSuccess) rc = hip_fail(he, "hipMemcpy"); }
  (void)hipFree(d);
  return rc;
}

// ---- the whole exchange of one step, in one call ----
int dcrx_decombine_sharded(dcrx_tables_t *tables, dcrx_comm_t *c, const dcrx_cfg_t *cfg, const dcrx_batch_t *d_shard, dcrx_record_t *d_records,
                           uint64_t *d_counters, const dcrx_tuple_layout_t *layout, void *d_message, uint64_t n_slots, void *const *d_gathered,
                           uint64_t *n_hits_by_rank, void *stream) {
  if (!tables || !c || !cfg || !d_shard || !d_records || !d_counters || !layout || !d_message || !n_hits_by_rank)
    return set_err(DCRX_E_INVALID, "null argument to dcrx_decombine_sharded");
  if (c->rank == 0 && c->world > 1 && !d_gathered) return set_err(DCRX_E_INVALID, "dcrx_decombine_sharded: rank 0 needs the ranks' message buffers");
  if (n_slots < d_shard->n_reads) return set_err(DCRX_E_INVALID, "dcrx_decombine_sharded: the message holds fewer read slots than the shard has reads");
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  hipStream_t s = (hipStream_t)stream;
  // the step's hot path, its tuples left in the message by the kernels themselves (dcrx_set_tuple_sink)
  int rc = dcrx_set_tuple_sink(tables, layout, d_message, n_slots, c->d_scratch);
  if (rc) return rc;
  rc = dcrx_decombine_device(tables, cfg, d_shard, d_records, d_counters, stream);
  (void)dcrx_set_tuple_sink(tables, nullptr, nullptr, 0, nullptr);
  if (rc) return rc;
  // the ranks' counts: all-gathered on the device, read by the host (the one wait of this call: the exact sizes are needed to post
  // the transfers; a caller that pipelines steps uses the pieces — dcrx_comm_allgather, dcrx_comm_gather_v — on a stream of its own)
  NCCL_TRY(R, R->AllGather(c->d_scratch, c->d_scratch + 2, 1, ncclUint64, c->comm, s));
  HIP_TRY(hipMemcpyAsync(c->h_scratch + 2, c->d_scratch + 2, (size_t)c->world * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<uint64_t> bytes((size_t)c->world);
  for (int r = 0; r < c->world; r++) {
    n_hits_by_rank[r] = c->h_scratch[2 + r];
    bytes[(size_t)r] = dcrx_tuple_message_bytes(layout, n_slots, n_hits_by_rank[r]);
  }
  rc = dcrx_comm_gather_v(c, d_message, bytes[(size_t)c->rank], d_gathered, bytes.data(), 0, stream);
  if (rc) return rc;
  // the counters of the whole job, on every rank (decombine.py:598: one Counter for the run)
  NCCL_TRY(R, R->AllReduce(d_counters, d_counters, DCRX_N_COUNTERS, ncclUint64, ncclSum, c->comm, s));
  return DCRX_OK;
}

// ---- streams and asynchronous copies, for callers that pipeline the pieces above without a HIP binding of their own ----
int dcrx_stream_create(void **stream) {
  if (!stream) return set_err(DCRX_E_INVALID, "null stream");
  hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = s; return DCRX_OK;
}
int dcrx_stream_destroy(void *stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return DCRX_OK; }
int dcrx_stream_synchronize(void *stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return DCRX_OK; }
int dcrx_stream_wait_event(void *stream, void *event) { HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0)); return DCRX_OK; }
int dcrx_event_create_ordering(void **event) {      // (no timestamps: what a stream waits for, not what is timed — cheaper to record)
  if (!event) return set_err(DCRX_E_INVALID, "null event");
  hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); *event = e; return DCRX_OK;
}
int dcrx_event_synchronize(void *event) { HIP_TRY(hipEventSynchronize((hipEvent_t)event)); return DCRX_OK; }
int dcrx_memcpy_d2h_async(void *dst_host, const void *src_device, size_t bytes, void *stream) {
  HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream)); return DCRX_OK;
}
int dcrx_memcpy_d2d_async(void *dst_device, const void *src_device, size_t bytes, void *stream) {
  HIP_TRY(hipMemcpyAsync(dst_device, src_device, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return DCRX_OK;
}
int dcrx_memset_device_async(void *dst_device, int value, size_t bytes, void *stream) {
  HIP_TRY(hipMemsetAsync(dst_device, value, bytes, (hipStream_t)stream)); return DCRX_OK;
}

}  // extern "C"
