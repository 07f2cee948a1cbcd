// dcrx_api.cpp — the C ABI declared in include/dcrx.h and include/dcrx_synth.h.
//
// Host side of the boundary: validates arguments, owns the table handle and its
// per-device workspace, launches the kernels of dcrx_kernels.hip.  There is no
// CPU implementation of the hot path in this library: every decombine entry
// point needs a GPU and fails with DCRX_E_NOGPU / DCRX_E_HIP without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dcrx.h"
#include "../../include/dcrx_synth.h"
#include "dcrx_launch.h"
#include "dcrx_synth_core.h"
#include <cstdlib>
#include <thread>

#include "dcrx_tables.h"

namespace dcrx {
hipError_t launch_synth(const DevTables &T, const SynthParams &P, uint64_t first, uint64_t n, uint32_t stride,
                        uint8_t *d_packed, hipStream_t s);
}

using namespace dcrx;

static thread_local std::string g_err;

static int set_err(int code, const std::string &m) { g_err = m; return code; }
namespace dcrx { int set_err(int code, const char *msg) { return ::set_err(code, std::string(msg)); } }
static int hip_err(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? DCRX_E_NOGPU : DCRX_E_HIP;
}
#define HIP_TRY(call)                                   \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_err(e_, #call);    \
  } while (0)

// reads per chunk of the host-buffer entry (dcrx_decombine): 80 MB in, 32 MB out at 150 nt
#ifndef DCRX_HOST_CHUNK
#define DCRX_HOST_CHUNK (2u << 20)
#endif

// memcpy over a few threads (a pinned staging buffer has to be filled faster than the link drains it)
static void par_memcpy(void *dst, const void *src, size_t n) {
  static const unsigned want = [] { const char *e = std::getenv("DCRX_HOST_THREADS"); unsigned k = e ? (unsigned)std::atoi(e) : std::min(12u, std::max(1u, std::thread::hardware_concurrency() / 2)); return std::max(1u, std::min(k, 16u)); }();
  const unsigned nt = n < (8u << 20) ? 1u : want;
  if (nt == 1) { std::memcpy(dst, src, n); return; }
  // (a worker that cannot be started — std::system_error, or no memory for the vector — leaves its slice to this thread:
  // nothing is thrown past the threads already running)
  std::thread th[16];
  const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
  for (unsigned k = 1; k < nt; k++) {
    const size_t a = std::min(n, per * k), b = std::min(n, per * (k + 1));
    if (b <= a) continue;
    try { th[k] = std::thread([=] { std::memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a); }); }
    catch (...) { std::memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a); }
  }
  std::memcpy(dst, src, std::min(n, per));
  for (auto &x : th) if (x.joinable()) x.join();
}

struct dcrx_tables {
  HostTables host;
  // state on the device the tables were last used on
  int device = -1;
  uint8_t *d_blob = nullptr;
  DevTables dev{};
  DevTables *d_dev = nullptr;   // the same struct in device memory (what a kernel's rare paths read instead of holding it in registers)
  LaunchPlan plan{};
  bool ws_dirty = true;       // work counters / exception bitmap must be zeroed before the next launch
  uint32_t reserved_cus = 0;
  uint32_t *d_exc_flag = nullptr;
  uint64_t exc_flag_reads = 0;
  uint32_t *d_queue = nullptr;  // [DCRX_QUEUE_HEADER work counters][exc_flag_reads rescue indices][exc_flag_reads general indices]
  void *d_v2_tail = nullptr, *d_v2_events = nullptr, *d_v2_slow = nullptr;
  // the tail list (a third of the lists' bytes) exists only for handles whose calls keep the tail a role of the finishing launch:
  // where the scan takes the tail through its ring in LDS nothing is ever stored in it.  Unknown: decided by the first use
  // (table sizes), corrected by the first launch that turns out to need the list (dcrx_decombine_device allocates it and launches again)
  int want_tail = -1;
  void *d_v2_left = nullptr;        // the finishing launch's left list
  uint64_t *d_v2_acc = nullptr;     // the v2 kernels' tallies of the call in flight (zero between calls)  // v2 kernels: the per-wave lists between scan and finishing
  hipStream_t v2_side = nullptr, v2_side2 = nullptr; hipEvent_t v2_ev_fork = nullptr, v2_ev_join = nullptr, v2_ev_join2 = nullptr;
  uint32_t *d_v2_counts = nullptr;
  // staging for the host-buffer entry point: two sets of device buffers and pinned host buffers, three streams
  // (copies in, kernels, copies out) and the events that order them
  uint8_t *d_stage = nullptr;
  size_t stage_bytes = 0;
  uint8_t *h_stage = nullptr;       // pinned
  size_t h_stage_bytes = 0;
  hipStream_t hs_in = nullptr, hs_run = nullptr, hs_out = nullptr;
  hipEvent_t hev_in[2] = {nullptr, nullptr}, hev_run[2] = {nullptr, nullptr}, hev_out[2] = {nullptr, nullptr};
  bool constants_ready = false;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;      // around the dominant kernel
  hipEvent_t ev_step_start = nullptr, ev_step_stop = nullptr;  // around every launch of a call
  V2Tune tune[2];                   // the handle's timing of its own finishing launches (launch_v2), per frame
  // the tuple sink (dcrx_set_tuple_sink): where the next calls leave their message, and the kernels' side of it on the device
  bool sink_on = false;
  dcrx_tuple_layout_t sink_layout{};
  uint8_t *sink_msg = nullptr; uint64_t sink_slots = 0; uint64_t *sink_total = nullptr;
  V2SinkDev *d_sink = nullptr;
  void *d_sink_items = nullptr; uint32_t *d_sink_ctr = nullptr;
  uint64_t sink_items_cap = 0; uint32_t sink_regions_cap = 0;
};

static int layout_dev(dcrx_tables *t, const dcrx_tuple_layout_t *L, TupleLayoutDev *D);
static int compact_workspace(uint64_t n_reads, uint32_t **tc, uint64_t **to);

static void free_device_state(dcrx_tables *t) {
  if (t->device < 0) return;
  (void)hipFree(t->d_dev); t->d_dev = nullptr;
  (void)hipFree(t->d_blob); (void)hipFree(t->d_exc_flag); (void)hipFree(t->d_queue); (void)hipFree(t->d_v2_tail); (void)hipFree(t->d_v2_events); (void)hipFree(t->d_v2_counts);
  (void)hipFree(t->d_v2_slow);
  for (V2Tune &U : t->tune) {
    const bool may_wait = U.may_wait;
    for (V2TuneSlot &K : U.slot) {
      if (K.created) for (auto &pair : K.ev) { (void)hipEventDestroy(pair[0]); (void)hipEventDestroy(pair[1]); }
      if (K.ev_counts) (void)hipEventDestroy(K.ev_counts);
      for (auto &pair : K.ev_e) for (auto &ev : pair) if (ev) (void)hipEventDestroy(ev);
      if (K.h_counts) (void)hipHostFree(K.h_counts);
    }
    U = V2Tune{};
    U.may_wait = may_wait;
  }
  (void)hipFree(t->d_sink); (void)hipFree(t->d_sink_items); (void)hipFree(t->d_sink_ctr);
  t->d_sink = nullptr; t->d_sink_items = nullptr; t->d_sink_ctr = nullptr; t->sink_items_cap = 0; t->sink_regions_cap = 0;
  if (t->v2_side) (void)hipStreamDestroy(t->v2_side);
  if (t->v2_side2) (void)hipStreamDestroy(t->v2_side2);
  if (t->v2_ev_fork) (void)hipEventDestroy(t->v2_ev_fork);
  if (t->v2_ev_join) (void)hipEventDestroy(t->v2_ev_join);
  if (t->v2_ev_join2) (void)hipEventDestroy(t->v2_ev_join2);
  t->v2_side = t->v2_side2 = nullptr; t->v2_ev_fork = t->v2_ev_join = t->v2_ev_join2 = nullptr;
  t->d_v2_tail = nullptr; t->d_v2_events = nullptr; t->d_v2_counts = nullptr; t->d_v2_slow = nullptr;
  (void)hipFree(t->d_v2_acc); t->d_v2_acc = nullptr; t->plan.v2_acc = nullptr;
  (void)hipFree(t->d_v2_left); t->d_v2_left = nullptr; t->plan.v2_left = nullptr;
  (void)hipFree(t->d_stage);
  if (t->h_stage) (void)hipHostFree(t->h_stage);
  t->h_stage = nullptr; t->h_stage_bytes = 0;
  if (t->hs_in) (void)hipStreamDestroy(t->hs_in);
  if (t->hs_run) (void)hipStreamDestroy(t->hs_run);
  if (t->hs_out) (void)hipStreamDestroy(t->hs_out);
  t->hs_in = t->hs_run = t->hs_out = nullptr;
  for (int k = 0; k < 2; k++) {
    if (t->hev_in[k]) (void)hipEventDestroy(t->hev_in[k]);
    if (t->hev_run[k]) (void)hipEventDestroy(t->hev_run[k]);
    if (t->hev_out[k]) (void)hipEventDestroy(t->hev_out[k]);
    t->hev_in[k] = t->hev_run[k] = t->hev_out[k] = nullptr;
  }
  t->d_blob = nullptr; t->d_exc_flag = nullptr; t->d_queue = nullptr;
  t->d_stage = nullptr;
  t->exc_flag_reads = 0; t->stage_bytes = 0; t->device = -1; t->constants_ready = false;
}

extern "C" {

int dcrx_abi_version(void) { return DCRX_ABI_VERSION; }
const char *dcrx_last_error(void) { return g_err.c_str(); }
const char *dcrx_build_info(void) { return "dcrx hip kernels: gfx950; v2 scan block 1024 (16-bit pair table), finishing blocks 256; reads <= 511 nt on register shapes of 10 / 20 / 32 words per read, 512 .. 65535 nt one read per lane from memory"; }

int dcrx_tables_create(const dcrx_tagset_t *tagset, dcrx_tables_t **out) {
  if (!out) return set_err(DCRX_E_INVALID, "out is null");
  *out = nullptr;
  dcrx_tables *t = new (std::nothrow) dcrx_tables();
  if (!t) return set_err(DCRX_E_NOMEM, "out of memory");
  std::string err;
  int rc;
  try { rc = compile_tables(tagset, &t->host, &err); }
  catch (...) { rc = DCRX_E_NOMEM; err = "out of memory while compiling the tag tables"; }
  if (rc != DCRX_OK) { delete t; return set_err(rc, err); }
  *out = t;
  return DCRX_OK;
}

void dcrx_tables_destroy(dcrx_tables_t *t) {
  if (!t) return;
  if (t->device >= 0) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) {
      if (cur != t->device) (void)hipSetDevice(t->device);
      free_device_state(t);
      if (cur != t->device && cur >= 0) (void)hipSetDevice(cur);
    }
  }
  delete t;
}

int dcrx_tables_info(const dcrx_tables_t *t, dcrx_tables_info_t *info) {
  if (!t || !info) return set_err(DCRX_E_INVALID, "null argument");
  std::memset(info, 0, sizeof *info);
  info->n_v = t->host.g[0].n; info->n_j = t->host.g[1].n;
  info->n_states = t->host.n_states; info->dfa_bytes = t->host.dfa_bytes;
  for (int c = 0; c < 6; c++) info->n_keywords[c] = t->host.n_keywords[c];
  info->max_tag_len = t->host.max_tag_len;
  info->tables_in_lds = t->host.rel.lds_image_bytes + 128 <= 120 * 1024;
  info->equal_len_per_automaton = t->host.equal_len_per_automaton ? 1 : 0;
  info->pair_scan_bytes = t->host.rel.dfa16_bytes;
  info->v2_tables = t->host.rel.v2_ok;
  for (int o = 0; o < 2; o++) { info->v2_states[o] = t->host.rel.v2[o].n_states; info->v2_scan_bytes[o] = t->host.rel.v2[o].trans_bytes; }
  info->max_read_len = DCRX_MAX_READ_LEN;
  return DCRX_OK;
}

// ---- device plumbing ------------------------------------------------------------
int dcrx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int dcrx_set_device(int device) { HIP_TRY(hipSetDevice(device)); return DCRX_OK; }
int dcrx_device_name(char *buf, size_t cap) {
  if (!buf || !cap) return set_err(DCRX_E_INVALID, "null buffer");
  int dev = 0; HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p; HIP_TRY(hipGetDeviceProperties(&p, dev));
  std::snprintf(buf, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return DCRX_OK;
}
int dcrx_malloc_device(void **ptr, size_t bytes) {
  if (!ptr) return set_err(DCRX_E_INVALID, "null ptr");
  HIP_TRY(hipMalloc(ptr, bytes ? bytes : 16)); return DCRX_OK;
}
int dcrx_free_device(void *ptr) { HIP_TRY(hipFree(ptr)); return DCRX_OK; }
int dcrx_malloc_host(void **ptr, size_t bytes) {
  if (!ptr) return set_err(DCRX_E_INVALID, "null argument");
  *ptr = nullptr;
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return DCRX_OK;
}
int dcrx_free_host(void *ptr) { if (ptr) HIP_TRY(hipHostFree(ptr)); return DCRX_OK; }
int dcrx_memcpy_h2d(void *d, const void *h, size_t bytes) { HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return DCRX_OK; }
int dcrx_memcpy_d2h(void *h, const void *d, size_t bytes) { HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return DCRX_OK; }
int dcrx_memset_device(void *d, int value, size_t bytes) { HIP_TRY(hipMemset(d, value, bytes)); return DCRX_OK; }
int dcrx_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return DCRX_OK; }
int dcrx_event_create(void **ev) {
  if (!ev) return set_err(DCRX_E_INVALID, "null event");
  hipEvent_t e; HIP_TRY(hipEventCreate(&e)); *ev = e; return DCRX_OK;
}
int dcrx_event_destroy(void *ev) { HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return DCRX_OK; }
int dcrx_event_record(void *ev, void *stream) { HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return DCRX_OK; }
int dcrx_event_elapsed_ms(void *a, void *b, float *ms) {
  if (!ms) return set_err(DCRX_E_INVALID, "null ms");
  HIP_TRY(hipEventSynchronize((hipEvent_t)b));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return DCRX_OK;
}

}  // extern "C"

// ---- per-device state -------------------------------------------------------------
static int ensure_device(dcrx_tables *t, uint64_t max_reads, uint32_t stride = 40, hipStream_t stream = nullptr) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (t->device != dev) {
    if (t->device >= 0) {  // tables move with the caller's current device
      int old = t->device;
      (void)hipSetDevice(old); free_device_state(t); (void)hipSetDevice(dev);
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    HIP_TRY(hipMalloc(&t->d_blob, t->host.blob.size()));
    HIP_TRY(hipMemcpy(t->d_blob, t->host.blob.data(), t->host.blob.size(), hipMemcpyHostToDevice));
    t->dev = t->host.resolve(t->d_blob);
    HIP_TRY(hipMalloc(&t->d_dev, sizeof(DevTables)));
    HIP_TRY(hipMemcpy(t->d_dev, &t->dev, sizeof(DevTables), hipMemcpyHostToDevice));
    // launch plan: persistent blocks of DCRX_BLOCK threads, the DFA resident in LDS
    const uint32_t lds_cap = 160 * 1024;
    const uint32_t want = t->host.rel.lds_image_bytes + DCRX_N_COUNTERS * 4;
    LaunchPlan P;
    P.n_cu = (uint32_t)prop.multiProcessorCount;
    P.table_in_lds = want <= 120 * 1024;
    P.lds_bytes = P.table_in_lds ? want : DCRX_N_COUNTERS * 4;
    const uint32_t side_bytes = t->host.rel.lds_image_bytes - t->host.rel.dfa_bytes;
    P.lds16_bytes = DCRX_N_COUNTERS * 4 + t->host.rel.dfa16_bytes + side_bytes;
    P.table16_in_lds = P.table_in_lds && t->host.rel.dfa16_bytes != 0 && P.lds16_bytes + 32768 <= lds_cap;  // + wq and tail buffers of the 1024-thread block
    uint32_t per_cu = std::min<uint32_t>(2048 / DCRX_BLOCK, std::max<uint32_t>(1, lds_cap / std::max<uint32_t>(P.lds_bytes, 1)));
    P.grid = (uint32_t)prop.multiProcessorCount * std::max<uint32_t>(per_cu, 1);  // upper bound for either fast kernel
    const uint32_t q_per_cu = std::min<uint32_t>(2048 / DCRX_QBLOCK, std::max<uint32_t>(1, lds_cap / std::max<uint32_t>(P.lds_bytes, 1)));
    P.qgrid = (uint32_t)prop.multiProcessorCount * q_per_cu;
    P.reserved_cus = t->reserved_cus;
    P.dev_tables = t->d_dev;
    P.tune = t->tune;
    t->plan = P;
    t->device = dev;
  }
  if (max_reads < 4096) max_reads = 4096;  // workspace exists even for empty batches
  if (max_reads > t->exc_flag_reads) {
    (void)hipFree(t->d_exc_flag); t->d_exc_flag = nullptr;
    (void)hipFree(t->d_queue); t->d_queue = nullptr;
    HIP_TRY(hipMalloc(&t->d_exc_flag, ((max_reads + 31) / 32) * 4 + 16));
    HIP_TRY(hipMalloc(&t->d_queue, (3 * max_reads + DCRX_QUEUE_HEADER) * 4));  // [work counters][rescue queue][general list][its exception-list offsets]
    t->exc_flag_reads = max_reads;
    t->ws_dirty = true;
  }
  if (t->host.rel.v2_ok && stride <= DCRX_FAST_MAX_STRIDE) {
    // the lists between the v2 kernels: every wave of the scan kernel (16 per CU) owns a region of tail and of event
    // entries; an entry carries the read's packed words, so the size follows the stride
    uint64_t tr = 0, er = 0;
    v2_list_rows(max_reads, stride, t->plan.n_cu, &tr, &er);
    // (the fused form: 150-nt shapes, pair tables of up to 64 KB — launch_v2; either frame may be asked for)
    if (t->want_tail < 0)
      t->want_tail = (stride <= 40 && std::max(t->host.rel.v2[0].trans_bytes, t->host.rel.v2[1].trans_bytes) <= 64u * 1024u) ? 0 : 1;
    if (t->want_tail && tr > t->plan.v2_tail_rows) {
      (void)hipFree(t->d_v2_tail); t->d_v2_tail = nullptr; t->plan.v2_tail = nullptr; t->plan.v2_tail_rows = 0;
      HIP_TRY(hipMalloc(&t->d_v2_tail, tr * 16));
      t->plan.v2_tail = reinterpret_cast<uint4 *>(t->d_v2_tail); t->plan.v2_tail_rows = tr;
    }
    if (er > t->plan.v2_event_rows || v2_slow_rows(max_reads, stride, t->plan.n_cu) > t->plan.v2_slow_rows) {
      (void)hipFree(t->d_v2_events); (void)hipFree(t->d_v2_counts); (void)hipFree(t->d_v2_slow);
      t->d_v2_events = nullptr; t->d_v2_counts = nullptr; t->d_v2_slow = nullptr;
      t->plan.v2_events = nullptr; t->plan.v2_slow = nullptr;
      t->plan.v2_event_rows = t->plan.v2_slow_rows = 0;
      const uint64_t sr = v2_slow_rows(max_reads, stride, t->plan.n_cu);
      HIP_TRY(hipMalloc(&t->d_v2_events, er * 16));
      HIP_TRY(hipMalloc(&t->d_v2_slow, sr * 16));
      HIP_TRY(hipMalloc(&t->d_v2_counts, (size_t)t->plan.n_cu * 16 * 16));
      HIP_TRY(hipMemset(t->d_v2_counts, 0, (size_t)t->plan.n_cu * 16 * 16));      // (no hint yet of a region's last share of tail reads: scan2_kernel, V2_L_TWHINT)
      if (!t->d_v2_acc) { HIP_TRY(hipMalloc(&t->d_v2_acc, DCRX_N_COUNTERS * 8)); t->ws_dirty = true; }
      t->plan.v2_acc = t->d_v2_acc;
      if (!t->d_v2_left) {      // V2_LEFT_CAP entries of 32-word reads, and a valid word per entry (zero between launches)
        const size_t left_bytes = (size_t)1024 * ((1 + 2 * DCRX_V2_NWLONG + 3) / 4) * 16 + 1024 * 4;
        HIP_TRY(hipMalloc(&t->d_v2_left, left_bytes));
        HIP_TRY(hipMemset(t->d_v2_left, 0, left_bytes));
      }
      t->plan.v2_left = reinterpret_cast<uint4 *>(t->d_v2_left);
      t->plan.v2_events = reinterpret_cast<uint4 *>(t->d_v2_events);
      t->plan.v2_slow = reinterpret_cast<uint4 *>(t->d_v2_slow);
      t->plan.v2_counts = t->d_v2_counts; t->plan.v2_event_rows = er; t->plan.v2_slow_rows = sr;
      if (!t->v2_side) {     // the two side streams (tail kernel; general form over list X) and the events that fork them off the caller's stream and join them back
        if (hipStreamCreateWithFlags(&t->v2_side, hipStreamNonBlocking) != hipSuccess) t->v2_side = nullptr;
        if (t->v2_side && hipStreamCreateWithFlags(&t->v2_side2, hipStreamNonBlocking) != hipSuccess) t->v2_side2 = nullptr;
        if (t->v2_side && (!t->v2_side2 || hipEventCreateWithFlags(&t->v2_ev_fork, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&t->v2_ev_join, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&t->v2_ev_join2, hipEventDisableTiming) != hipSuccess)) {
          (void)hipStreamDestroy(t->v2_side); t->v2_side = nullptr;
          if (t->v2_side2) { (void)hipStreamDestroy(t->v2_side2); t->v2_side2 = nullptr; }
        }
      }
      t->plan.v2_side = t->v2_side; t->plan.v2_side2 = t->v2_side2;
      t->plan.v2_ev_fork = t->v2_ev_fork; t->plan.v2_ev_join = t->v2_ev_join; t->plan.v2_ev_join2 = t->v2_ev_join2;
    }
  }
  if (t->ws_dirty) {
    // the kernels leave the work counters and the exception bitmap zeroed; they are zeroed here
    // only once per allocation, or after a launch that failed part-way
    // on the stream the kernels will run on (a non-blocking stream does not order against the null stream)
    HIP_TRY(hipMemsetAsync(t->d_exc_flag, 0, ((t->exc_flag_reads + 31) / 32) * 4 + 16, stream));
    HIP_TRY(hipMemsetAsync(t->d_queue, 0, DCRX_QUEUE_HEADER * 4, stream));
    if (t->d_v2_acc) HIP_TRY(hipMemsetAsync(t->d_v2_acc, 0, DCRX_N_COUNTERS * 8, stream));
    // dcrx_reserve_device and the host-buffer entry come here with the null stream: the fills must have landed before a
    // later call's kernels start on a non-blocking stream of the caller's, which nothing orders against the null stream
    if (!stream) HIP_TRY(hipStreamSynchronize(nullptr));
    t->ws_dirty = false;
  }
  return DCRX_OK;
}

// The device side of the tuple sink for batches of up to max_reads reads: the regions' slabs of items, their counters and
// the descriptor the kernels read (allocated on the first call that needs them, grown by a larger batch: synchronises).
static int ensure_sink(dcrx_tables *t, uint64_t max_reads, hipStream_t stream) {
  const uint64_t want = v2_sink_items(std::max<uint64_t>(max_reads, 4096), t->plan.n_cu);
  const uint32_t regions = t->plan.n_cu;
  if (t->d_sink && want <= t->sink_items_cap && regions <= t->sink_regions_cap) return DCRX_OK;
  HIP_TRY(hipStreamSynchronize(stream));
  (void)hipFree(t->d_sink_items); (void)hipFree(t->d_sink_ctr); (void)hipFree(t->d_sink);
  t->d_sink_items = nullptr; t->d_sink_ctr = nullptr; t->d_sink = nullptr; t->sink_items_cap = 0; t->sink_regions_cap = 0;
  HIP_TRY(hipMalloc(&t->d_sink_items, want * 8));
  HIP_TRY(hipMalloc(&t->d_sink_ctr, ((size_t)2 * regions + 16) * 4));
  HIP_TRY(hipMemset(t->d_sink_ctr, 0, ((size_t)2 * regions + 16) * 4));
  HIP_TRY(hipMalloc(&t->d_sink, sizeof(V2SinkDev)));
  V2SinkDev D;
  D.late = t->d_sink_ctr + regions; D.ticket = t->d_sink_ctr + 2 * regions;      // (the regions' counts of decombined reads in front: V2SinkCall::hits)
  D.j_tag_len = t->dev.g[1].tag_len; D.j_jump = t->dev.g[1].jump;
  HIP_TRY(hipMemcpy(t->d_sink, &D, sizeof D, hipMemcpyHostToDevice));
  t->sink_items_cap = want; t->sink_regions_cap = regions;
  return DCRX_OK;
}

static int check_batch(const dcrx_batch_t *b) {
  if (!b) return set_err(DCRX_E_INVALID, "batch is null");
  if (b->n_reads >= (1ull << 32)) return set_err(DCRX_E_INVALID, "more than 2^32-1 reads in one call");
  if (b->stride == 0 || (b->stride & 7u)) return set_err(DCRX_E_INVALID, "stride must be a positive multiple of 8");
  if (b->stride > DCRX_MAX_STRIDE) return set_err(DCRX_E_UNSUPPORTED, "stride > 16384 bytes: reads longer than 65535 nt are not supported");
  if (!b->lens && b->read_len > DCRX_MAX_READ_LEN) return set_err(DCRX_E_UNSUPPORTED, "reads longer than 65535 nt are not supported");
  if (!b->lens && b->read_len > 4 * b->stride) return set_err(DCRX_E_INVALID, "read_len exceeds 4*stride");
  if (b->n_reads && !b->packed) return set_err(DCRX_E_INVALID, "packed is null");
  if (b->n_exc && (!b->exc_read || !b->exc_pos || !b->exc_chr)) return set_err(DCRX_E_INVALID, "exception arrays are null");
  if (b->n_exc >= (1ull << 32)) return set_err(DCRX_E_INVALID, "more than 2^32-1 exception entries in one call");
  return DCRX_OK;
}

extern "C" {

int dcrx_set_timing_events(dcrx_tables_t *t, void *start_event, void *stop_event) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  t->ev_start = (hipEvent_t)start_event; t->ev_stop = (hipEvent_t)stop_event;
  return DCRX_OK;
}

int dcrx_set_step_events(dcrx_tables_t *t, void *start_event, void *stop_event) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  t->ev_step_start = (hipEvent_t)start_event; t->ev_step_stop = (hipEvent_t)stop_event;
  return DCRX_OK;
}

int dcrx_reserve_device(dcrx_tables_t *t, uint64_t max_reads) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  return ensure_device(t, max_reads);
}

int dcrx_decombine_device(dcrx_tables_t *t, const dcrx_cfg_t *cfg, const dcrx_batch_t *b, dcrx_record_t *d_records,
                          uint64_t *d_counters, void *stream) {
  if (!t || !cfg || !d_counters) return set_err(DCRX_E_INVALID, "null argument");
  int rc = check_batch(b);
  if (rc) return rc;
  if (b->n_reads && !d_records) return set_err(DCRX_E_INVALID, "d_records is null");
  if (cfg->orientation < 0 || cfg->orientation > 2) return set_err(DCRX_E_INVALID, "orientation must be 0, 1 or 2");
  if (cfg->flags & DCRX_F_PROFILE_MASK) {       // profiling switches: the records are then not results
    static const bool debug_flags = getenv("DCRX_DEBUG_FLAGS") && getenv("DCRX_DEBUG_FLAGS")[0] == '1';
    if (!debug_flags) return set_err(DCRX_E_INVALID, "cfg.flags holds a profiling switch (records would not be results): set DCRX_DEBUG_FLAGS=1 to allow it");
  }
  rc = ensure_device(t, b->n_reads, b->stride, (hipStream_t)stream);
  if (rc) return rc;
  BatchDev B;
  B.packed = b->packed; B.stride = b->stride; B.read_len = b->read_len; B.lens = b->lens;
  B.n_reads = b->n_reads; B.n_exc = b->n_exc; B.exc_read = b->exc_read; B.exc_pos = b->exc_pos;
  B.exc_chr = b->exc_chr; B.exc_flag = t->d_exc_flag;
  CfgDev C{cfg->orientation, cfg->allow_ns, cfg->lenthreshold, cfg->flags};
  t->plan.ev_step_start = t->ev_step_start; t->plan.ev_step_stop = t->ev_step_stop;
  // the call's tuple sink: the kernels' own (tuples of up to 40 bits, the shipped launch shape: launch_v2), else a compaction
  // of the records behind the call, on the same stream
  bool sink_done = false;
  t->plan.sink = V2SinkJob{};
  TupleLayoutDev LD{};
  if (t->sink_on) {
    if (t->sink_slots < b->n_reads) return set_err(DCRX_E_INVALID, "tuple sink: the message holds fewer read slots than the batch has reads");
    rc = layout_dev(t, &t->sink_layout, &LD);
    if (rc) return rc;
    // (v_start and j_end travel in w_pos bits each: a batch whose reads can be longer than those bits hold would spill into the
    // neighbouring fields of every tuple)
    const uint64_t longest = b->lens ? (uint64_t)b->stride * 4u : (uint64_t)b->read_len;
    if (LD.w_pos < 32 && (longest >> LD.w_pos) != 0)
      return set_err(DCRX_E_INVALID, b->lens ? "tuple sink: a batch with per-read lengths is priced at the longest read its stride holds (4 * stride bases: the lengths "
                                                "live in device memory); give dcrx_tuple_layout a max_read_len of 4 * stride for such batches"
                                              : "tuple sink: the batch's reads are longer than the layout's position fields hold (dcrx_tuple_layout's max_read_len)");
    if (t->sink_layout.bits <= 40 && t->host.rel.v2_ok && b->stride <= DCRX_FAST_MAX_STRIDE) {
      rc = ensure_sink(t, b->n_reads, (hipStream_t)stream);
      if (rc) return rc;
      V2SinkJob &J = t->plan.sink;
      J.dev = t->d_sink; J.items = static_cast<uint2 *>(t->d_sink_items); J.hits = t->d_sink_ctr; J.items_cap = t->sink_items_cap; J.regions_cap = t->sink_regions_cap;
      J.wpack = LD.w_v | (LD.w_j << 5) | (LD.w_vdel << 10) | (LD.w_jdel << 15) | (LD.w_pos << 20);
      J.bytes = LD.bytes; J.msg = t->sink_msg; J.n_slots = t->sink_slots; J.d_total = t->sink_total; J.done = &sink_done;
    }
  }
  hipError_t le = launch_decombine(t->plan, t->dev, B, C, d_records, t->d_queue + DCRX_QUEUE_HEADER,
                                   t->d_queue + DCRX_QUEUE_HEADER + t->exc_flag_reads, t->d_queue, d_counters,
                                   (hipStream_t)stream, t->ev_start, t->ev_stop);
  if (le == hipErrorNotReady && !t->want_tail) {
    // the launch keeps the tail a role of the finishing launch (a frame whose table does not fuse, an A/B switch) and the handle
    // has no tail list yet: nothing was launched — the list is allocated and the call launched again
    // (orientation `both`: the first pass may have run and tallied before the second found its frame without a tail list — the
    // workspace is zeroed again, on the same stream behind whatever ran, before the call starts over)
    (void)hipGetLastError();
    t->want_tail = 1;
    t->ws_dirty = true;
    rc = ensure_device(t, b->n_reads, b->stride, (hipStream_t)stream);
    if (rc) { t->plan.sink = V2SinkJob{}; return rc; }
    le = launch_decombine(t->plan, t->dev, B, C, d_records, t->d_queue + DCRX_QUEUE_HEADER, t->d_queue + DCRX_QUEUE_HEADER + t->exc_flag_reads,
                          t->d_queue, d_counters, (hipStream_t)stream, t->ev_start, t->ev_stop);
  }
  t->plan.sink = V2SinkJob{};
  if (le != hipSuccess) { t->ws_dirty = true; return hip_err(le, "launch_decombine"); }
  if (t->sink_on && !sink_done) {
    uint32_t *tc = nullptr; uint64_t *to = nullptr;
    rc = compact_workspace(b->n_reads, &tc, &to);
    if (rc) return rc;
    HIP_TRY(launch_compact_narrow(d_records, b->n_reads, t->sink_msg, t->sink_slots, LD, t->sink_total, tc, to, (hipStream_t)stream));
  }
  return DCRX_OK;
}

static int decombine_host(dcrx_tables_t *t, const dcrx_cfg_t *cfg, const dcrx_batch_t *hb, dcrx_record_t *records, uint64_t *counters);

// The host-buffer entry.  Whatever goes wrong inside the chunk pipeline — a HIP error, no memory, a helper thread that cannot
// be started — comes back as a code, and only after the three streams have drained: their asynchronous copies may target the
// caller's own (pinned) `packed` and `records` buffers, which the caller is free to release once this returns.
int dcrx_decombine(dcrx_tables_t *t, const dcrx_cfg_t *cfg, const dcrx_batch_t *hb, dcrx_record_t *records,
                   uint64_t *counters) {
  int rc;
  const bool sink_was_on = t && t->sink_on;      // (the tuple sink concerns the device entry: a chunked host call has no one message)
  if (t) t->sink_on = false;
  try { rc = decombine_host(t, cfg, hb, records, counters); }
  catch (const std::bad_alloc &) { rc = set_err(DCRX_E_NOMEM, "out of host memory in dcrx_decombine"); }
  catch (const std::exception &e) { rc = set_err(DCRX_E_NOMEM, std::string("dcrx_decombine: ") + e.what()); }
  catch (...) { rc = set_err(DCRX_E_NOMEM, "dcrx_decombine: unexpected exception"); }
  if (rc != DCRX_OK && t && t->hs_in) {
    const std::string keep = g_err;       // (the synchronising calls must not replace the message of what failed)
    (void)hipStreamSynchronize(t->hs_in); (void)hipStreamSynchronize(t->hs_run); (void)hipStreamSynchronize(t->hs_out);
    (void)hipGetLastError();
    t->ws_dirty = true;
    g_err = keep;
  }
  if (t) t->sink_on = sink_was_on;
  return rc;
}

}  // extern "C"

static int decombine_host(dcrx_tables_t *t, const dcrx_cfg_t *cfg, const dcrx_batch_t *hb, dcrx_record_t *records, uint64_t *counters) {
  if (!t || !cfg || !counters) return set_err(DCRX_E_INVALID, "null argument");
  int rc = check_batch(hb);
  if (rc) return rc;
  const uint64_t n = hb->n_reads;
  if (n && !records) return set_err(DCRX_E_INVALID, "records is null");
  // host-side validation the device entry cannot afford
  if (hb->lens) {
    for (uint64_t r = 0; r < n; r++) {
      if (hb->lens[r] > 4 * hb->stride) return set_err(DCRX_E_INVALID, "a read is longer than 4*stride");
      if (hb->lens[r] > DCRX_MAX_READ_LEN) return set_err(DCRX_E_UNSUPPORTED, "a read is longer than 65535 nt");
    }
  }
  for (uint64_t i = 0; i < hb->n_exc; i++) {
    const uint8_t c = hb->exc_chr[i];
    if (c == 'A' || c == 'C' || c == 'G' || c == 'T') return set_err(DCRX_E_INVALID, "exception byte is one of ACGT");
    if (hb->exc_read[i] >= n) return set_err(DCRX_E_INVALID, "exception read index out of range");
    if (i && (hb->exc_read[i] < hb->exc_read[i - 1] ||
              (hb->exc_read[i] == hb->exc_read[i - 1] && hb->exc_pos[i] <= hb->exc_pos[i - 1])))
      return set_err(DCRX_E_INVALID, "exceptions are not sorted by (read, pos)");
    const uint32_t len = hb->lens ? hb->lens[hb->exc_read[i]] : hb->read_len;
    if (hb->exc_pos[i] >= len) return set_err(DCRX_E_INVALID, "exception position beyond the read");
  }
  // The batch goes through in chunks of DCRX_HOST_CHUNK reads, three streams deep: while the kernels of chunk k run, chunk
  // k + 1 is copied in and the records of chunk k - 1 are copied out (PCIe is full duplex: 40 bytes per read one way, 16 the
  // other), through pinned staging buffers (a copy from pageable memory would not overlap anything).  The loop it stands
  // for is the reference's read loop (decombine.py:963-1050).
  // (long reads: a chunk's packed bytes stay within what 2 M reads of 150 nt take)
  const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(n, 1), std::max<uint64_t>(1024, std::min<uint64_t>(DCRX_HOST_CHUNK, ((uint64_t)DCRX_HOST_CHUNK * 40) / hb->stride)));
  rc = ensure_device(t, chunk, hb->stride);
  if (rc) return rc;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  // one chunk's staging: packed | lens | exc_read | exc_pos | exc_chr || records | counters    (x 2 sets)
  uint64_t max_exc = 0;       // the most exception entries any chunk holds
  {
    uint64_t e = 0;
    for (uint64_t c0 = 0; c0 < n; c0 += chunk) {
      const uint64_t e0 = e;
      while (e < hb->n_exc && hb->exc_read[e] < c0 + chunk) e++;
      max_exc = std::max(max_exc, e - e0);
    }
  }
  const size_t o_packed = 0;
  const size_t o_lens = o_packed + al(chunk * hb->stride + 16);
  const size_t o_er = o_lens + al(hb->lens ? chunk * 2 : 0);
  const size_t o_ep = o_er + al(max_exc * 4);
  const size_t o_ec = o_ep + al(max_exc * 2);
  const size_t in_bytes = o_ec + al(max_exc);
  const size_t o_rec = in_bytes;
  const size_t o_cnt = o_rec + al(chunk * sizeof(dcrx_record_t));
  const size_t set_bytes = o_cnt + al(DCRX_N_COUNTERS * 8);
  if (2 * set_bytes > t->stage_bytes) {
    (void)hipFree(t->d_stage); t->d_stage = nullptr; t->stage_bytes = 0;
    HIP_TRY(hipMalloc(&t->d_stage, 2 * set_bytes));
    t->stage_bytes = 2 * set_bytes;
  }
  if (2 * set_bytes > t->h_stage_bytes) {
    if (t->h_stage) (void)hipHostFree(t->h_stage);
    t->h_stage = nullptr; t->h_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&t->h_stage, 2 * set_bytes, hipHostMallocDefault));
    t->h_stage_bytes = 2 * set_bytes;
  }
  if (!t->hs_in) {
    HIP_TRY(hipStreamCreateWithFlags(&t->hs_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->hs_run, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->hs_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipEventCreateWithFlags(&t->hev_in[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&t->hev_run[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&t->hev_out[k], hipEventDisableTiming));
    }
  }
  for (int c = 0; c < DCRX_N_COUNTERS; c++) counters[c] = 0;
  // Buffers the caller has pinned (dcrx_malloc_host, hipHostMalloc, hipHostRegister) are copied from and to directly: the
  // staging copies — half of this call's time from pageable memory — fall away for them.
  auto pinned = [](const void *p) {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  };
  const bool in_direct = n && pinned(hb->packed) && pinned(hb->packed + (size_t)n * hb->stride - 1);
  const bool out_direct = n && pinned(records) && pinned(reinterpret_cast<const uint8_t *>(records + n) - 1);
  const uint64_t n_chunks = n ? (n + chunk - 1) / chunk : 1;
  uint64_t exc_at = 0;
  auto drain = [&](uint64_t k) -> int {       // chunk k's records and counters: from the pinned buffer to the caller's
    const int set = (int)(k & 1);
    HIP_TRY(hipEventSynchronize(t->hev_out[set]));
    const uint8_t *h = t->h_stage + (size_t)set * set_bytes;
    const uint64_t c0 = k * chunk, cn = std::min<uint64_t>(chunk, n - c0);
    if (cn && !out_direct) par_memcpy(records + c0, h + o_rec, cn * sizeof(dcrx_record_t));
    const uint64_t *hc = reinterpret_cast<const uint64_t *>(h + o_cnt);
    for (int c = 0; c < DCRX_N_COUNTERS; c++) counters[c] += hc[c];
    return DCRX_OK;
  };
  for (uint64_t k = 0; k < n_chunks; k++) {
    const int set = (int)(k & 1);
    const uint64_t c0 = k * chunk, cn = n ? std::min<uint64_t>(chunk, n - c0) : 0;
    uint8_t *h = t->h_stage + (size_t)set * set_bytes, *d = t->d_stage + (size_t)set * set_bytes;
    // the records of chunk k - 2 leave the set's pinned buffer on a helper thread while this thread fills its input half
    // (the halves do not overlap; the copy out of chunk k - 2 has landed before the helper touches anything)
    int drc = DCRX_OK;
    std::thread helper;
    if (k >= 2) helper = std::thread([&, k] { drc = drain(k - 2); });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_helper{helper};
    // the chunk into the pinned buffer (several threads: one memcpy does not keep up with the link) — once the copy in of
    // chunk k - 2, which read the same bytes, is over
    if (k >= 2) HIP_TRY(hipEventSynchronize(t->hev_in[set]));
    uint64_t e0 = exc_at;
    while (exc_at < hb->n_exc && hb->exc_read[exc_at] < c0 + cn) exc_at++;
    const uint64_t ne = exc_at - e0;
    if (cn && !in_direct) par_memcpy(h + o_packed, hb->packed + c0 * hb->stride, cn * hb->stride);
    if (hb->lens && cn) std::memcpy(h + o_lens, hb->lens + c0, cn * 2);
    uint32_t *her = reinterpret_cast<uint32_t *>(h + o_er);
    for (uint64_t i = 0; i < ne; i++) her[i] = hb->exc_read[e0 + i] - (uint32_t)c0;      // read indices inside the chunk
    if (ne) { std::memcpy(h + o_ep, hb->exc_pos + e0, ne * 2); std::memcpy(h + o_ec, hb->exc_chr + e0, ne); }
    if (helper.joinable()) helper.join();
    if (drc) return drc;
    // copy in (after the kernels that last read this set's device buffers), kernels, copy out
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(t->hs_in, t->hev_run[set], 0));
    if (in_direct) {
      HIP_TRY(hipMemcpyAsync(d + o_packed, hb->packed + c0 * hb->stride, cn * hb->stride, hipMemcpyHostToDevice, t->hs_in));
      if (in_bytes > o_lens) HIP_TRY(hipMemcpyAsync(d + o_lens, h + o_lens, in_bytes - o_lens, hipMemcpyHostToDevice, t->hs_in));
    } else {
      HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, t->hs_in));
    }
    HIP_TRY(hipEventRecord(t->hev_in[set], t->hs_in));
    HIP_TRY(hipStreamWaitEvent(t->hs_run, t->hev_in[set], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(t->hs_run, t->hev_out[set], 0));     // (the records buffer of chunk k - 2 has been copied out)
    dcrx_batch_t db = *hb;
    db.n_reads = cn;
    db.packed = d + o_packed;
    db.lens = hb->lens ? reinterpret_cast<const uint16_t *>(d + o_lens) : nullptr;
    db.n_exc = ne;
    db.exc_read = reinterpret_cast<const uint32_t *>(d + o_er);
    db.exc_pos = reinterpret_cast<const uint16_t *>(d + o_ep);
    db.exc_chr = d + o_ec;
    rc = dcrx_decombine_device(t, cfg, &db, reinterpret_cast<dcrx_record_t *>(d + o_rec), reinterpret_cast<uint64_t *>(d + o_cnt), t->hs_run);
    if (rc) return rc;          // (the caller of this function drains the streams)
    HIP_TRY(hipEventRecord(t->hev_run[set], t->hs_run));
    HIP_TRY(hipStreamWaitEvent(t->hs_out, t->hev_run[set], 0));
    if (out_direct) {
      if (cn) HIP_TRY(hipMemcpyAsync(records + c0, d + o_rec, cn * sizeof(dcrx_record_t), hipMemcpyDeviceToHost, t->hs_out));
      HIP_TRY(hipMemcpyAsync(h + o_cnt, d + o_cnt, set_bytes - o_cnt, hipMemcpyDeviceToHost, t->hs_out));
    } else {
      HIP_TRY(hipMemcpyAsync(h + o_rec, d + o_rec, set_bytes - o_rec, hipMemcpyDeviceToHost, t->hs_out));
    }
    HIP_TRY(hipEventRecord(t->hev_out[set], t->hs_out));
  }
  for (uint64_t k = n_chunks >= 2 ? n_chunks - 2 : 0; k < n_chunks; k++) { rc = drain(k); if (rc) return rc; }
  // (include/dcrx_codes.h: a wave that gave up waiting for another says so in the call's counters — the records would not be
  // complete, and this entry, which has the counters in hand, does not return them as if they were)
  if (counters[DCRX_C_DEVICE_ERRORS]) return set_err(DCRX_E_HIP, "a device-side wait timed out (the fused scan's ring): the records of this call are incomplete");
  return DCRX_OK;
}

// tile counts and offsets of a compaction: a process-wide slot keyed by device (compaction does not need tables)
static int compact_workspace(uint64_t n_reads, uint32_t **tc, uint64_t **to) {
  static thread_local struct { int dev = -1; uint32_t *tc = nullptr; uint64_t *to = nullptr; uint64_t cap = 0; } ws;
  int dev = -1; HIP_TRY(hipGetDevice(&dev));
  if (ws.dev != dev || n_reads > ws.cap) {
    if (ws.dev == dev) { (void)hipFree(ws.tc); (void)hipFree(ws.to); }
    ws.tc = nullptr; ws.to = nullptr; ws.cap = 0; ws.dev = -1;
    const size_t tiles = compact_tiles(n_reads) + 1024;
    HIP_TRY(hipMalloc(&ws.tc, tiles * 4));
    HIP_TRY(hipMalloc(&ws.to, tiles * 8));
    ws.dev = dev; ws.cap = n_reads;
  }
  *tc = ws.tc; *to = ws.to;
  return DCRX_OK;
}

static int compact_hits(const dcrx_record_t *d_records, uint64_t n_reads, uint64_t first_index, dcrx_record_t *d_hits,
                        uint64_t *d_hit_index, uint64_t *d_ok_bitmap, int packed12, uint64_t *d_n_hits, void *stream) {
  if (!d_n_hits || (n_reads && (!d_records || !d_hits || (!d_hit_index && !d_ok_bitmap)))) return set_err(DCRX_E_INVALID, "null argument");
  uint32_t *tc = nullptr; uint64_t *to = nullptr;
  int rc = compact_workspace(n_reads, &tc, &to);
  if (rc) return rc;
  HIP_TRY(launch_compact(d_records, n_reads, first_index, d_hits, d_hit_index, d_ok_bitmap, packed12, d_n_hits, tc, to,
                         (hipStream_t)stream));
  return DCRX_OK;
}

extern "C" {

int dcrx_compact_hits_device(const dcrx_record_t *d_records, uint64_t n_reads, uint64_t first_index,
                             dcrx_record_t *d_hits, uint64_t *d_hit_index, uint64_t *d_n_hits, void *stream) {
  if (n_reads && !d_hit_index) return set_err(DCRX_E_INVALID, "null argument");
  return compact_hits(d_records, n_reads, first_index, d_hits, d_hit_index, nullptr, 0, d_n_hits, stream);
}

int dcrx_compact_hits_bitmap_device(const dcrx_record_t *d_records, uint64_t n_reads, dcrx_record_t *d_hits,
                                    uint64_t *d_ok_bitmap, uint64_t *d_n_hits, void *stream) {
  if (n_reads && !d_ok_bitmap) return set_err(DCRX_E_INVALID, "null argument");
  return compact_hits(d_records, n_reads, 0, d_hits, nullptr, d_ok_bitmap, 0, d_n_hits, stream);
}

int dcrx_compact_hits_packed_device(const dcrx_record_t *d_records, uint64_t n_reads, void *d_tuples12,
                                    uint64_t *d_ok_bitmap, uint64_t *d_n_hits, void *stream) {
  if (n_reads && !d_ok_bitmap) return set_err(DCRX_E_INVALID, "null argument");
  return compact_hits(d_records, n_reads, 0, reinterpret_cast<dcrx_record_t *>(d_tuples12), nullptr, d_ok_bitmap, 1, d_n_hits, stream);
}

int dcrx_compact_hits_packed8_device(const dcrx_record_t *d_records, uint64_t n_reads, void *d_tuples8,
                                     uint64_t *d_ok_bitmap, uint64_t *d_n_hits, void *stream) {
  if (n_reads && !d_ok_bitmap) return set_err(DCRX_E_INVALID, "null argument");
  return compact_hits(d_records, n_reads, 0, reinterpret_cast<dcrx_record_t *>(d_tuples8), nullptr, d_ok_bitmap, 2, d_n_hits, stream);
}

static uint8_t bits_of(uint64_t x) { uint8_t b = 1; while (b < 64 && (x >> b)) b++; return b; }   // bits that hold 0..x (at least one)

int dcrx_tuple_layout(const dcrx_tables_t *t, uint32_t max_read_len, dcrx_tuple_layout_t *L) {
  if (!t || !L) return set_err(DCRX_E_INVALID, "null argument");
  std::memset(L, 0, sizeof *L);
  const GeneHost &V = t->host.g[0], &J = t->host.g[1];
  int64_t max_vdel = 0, max_jdel = 0;
  for (uint32_t k = 0; k < V.n; k++) max_vdel = std::max<int64_t>(max_vdel, (int64_t)V.jumps[k] - (int64_t)V.tags[k].size());
  for (uint32_t k = 0; k < J.n; k++) max_jdel = std::max<int64_t>(max_jdel, (int64_t)J.jumps[k]);
  // (a record's vdel / jdel are bytes: deletions beyond 255 never reach a record)
  L->w_v = bits_of(V.n ? V.n - 1 : 0); L->w_j = bits_of(J.n ? J.n - 1 : 0);
  L->w_vdel = bits_of((uint64_t)std::min<int64_t>(max_vdel, 255)); L->w_jdel = bits_of((uint64_t)std::min<int64_t>(max_jdel, 255));
  L->w_pos = bits_of(max_read_len);
  const uint32_t bits = L->w_v + L->w_j + L->w_vdel + L->w_jdel + 2u * L->w_pos + 2u;
  L->max_read_len = max_read_len;
  if (bits > 64) return set_err(DCRX_E_UNSUPPORTED, "narrow tuple wider than 64 bits");
  L->bits = (uint8_t)bits;
  L->bytes = (uint8_t)std::max<uint32_t>(4, (bits + 7) / 8);
  return DCRX_OK;
}

uint64_t dcrx_tuple_message_bytes(const dcrx_tuple_layout_t *L, uint64_t n_reads, uint64_t n_hits) {
  if (!L) return 0;
  return ((n_reads + 63) / 64) * 8 + n_hits * L->bytes;
}

// the layout as a caller handed it back: what dcrx_tuple_layout would say for these tables
static int layout_dev(dcrx_tables *t, const dcrx_tuple_layout_t *L, TupleLayoutDev *D) {
  dcrx_tuple_layout_t want;
  int rc = dcrx_tuple_layout(t, L->max_read_len, &want);
  if (rc) return rc;
  if (std::memcmp(&want, L, sizeof want) != 0) return set_err(DCRX_E_INVALID, "tuple layout does not belong to these tables");
  D->w_v = L->w_v; D->w_j = L->w_j; D->w_vdel = L->w_vdel; D->w_jdel = L->w_jdel; D->w_pos = L->w_pos; D->bytes = L->bytes;
  D->j_tag_len = t->dev.g[1].tag_len; D->j_jump = t->dev.g[1].jump;
  return DCRX_OK;
}

int dcrx_compact_hits_narrow_device(dcrx_tables_t *t, const dcrx_tuple_layout_t *L, const dcrx_record_t *d_records,
                                    uint64_t n_reads, uint64_t n_slots, void *d_message, uint64_t *d_n_hits, void *stream) {
  if (!t || !L || !d_n_hits || (n_reads && !d_records) || (n_slots && !d_message)) return set_err(DCRX_E_INVALID, "null argument");
  if (n_slots < n_reads) return set_err(DCRX_E_INVALID, "n_slots < n_reads");
  try {
    int rc = ensure_device(t, 0, 40, (hipStream_t)stream);
    if (rc) return rc;
    TupleLayoutDev D;
    rc = layout_dev(t, L, &D);
    if (rc) return rc;
    uint32_t *tc = nullptr; uint64_t *to = nullptr;
    rc = compact_workspace(n_reads, &tc, &to);
    if (rc) return rc;
    HIP_TRY(launch_compact_narrow(d_records, n_reads, static_cast<uint8_t *>(d_message), n_slots, D, d_n_hits, tc, to, (hipStream_t)stream));
  } catch (const std::exception &e) {
    return set_err(DCRX_E_NOMEM, e.what());
  }
  return DCRX_OK;
}

int dcrx_set_tuple_sink(dcrx_tables_t *t, const dcrx_tuple_layout_t *L, void *d_message, uint64_t n_slots, uint64_t *d_n_hits) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  if (!L) { t->sink_on = false; t->sink_msg = nullptr; t->sink_slots = 0; t->sink_total = nullptr; return DCRX_OK; }
  if (!d_message || !d_n_hits || !n_slots) return set_err(DCRX_E_INVALID, "null argument");
  dcrx_tuple_layout_t want;
  int rc = dcrx_tuple_layout(t, L->max_read_len, &want);
  if (rc) return rc;
  if (std::memcmp(&want, L, sizeof want) != 0) return set_err(DCRX_E_INVALID, "tuple layout does not belong to these tables");
  t->sink_on = true; t->sink_layout = *L;
  t->sink_msg = static_cast<uint8_t *>(d_message); t->sink_slots = n_slots; t->sink_total = d_n_hits;
  return DCRX_OK;
}

int dcrx_tune_state(const dcrx_tables_t *t, int orientation, uint64_t n_reads, dcrx_tune_state_t *out) {
  if (!t || !out) return set_err(DCRX_E_INVALID, "null argument");
  *out = dcrx_tune_state_t{0u, 0u, 0.f, 0.f, 0u, 0u};
  const V2Tune &F = t->tune[orientation == DCRX_ORIENT_FORWARD ? 0 : 1];
  out->launch_form = F.last_form;
  out->candidates = n_reads >= V2Tune::BIG_BATCH ? (8192u | (4096u << 16)) : (4096u | (3072u << 16));
  const int k = V2Tune::size_class(n_reads);
  if (k < 0) return DCRX_OK;
  const V2TuneSlot &U = F.slot[k];
  out->rescue_waves = U.choice; out->launches = (uint32_t)U.launches; out->us_first = U.us[0]; out->us_second = U.us[1];
  return DCRX_OK;
}

int dcrx_set_tune_wait(dcrx_tables_t *t, int allow) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  for (V2Tune &U : t->tune) U.may_wait = allow != 0;
  return DCRX_OK;
}

int dcrx_set_reserved_cus(dcrx_tables_t *t, uint32_t n_cus) {
  if (!t) return set_err(DCRX_E_INVALID, "tables is null");
  t->reserved_cus = n_cus;
  t->plan.reserved_cus = n_cus;
  return DCRX_OK;
}

// ---- host-side packing --------------------------------------------------------------
extern "C++" {
// Packs one read: 16 bases per step through a byte table whose top bit marks a byte that is
// not one of "ACGT"; such a word is redone base by base to record the exception entries.
namespace {
struct PackCode {
  uint8_t t[256];
  PackCode() {
    for (int c = 0; c < 256; c++) t[c] = 0x80;
    t[(int)'A'] = 0; t[(int)'C'] = 1; t[(int)'G'] = 2; t[(int)'T'] = 3;
  }
};
const PackCode g_pack_code;

struct ExcEntry { uint32_t read; uint16_t pos; uint8_t chr; };

inline void pack_one(const uint8_t *s, uint32_t len, uint8_t *out, uint32_t stride, uint32_t r, std::vector<ExcEntry> &exc) {
  const uint8_t *t = g_pack_code.t;
  uint32_t *ow = reinterpret_cast<uint32_t *>(out);
  const uint32_t nwords = stride / 4, full = len / 16;
  uint32_t k = 0;
  for (; k < full; k++) {
    const uint8_t *b = s + 16 * k;
    uint32_t w = 0, bad = 0;
    for (int i = 0; i < 16; i++) { const uint32_t c = t[b[i]]; bad |= c; w |= (c & 3u) << (2 * i); }
    if (bad & 0x80u)
      for (int i = 0; i < 16; i++)
        if (t[b[i]] & 0x80u) exc.push_back(ExcEntry{r, (uint16_t)(16 * k + i), b[i]});
    ow[k] = w;
  }
  if (len & 15u) {
    const uint8_t *b = s + 16 * k;
    uint32_t w = 0;
    for (uint32_t i = 0; i < (len & 15u); i++) {
      const uint32_t c = t[b[i]];
      if (c & 0x80u) exc.push_back(ExcEntry{r, (uint16_t)(16 * k + i), b[i]});
      w |= (c & 3u) << (2 * i);
    }
    ow[k++] = w;
  }
  for (; k < nwords; k++) ow[k] = 0;
}

// start/len accessors differ between the two entry points; everything else is shared.  Reads are
// split into contiguous ranges over a few threads; the ranges' exception lists are concatenated
// in range order, which keeps the list sorted by read.
template <class Span>
int64_t pack_all(const char *ascii, const Span &span, uint64_t n_reads, uint32_t stride, uint8_t *packed, uint16_t *lens,
                 uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr, uint64_t exc_cap) {
  if ((n_reads && (!ascii || !packed)) || stride == 0 || (stride & 7u)) return set_err(DCRX_E_INVALID, "bad argument to the read packer");
  if (n_reads > 0xFFFFFFFFull) return set_err(DCRX_E_INVALID, "more than 2^32-1 reads in one batch");
  unsigned nt = 1;
  if (n_reads >= (1u << 16)) {
    nt = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("DCRX_HOST_THREADS")) nt = (unsigned)std::atoi(e);
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
  }
  std::vector<std::vector<ExcEntry>> exc(nt);
  std::vector<int> bad(nt, 0);
  auto work = [&](unsigned k) {
    const uint64_t lo = n_reads * k / nt, hi = n_reads * (k + 1) / nt;
    for (uint64_t r = lo; r < hi; r++) {
      const uint64_t len = span.len(r);
      if (len > 4ull * stride || len > 65535) { bad[k] = 1; return; }
      if (lens) lens[r] = (uint16_t)len;
      pack_one(reinterpret_cast<const uint8_t *>(ascii) + span.start(r), (uint32_t)len, packed + r * (uint64_t)stride, stride,
               (uint32_t)r, exc[k]);
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++) th.emplace_back(work, k);
    for (auto &x : th) x.join();
  }
  for (unsigned k = 0; k < nt; k++) if (bad[k]) return set_err(DCRX_E_INVALID, "read longer than 4*stride");
  uint64_t n_exc = 0;
  for (unsigned k = 0; k < nt; k++)
    for (const ExcEntry &e : exc[k]) {
      if (n_exc < exc_cap && exc_read) { exc_read[n_exc] = e.read; exc_pos[n_exc] = e.pos; exc_chr[n_exc] = e.chr; }
      n_exc++;
    }
  return (int64_t)n_exc;
}
struct SpanOffsets { const uint64_t *o; uint64_t start(uint64_t r) const { return o[r]; } uint64_t len(uint64_t r) const { return o[r + 1] - o[r]; } };
struct SpanStartLen { const uint64_t *s; const uint32_t *l; uint64_t start(uint64_t r) const { return s[r]; } uint64_t len(uint64_t r) const { return l[r]; } };
}  // namespace

}  // extern "C++"

int64_t dcrx_pack_reads(const char *ascii, const uint64_t *offsets, uint64_t n_reads, uint32_t stride,
                        uint8_t *packed, uint16_t *lens, uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr,
                        uint64_t exc_cap) {
  if (n_reads && !offsets) return set_err(DCRX_E_INVALID, "bad argument to dcrx_pack_reads");
  try {
    return pack_all(ascii, SpanOffsets{offsets}, n_reads, stride, packed, lens, exc_read, exc_pos, exc_chr, exc_cap);
  } catch (...) { return set_err(DCRX_E_NOMEM, "out of memory in dcrx_pack_reads"); }
}

int64_t dcrx_pack_reads_span(const char *ascii, const uint64_t *start, const uint32_t *len, uint64_t n_reads, uint32_t stride,
                             uint8_t *packed, uint16_t *lens, uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr,
                             uint64_t exc_cap) {
  if (n_reads && (!start || !len)) return set_err(DCRX_E_INVALID, "bad argument to dcrx_pack_reads_span");
  try {
    return pack_all(ascii, SpanStartLen{start, len}, n_reads, stride, packed, lens, exc_read, exc_pos, exc_chr, exc_cap);
  } catch (...) { return set_err(DCRX_E_NOMEM, "out of memory in dcrx_pack_reads_span"); }
}

int dcrx_unpack_reads(const dcrx_batch_t *b, const uint64_t *offsets, char *ascii) {
  if (!b || !offsets || !ascii) return set_err(DCRX_E_INVALID, "null argument");
  for (uint64_t r = 0; r < b->n_reads; r++) {
    const uint32_t len = b->lens ? b->lens[r] : b->read_len;
    const uint8_t *in = b->packed + r * (uint64_t)b->stride;
    char *o = ascii + offsets[r];
    for (uint32_t i = 0; i < len; i++) o[i] = "ACGT"[(in[i >> 2] >> (2 * (i & 3))) & 3];
  }
  for (uint64_t i = 0; i < b->n_exc; i++) ascii[offsets[b->exc_read[i]] + b->exc_pos[i]] = (char)b->exc_chr[i];
  return DCRX_OK;
}

// ---- synthetic reads ------------------------------------------------------------------
static SynthParams synth_params(const dcrx_synth_cfg_t *c) {
  SynthParams P;
  P.seed = c->seed; P.read_len = c->read_len;
  auto clamp01 = [](double x) { return x < 0 ? 0.0 : (x > 1 ? 1.0 : x); };
  P.p_rearr_u16 = (uint32_t)(clamp01(c->p_rearranged) * 65536.0 + 0.5);
  P.sub_u16 = (uint32_t)(clamp01(c->sub_rate) * 65536.0 + 0.5);
  double nr = clamp01(c->n_rate) * 4294967296.0;
  P.n_u32 = nr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)(nr + 0.5);
  return P;
}

static int check_synth(const dcrx_tables_t *t, const dcrx_synth_cfg_t *c, uint32_t stride) {
  if (!t || !c) return set_err(DCRX_E_INVALID, "null argument");
  if (stride == 0 || (stride & 7u) || c->read_len > 4 * stride) return set_err(DCRX_E_INVALID, "bad stride for read_len");
  if (c->read_len > 65535) return set_err(DCRX_E_INVALID, "read_len too large");
  return DCRX_OK;
}

int dcrx_synth_reads_host(const dcrx_tables_t *t, const dcrx_synth_cfg_t *c, uint64_t first, uint64_t n,
                          uint32_t stride, uint8_t *packed) {
  int rc = check_synth(t, c, stride);
  if (rc) return rc;
  if (n && !packed) return set_err(DCRX_E_INVALID, "packed is null");
  const DevTables T = t->host.resolve(t->host.blob.data());
  const SynthParams P = synth_params(c);
  for (uint64_t i = 0; i < n; i++)
    synth_read(T.g[0], T.g[1], P, first + i, reinterpret_cast<uint32_t *>(packed + i * (uint64_t)stride), stride / 4);
  return DCRX_OK;
}

int64_t dcrx_synth_exceptions_host(const dcrx_tables_t *t, const dcrx_synth_cfg_t *c, uint64_t first, uint64_t n,
                                   uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr, uint64_t cap) {
  if (!t || !c) return set_err(DCRX_E_INVALID, "null argument");
  const SynthParams P = synth_params(c);
  uint64_t cnt = 0;
  if (P.n_u32 == 0 || P.read_len == 0) return 0;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t key = synth_mix64(P.seed ^ synth_mix64(first + i));
    const uint64_t dn = synth_draw(key, 8);
    if ((uint32_t)(dn & 0xFFFFFFFFu) < P.n_u32) {
      if (cnt < cap && exc_read) {
        exc_read[cnt] = (uint32_t)i; exc_pos[cnt] = (uint16_t)((dn >> 32) % (uint64_t)P.read_len); exc_chr[cnt] = 'N';
      }
      cnt++;
    }
  }
  return (int64_t)cnt;
}

int dcrx_synth_reads_device(dcrx_tables_t *t, const dcrx_synth_cfg_t *c, uint64_t first, uint64_t n, uint32_t stride,
                            uint8_t *d_packed, void *stream) {
  int rc = check_synth(t, c, stride);
  if (rc) return rc;
  if (n && !d_packed) return set_err(DCRX_E_INVALID, "d_packed is null");
  rc = ensure_device(t, 0);
  if (rc) return rc;
  HIP_TRY(launch_synth(t->dev, synth_params(c), first, n, stride, d_packed, (hipStream_t)stream));
  return DCRX_OK;
}

}  // extern "C"
