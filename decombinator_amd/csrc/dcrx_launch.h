// dcrx_launch.h — what dcrx_api.cpp (host) and dcrx_kernels.hip share.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcrx.h"
#include "dcrx_device.h"

#include "dcrx_launch_types.h"

namespace dcrx {

// the narrow tuple (include/dcrx.h, dcrx_tuple_layout) as the kernels take it
struct TupleLayoutDev {
  uint32_t w_v, w_j, w_vdel, w_jdel, w_pos, bytes;
  const uint8_t *j_tag_len;    // len(j_seqs[k])
  const int32_t *j_jump;       // jump_to_start_j[k]
};
// The tuple sink of a handle (dcrx_sink_device.h).  V2SinkDev lives in device memory and stays as it is while the sink is on;
// V2SinkCall travels with a launch (dev == nullptr: no sink).
struct V2SinkDev {
  uint32_t *late;              // [regions] items of the late section (zero between calls)
  uint32_t *ticket;            // blocks of the place kernel that have read the counts (zero between calls)
  const uint8_t *j_tag_len;
  const int32_t *j_jump;
};
struct V2SinkCall {
  const V2SinkDev *dev;
  uint2 *items;                // [regions][stride]: the tuple's low word | the read's index inside its region (24 bits; V2_SINK_EMPTY: no tuple) and the tuple's bits 32-39
  uint32_t *hits;              // [regions] decombined reads (zero between calls)
  uint32_t stride;             // items of a region's slab: sections tail (at 0), E, C, late
  uint32_t e_off, c_off, late_off, late_cap;
  uint32_t per_block;          // reads of a region
  uint32_t wpack;              // w_v | w_j << 5 | w_vdel << 10 | w_jdel << 15 | w_pos << 20
};
// what a launch is asked to leave in the sink, and whether it did (else the caller compacts the records)
struct V2SinkJob {
  const V2SinkDev *dev = nullptr;
  uint2 *items = nullptr; uint32_t *hits = nullptr;
  uint64_t items_cap = 0;      // items allocated
  uint32_t regions_cap = 0;    // regions the counters hold
  uint32_t wpack = 0, bytes = 0;
  uint8_t *msg = nullptr; uint64_t n_slots = 0; uint64_t *d_total = nullptr;
  bool *done = nullptr;
};

// A handle's own choice of the waves that share a region's list E in the finishing launch of the fused form (launch_v2): 16 per
// region suit some tag sets and 12 others by 1-4 % of the step, whatever the reads are, so a handle
// times its own launches: behind its first launch of a batch size two finishing launches on 16 and two on 12 in turn carry
// a pair of events on their dispatch (no marker packets), later launches look (hipEventQuery: no waiting) whether the pairs
// have completed, and the setting whose launches were at least 1.5 % shorter on average stays; else 16.  One per frame.
struct V2TuneSlot {
  static constexpr int SAMPLES = 2;      // (one per setting — the two differ by a tenth of the launch, samples by a hundredth —: launches 1 and 2 of a size class, so that a caller's third or fourth launch finds them complete and runs on the choice)
  uint32_t choice = 0;               // rescue waves once settled (0: not yet)
  int launches = 0;                  // launches seen in this size class
  hipEvent_t ev[SAMPLES][2] = {};    // (start, stop) of the finishing launch of sample k
  bool created = false;
  // list E inside the scan kernel (FUSE_E) or a role of the finishing launch: decided once per size class from the share of the
  // reads that were list-E entries in the class's first launch (the regions' counts copied to pinned memory behind that launch,
  // read when the copy's event has passed: no wait)
  int fuse_e = -1;                   // -1 not known yet, -2 the share allows it: the two forms are being timed, 0 a role, 1 inside the scan
  static constexpr int E_PAIRS = 3, E_FIRST = 8;      // pairs of timed launches (a role, fused), from the class's E_FIRST-th eligible launch on (the clocks have come up by then)
  int e_phase = 0;                   // (fuse_e == -2) eligible launches seen: E_FIRST + 2 k runs as a role under a pair of events, E_FIRST + 2 k + 1 fused; then the events are read
  hipEvent_t ev_e[2 * E_PAIRS][2] = {};      // (start on the scan's dispatch, stop on the finishing launch's) per timed launch
  float us_e[2] = {0.f, 0.f};        // what the samples said: mean of the launches with list E a role / inside the scan
  bool e_sampling = false;
  hipEvent_t ev_counts = nullptr;
  uint32_t *h_counts = nullptr;      // pinned, V2_L_COUNTS words per region
  uint32_t e_regions = 0;
  uint64_t e_reads = 0;
  float e_share = -1.f;
  float us[2] = {0.f, 0.f};          // what the samples said: a finishing launch on 4 096 / on 3 072 rescue waves (dcrx_tune_state)
};
// ... per size class (batches of 2^(20 + k) .. 2^(21 + k) - 1 reads share a slot): a short last chunk of a host call, or the
// short last step of a shard, falls into another class and leaves the settled one alone.
struct V2Tune {
  static constexpr int SAMPLES = V2TuneSlot::SAMPLES;
  static constexpr int CLASSES = 12;
  static constexpr uint64_t BIG_BATCH = 1ull << 25;      // reads: from here a handle chooses between 8 192 and 4 096 rescue waves (below: 4 096 and 3 072)
  V2TuneSlot slot[CLASSES];
  bool may_wait = false;             // dcrx_set_tune_wait: the fourth call of a big-batch size class may wait for the third's finishing launch, once
  uint32_t last_form = 0;            // the frame's last call: 0 none yet, 1 the three-launch form, 2 the v2 kernels (tail as a role), 3 v2 with the tail inside the scan
  static int size_class(uint64_t n_reads) {      // -1: below a million reads (not tuned)
    if (n_reads < (1ull << 20)) return -1;
    int k = 0;
    while (k + 1 < CLASSES && (n_reads >> (21 + k)) != 0) k++;
    return k;
  }
};

struct LaunchPlan {
  uint32_t n_cu;
  uint32_t grid;   // fast kernel (upper bound; capped by measured occupancy at launch)
  uint32_t qgrid;  // list kernel
  uint32_t lds_bytes;
  bool table_in_lds;
  bool table16_in_lds;       // the two-bases-per-step table + side tables fit beside the fast kernel's buffers
  uint32_t lds16_bytes;      // counters + that table + side tables
  uint32_t reserved_cus = 0; // compute units the persistent grids leave free (dcrx_set_reserved_cus)
  const DevTables *dev_tables = nullptr;   // DevTables in device memory
  // v2 kernels: the per-wave lists between the scan and the finishing kernel (device memory of the tables handle)
  uint4 *v2_tail = nullptr; uint4 *v2_events = nullptr; uint32_t *v2_counts = nullptr;
  uint4 *v2_slow = nullptr;
  uint4 *v2_left = nullptr;             // the finishing launch's left list (V2_LEFT_CAP event entries of the longest shape)
  uint64_t *v2_acc = nullptr;           // the call's tallies (uint64[DCRX_N_COUNTERS]): zero between calls, handed to the caller by the list kernel
  uint64_t v2_tail_rows = 0, v2_event_rows = 0, v2_slow_rows = 0;       // 16-byte rows allocated for each list
  hipStream_t v2_side = nullptr, v2_side2 = nullptr;  // the tail kernel / the general form over the reads with exception bytes run here, beside the rescue kernel
  hipEvent_t v2_ev_fork = nullptr, v2_ev_join = nullptr, v2_ev_join2 = nullptr;
  // optional, set per call (dcrx_set_step_events): start of the first and end of the last kernel of the call.  Attached to
  // those kernels' own dispatches (hipExtLaunchKernelGGL): a separate event record costs the stream ~10 us of gap each
  hipEvent_t ev_step_start = nullptr, ev_step_stop = nullptr;
  V2SinkJob sink;              // set per call while a tuple sink is on (dcrx_set_tuple_sink)
  V2Tune *tune = nullptr;      // [2]: per frame (the handle owns them)
};

hipError_t launch_decombine(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg,
                            dcrx_record_t *rec, uint32_t *queue, uint32_t *gqueue, uint32_t *queue_count,
                            uint64_t *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
// dcrx_kernels_v2.hip
void v2_list_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu, uint64_t *tail_rows, uint64_t *event_rows);
uint64_t v2_slow_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu);
bool v2_applies(const LaunchPlan &P, const DevTables &T, const CfgDev &cfg, uint32_t stride);
// what a v2 launch that serves the call's tuple sink leaves for the list kernel and the place kernel behind it
struct V2SinkLaunch { V2SinkCall S{}; uint32_t n_regions = 0, tcap = 0, ecap = 0, ccap = 0, fused = 0; const uint32_t *counts = nullptr; };
hipError_t launch_v2_any(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                         uint32_t *queue, uint32_t *gqueue, uint32_t qcap, uint32_t *queue_count, unsigned long long *d_counters,
                         hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t retry = 0, V2SinkLaunch *sink = nullptr);
hipError_t launch_v2_place(const LaunchPlan &P, const V2SinkLaunch &K, uint64_t n_reads, hipStream_t s, hipEvent_t ev_stop);
uint64_t v2_sink_items(uint64_t max_reads, uint32_t n_cu);      // items a handle's sink needs for batches of up to max_reads reads
hipError_t launch_compact(const dcrx_record_t *rec, uint64_t n, uint64_t first_index, dcrx_record_t *hits,
                          uint64_t *hit_index, uint64_t *ok_bitmap, int packed12, uint64_t *d_total, uint32_t *tile_count,
                          uint64_t *tile_off, hipStream_t s);
hipError_t launch_compact_narrow(const dcrx_record_t *rec, uint64_t n, uint8_t *msg, uint64_t n_slots, const TupleLayoutDev &L, uint64_t *d_total,
                                 uint32_t *tile_count, uint64_t *tile_off, hipStream_t s);
uint32_t compact_tiles(uint64_t n);

}  // namespace dcrx
