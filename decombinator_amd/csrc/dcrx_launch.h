// dcrx_launch.h — what dcrx_api.cpp (host) and dcrx_kernels.hip share.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcrx.h"
#include "dcrx_device.h"

#include "dcrx_launch_types.h"

namespace dcrx {

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
};

hipError_t launch_decombine(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg,
                            dcrx_record_t *rec, uint32_t *queue, uint32_t *gqueue, uint32_t *queue_count,
                            uint64_t *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
// dcrx_kernels_v2.hip
void v2_list_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu, uint64_t *tail_rows, uint64_t *event_rows);
uint64_t v2_slow_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu);
bool v2_applies(const LaunchPlan &P, const DevTables &T, const CfgDev &cfg);
hipError_t launch_v2_any(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                         uint32_t *queue, uint32_t *gqueue, uint32_t qcap, uint32_t *queue_count, unsigned long long *d_counters,
                         hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t retry = 0);
hipError_t launch_compact(const dcrx_record_t *rec, uint64_t n, uint64_t first_index, dcrx_record_t *hits,
                          uint64_t *hit_index, uint64_t *ok_bitmap, int packed12, uint64_t *d_total, uint32_t *tile_count,
                          uint64_t *tile_off, hipStream_t s);
uint32_t compact_tiles(uint64_t n);

}  // namespace dcrx
