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
};

hipError_t launch_decombine(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg,
                            dcrx_record_t *rec, uint32_t *queue, uint32_t *gqueue, uint32_t *queue_count,
                            uint64_t *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop);
hipError_t launch_compact(const dcrx_record_t *rec, uint64_t n, uint64_t first_index, dcrx_record_t *hits,
                          uint64_t *hit_index, uint64_t *ok_bitmap, int packed12, uint64_t *d_total, uint32_t *tile_count,
                          uint64_t *tile_off, hipStream_t s);
uint32_t compact_tiles(uint64_t n);

}  // namespace dcrx
