// dcrx_debug_flags.h — private: test and profiling switches carried in dcrx_cfg_t.flags (include/dcrx.h documents only
// DCRX_F_NONE).  A/B switches leave the records unchanged; DCRX_F_PROFILE_* make a kernel stop half-way — the records are
// then NOT results, and dcrx_decombine_device refuses them unless the environment holds DCRX_DEBUG_FLAGS=1.
#pragma once

#define DCRX_F_FORCE_SLOW_READER 1u /* tests: route every read through the exception-aware reader */
#define DCRX_F_PROFILE_SCAN_ONLY 2u  /* profiling: fast kernel stops after the DFA scan (records are NOT results) */
#define DCRX_F_PROFILE_LIST_SCAN_ONLY 8u /* profiling: list kernel stops after its collecting scan (records are NOT results) */
#define DCRX_F_PROFILE_RESCUE_HITS_ONLY 32u /* profiling: rescue kernel stops after resolving the half-tag hit lists (records are NOT results) */
#define DCRX_F_LIST_RESCUE 16u       /* rescue queue through the list kernel (one-base collecting scan) even when the pair form applies (A/B, tests) */
#define DCRX_F_PROFILE_NO_FINISH 128u /* profiling: the v2 kernel scans and sorts reads onto its stacks but finishes none of them (records are NOT results) */
#define DCRX_F_PROFILE_NO_EVENTS 1024u /* profiling: the v2 kernel finishes its tail entries but drops its event entries (records are NOT results) */
#define DCRX_F_PROFILE_NO_TAIL 2048u /* profiling: the v2 finishing kernel skips its tail entries (records are NOT results) */
#define DCRX_F_PROFILE_TAIL_STREAM_ONLY 16384u /* profiling: the tail kernel reads its entries and writes records but resolves nothing (records are NOT results) */
#define DCRX_F_V2_LEAN_SERIAL 32768u /* A/B: the lean rescue, the lean tail and the X pass as launches of their own, one after the other on the caller's stream (default: roles of one launch, finish2_kernel) */
#define DCRX_F_V2_NO_FUSE 131072u /* A/B: the tail as a role of the finishing launch even where the scan kernel could take it (scan2_kernel, FUSE) */
#define DCRX_F_V2_SIDE_STREAMS 65536u /* A/B: round 3's shape — the tail kernel and the X pass on side streams of the handle beside the rescue kernel (fork / join events) */
#define DCRX_F_V2_NO_LEAN_RESCUE 8192u /* A/B: the scan kernel's event entries go to the general form at once, without the lean rescue kernel */
#define DCRX_F_V1_KERNELS 64u         /* the three-launch form (fast kernel with 32-bit pair entries, rescue kernel) even where the v2 kernel applies (A/B, tests) */
#define DCRX_F_V2_SHAPE(k) ((uint32_t)(k) << 8) /* v2 kernel launch shape, A/B: 0 default, 2 = two reads per lane, 3 = one read per lane (one 1024-thread block per CU either way) */
#define DCRX_F_ONE_BASE_SCAN 4u      /* use the one-base-per-step fast kernel even when the two-base table fits LDS (A/B, tests) */


/* the switches after which records are not results */
#define DCRX_F_PROFILE_MASK (DCRX_F_PROFILE_SCAN_ONLY | DCRX_F_PROFILE_LIST_SCAN_ONLY | DCRX_F_PROFILE_RESCUE_HITS_ONLY | DCRX_F_PROFILE_NO_FINISH | \
                             DCRX_F_PROFILE_NO_EVENTS | DCRX_F_PROFILE_NO_TAIL | DCRX_F_PROFILE_TAIL_STREAM_ONLY)

// The A/B and developer switches the library reads from the environment (DCRX_DEBUG_*, DCRX_STAMPS_DUMP) are honoured only while
// DCRX_DEBUG_FLAGS=1 is set, like the cfg flags above: a production process cannot be steered by a stray variable.
#include <cstdlib>
static inline const char *dcrx_debug_env(const char *name) {
  static const bool on = [] { const char *e = std::getenv("DCRX_DEBUG_FLAGS"); return e && e[0] == '1'; }();
  return on ? std::getenv(name) : nullptr;
}
