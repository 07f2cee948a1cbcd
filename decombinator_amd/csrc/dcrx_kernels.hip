// dcrx_kernels.hip — the `decombine` hot path as HIP kernels for gfx950 (CDNA4).
//
// One read per lane.  The merged Aho-Corasick automaton (V tags, J tags and the
// four half-tag sets: the six acora automata of reference
// src/decombinator/decombine.py:722-746) lives in LDS as a goto-only DFA whose
// entries carry the output classes of their target state.  Reads arrive 2-bit
// packed (A0 C1 G2 T3) and are never reverse-complemented in memory: the
// `reverse` frame (decombine.py:1000, revcomp :182-184) is read by walking the
// packed words backwards and inverting the bits, and germline windows are
// compared against a pre-reverse-complemented packed copy of the region.
//
// Per read (decombine.py:534-585, dcr()):
//   scan    one pass of the DFA over the frame, two bases per LDS look-up when the
//           pair table fits: V/J full-tag hit count, the single hit's location, and
//           "any half-tag hit" flags (replaces the findall() calls at :275, :399 and
//           tells whether the ones at :294, :339, :422, :473 would be non-empty)
//   V       full hit -> get_v_deletions walk (:749-785); no hit -> half-tag rescue:
//           candidates in findall order, tested with Hamming <= 1 (:294-390)
//   J       same (:397-531, walk :788-817), only when V succeeded (:544-548)
//   filters :553-569, then the 16-byte record
//
// Three launches per batch:
//   prologue_kernel          zeroes the counters, marks / lists the reads with non-ACGT bytes
//   decombine_kernel         every clean read: scan; reads with one V tag are ballot-compacted
//                            and finished in full waves; reads that need a half-tag rescue are
//                            queued
//   decombine_rescue_kernel  the queue and the listed reads (decombine_list_kernel instead when
//                            there is no pair table, for orientation `both`, or on request)
//
// Integer/byte work only; no MFMA.  Bound: VALU issue and LDS look-ups of the scan;
// HBM carries the packed reads (40 B) and the records (16 B).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../../include/dcrx.h"
#include "dcrx_device.h"
#include "dcrx_launch.h"
#include "dcrx_dcr_device.h"
#include "dcrx_sink_device.h"

namespace dcrx {

// LDS beyond the counters + DFA (LaunchPlan::lds_bytes): the fast kernel's per-wave deferral
// buffers, the list kernel's half-tag hit lists
constexpr int DCRX_WQ_CAP = 128;
constexpr int DCRX_GTILE = 8;  // general-list reads per ticket (one wave)
constexpr int DCRX_CHUNK = 2;  // 64-read tiles of the rescue queue a wave claims per ticket
constexpr uint32_t DCRX_FAST_LDS_EXTRA = (DCRX_BLOCK / 64) * DCRX_WQ_CAP * 4;
constexpr int DCRX_TQ_CAP = 128;  // tail entries per wave (flushed at 64; 3 dwords each)
constexpr uint32_t DCRX_FAST16_LDS_EXTRA = (DCRX_BLOCK16 / 64) * (DCRX_WQ_CAP + 3 * DCRX_TQ_CAP) * 4;  // 32 KB; keeps 64-byte rows aligned after the 128-byte counters
static_assert((DCRX_N_COUNTERS * 4 + DCRX_FAST16_LDS_EXTRA) % 64 == 0, "pair-scan rows must be 64-byte aligned");
constexpr int DCRX_LSLOT = (HH_STRIDE + DCRX_GSLOT_EXTRA) | 1;  // per-lane dwords: hit lists + exception copy; odd: conflict-free
constexpr int DCRX_LSLOT_PAD = (4 - (DCRX_N_COUNTERS + DCRX_QBLOCK * DCRX_LSLOT) % 4) % 4;
constexpr uint32_t DCRX_QUEUE_LDS_EXTRA = (DCRX_QBLOCK * DCRX_LSLOT + DCRX_LSLOT_PAD) * 4;

// ------------------------------------------------------------------------------
// Fast kernel: persistent blocks, each stages the DFA into LDS once and then
// strides over tiles of DCRX_BLOCK reads.  Reads that need the general path
// (half-tag rescue) are compacted with a wavefront ballot into a queue of read
// indices for the rescue kernel, so that the rare, long, divergent work runs in
// dense waves instead of stalling this one; reads with exception bytes are
// skipped here (the prologue put them on the general list).
// ------------------------------------------------------------------------------
template <bool TABLE_LDS, bool UNIFORM_LEN, int NW, int ARITY>
__global__ __launch_bounds__(ARITY == 16 ? DCRX_BLOCK16 : DCRX_BLOCK) void decombine_kernel(
    DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ queue, uint32_t *__restrict__ queue_count) {
  constexpr int BLOCK = ARITY == 16 ? DCRX_BLOCK16 : DCRX_BLOCK;
  extern __shared__ __align__(64) uint32_t smem[];
  uint32_t *lds_counts = smem;                        // [DCRX_N_COUNTERS]
  uint32_t *lds_wq = smem + DCRX_N_COUNTERS;          // [waves][DCRX_WQ_CAP]
  uint32_t *lds_tq = lds_wq + (BLOCK / 64) * DCRX_WQ_CAP;     // [waves][3 * DCRX_TQ_CAP] tail entries (pair scan only)
  uint32_t *lds_trans = lds_tq + (ARITY == 16 ? (BLOCK / 64) * 3 * DCRX_TQ_CAP : 0);  // the DFA (rows of 16 or 64 bytes), then the side tables
  const int tid = threadIdx.x;
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  DevTables T = T0;
  if (TABLE_LDS) {
    const uint32_t lds_addr = dcrx_lds_address(reinterpret_cast<const uint8_t *>(lds_trans));
    if (ARITY == 16) {
      uint32_t *lds_side = lds_trans + T0.dfa16_bytes / 4;
      stage_lds<BLOCK>(reinterpret_cast<const uint8_t *>(T0.trans16), lds_trans, T0.dfa16_bytes / 16, T0.dfa16_bytes / 16,
                       lds_addr, tid);
      stage_lds<BLOCK>(T0.image + T0.dfa_bytes, lds_side, (T0.lds_image_bytes - T0.dfa_bytes) / 16, 0, 0, tid);
      T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_side), T0.dfa_bytes);
      T.row16_0 = lds_addr;
    } else {
      stage_lds<BLOCK>(T0.image, lds_trans, T0.lds_image_bytes / 16, T0.dfa_bytes / 16, lds_addr, tid);
      T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_trans), 0);
      T.row0 = lds_addr;
    }
  }
  __syncthreads();
  const Counters C{lds_counts};
  const uint32_t nw = B.stride >> 2;
  const int lane = tid & 63;
  // per-wave staging of deferred read indices: flushed to the global queue 64+ at a time, so
  // that the queue tail sees one atomic per ~100 deferred reads instead of one per wave and tile
  uint32_t *wq = lds_wq + (tid >> 6) * DCRX_WQ_CAP;
  uint32_t wq_n = 0;

  // Static work distribution: block b takes tiles b, b + grid, ... (the grid is exactly the
  // resident capacity, so every block runs from the start; a global ticket per wave-tile was
  // measured slower: one atomic address sustains only ~90 tickets/us).
  // Queue entries carry two hints in their top bits when the batch allows (< 2^30 reads): bit 31 =
  // the V tag needs the rescue, bit 30 = the J tag may need it.  The rescue kernel then only
  // marks and resolves the half-tag hits of the gene(s) concerned.  Hints err on the side of 1.
  const bool tagged = B.n_reads < (1ull << 30);
  auto flush = [&]() {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(queue_count, wq_n);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < wq_n; i += 64) queue[base + i] = wq[i];
    wq_n = 0;
  };
  auto to_rescue = [&](bool defer, uint32_t r32, uint32_t hints) {
    const unsigned long long m = __ballot(defer);
    if (m) {
      if (defer) wq[wq_n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = tagged ? (r32 | (hints << 30)) : r32;
      wq_n += (uint32_t)__popcll(m);
      if (wq_n >= 64) flush();
    }
  };
  // Pair scan, uniform even read length: the tail (hit location, walks, filters) is batched.
  // Reads with exactly one V tag are ballot-compacted, tile after tile, into this wave's LDS
  // buffer of packed accumulators; whenever 64 are waiting the tail runs on a full wave.
  const bool batched = ARITY == 16 && UNIFORM_LEN && !(B.read_len & 1u);
  if (ARITY == 16 && batched) {
    uint32_t *tq = lds_tq + (tid >> 6) * (3 * DCRX_TQ_CAP);
    uint32_t tq_n = 0;
    auto run_tail = [&](uint32_t first, uint32_t count) {   // entries [first, first+count) of this wave's buffer
      bool defer = false;
      uint32_t r32 = 0;
      if ((uint32_t)lane < count) {
        const uint32_t *en = tq + 3 * (first + lane);
        const TailEntry te{en[0], en[1], en[2]};
        r32 = te.r;
        defer = fast16_tail_one<TABLE_LDS, UNIFORM_LEN>(T, lds_trans, B, cfg, (uint64_t)te.r, tail_unpack(te, T.row16_0), C,
                                                        records) == FAST_TO_RESCUE;
      }
      to_rescue(defer, r32, 1u);      // a read with its V tag in hand: only the J tag needs the rescue
    };
    for (uint64_t tile = blockIdx.x; tile * BLOCK < B.n_reads; tile += gridDim.x) {
      const uint64_t r = tile * BLOCK + tid;
      bool live = r < B.n_reads;
      if (live && B.n_exc && ((B.exc_flag[r >> 5] >> (r & 31)) & 1u)) live = false;   // on the general list already
      ScanAcc16 a{0, 0, 0, 0};
      if (live) a = fast16_scan_one<TABLE_LDS, UNIFORM_LEN, NW>(T, B, cfg, r, nw);
      const bool one_v = live && !(cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) && DCRX_ACC16_COUNT(a.vacc) == 1 &&
                         !((a.acc >> TE_VMULTI_BIT) & 1u);
      // No single V tag: the outcome needs no walk.  No V tag and no V half tag -> NoVDetected
      // (decombine.py:393); several V tags -> MultipleVtagMatches (:278-280); V half tags only ->
      // rescue queue.  Tallies go through one ballot per outcome and wave.
      bool defer = false;
      if (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) {
        if (live) fast16_tail_one<TABLE_LDS, UNIFORM_LEN>(T, lds_trans, B, cfg, r, a, C, records);
      } else {
        const uint32_t vcnt = DCRX_ACC16_COUNT(a.vacc);
        const bool vmulti = live && !one_v && (vcnt > 1 || ((a.acc >> TE_VMULTI_BIT) & 1u) || ((a.acc >> TE_VFULL_BIT) & 1u));
        const bool vhalf = live && !one_v && !vmulti && ((DCRX_ACC16_FLAGS(a.acc) >> TE_VH1_BIT) & 3u);
        const bool vnone = live && !one_v && !vmulti && !vhalf;
        defer = vhalf;
        if (vnone || vmulti) {
          __align__(16) dcrx_record_t rec;
          rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
          rec.vdel = rec.jdel = 0;
          rec.status = (uint8_t)(vnone ? DCRX_S_V_NONE : DCRX_S_V_MULTI);
          rec.frame = (uint8_t)(cfg.orientation == DCRX_ORIENT_FORWARD ? 1 : 0);
          dcrx_store_record(records + r, rec);
        }
        const unsigned long long mn = __ballot(vnone), mm = __ballot(vmulti);
        if (lane == 0) {
          if (mn) atomicAdd(&lds_counts[DCRX_C_NO_VTAGS_FOUND], (uint32_t)__popcll(mn));
          if (mm) atomicAdd(&lds_counts[DCRX_C_MULTIPLE_V_MATCHES], (uint32_t)__popcll(mm));
          if (mn | mm) atomicAdd(&lds_counts[DCRX_C_READ_COUNT], (uint32_t)__popcll(mn | mm));
        }
      }
      // V needs the rescue; J may need it unless exactly one J tag was seen
      to_rescue(defer, (uint32_t)r, 2u | (DCRX_ACC16_COUNT(a.jacc) != 1u ? 1u : 0u));
      const unsigned long long m = __ballot(one_v);
      if (m) {
        if (one_v) {
          const TailEntry te = tail_pack((uint32_t)r, a);
          uint32_t *en = tq + 3 * (tq_n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
          en[0] = te.r; en[1] = te.a; en[2] = te.b;
        }
        tq_n += (uint32_t)__popcll(m);
        if (tq_n >= 64) { tq_n -= 64; run_tail(tq_n, 64); }
      }
    }
    if (tq_n) run_tail(0, tq_n);
  } else {
  for (uint64_t tile = blockIdx.x; tile * BLOCK < B.n_reads; tile += gridDim.x) {
    const uint64_t r = tile * BLOCK + tid;
    int what = FAST_DONE;
    if (r < B.n_reads) what = decombine_fast_one<TABLE_LDS, UNIFORM_LEN, NW, ARITY>(T, lds_trans, B, cfg, r, nw, C, records);
    // FAST_TO_GENERAL reads (exception bytes) are already on the general list, which is built
    // from the exception list before this kernel starts
    to_rescue(what == FAST_TO_RESCUE, (uint32_t)r, 3u);
  }
  }
  if (wq_n) flush();
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
}

// Entry k of rescue tile `tile` (per_tile entries each), 0xFFFFFFFF past the end.  With `tagged`
// the top two bits are the fast kernel's hints (see there); *hints gets them (3 when untagged).
__device__ __forceinline__ uint32_t rescue_entry(const uint32_t *__restrict__ queue, uint32_t n_rescue, uint32_t tile,
                                                 uint32_t per_tile, uint32_t k, bool tagged, uint32_t *hints) {
  const uint32_t i = tile * per_tile + k;
  *hints = 3u;
  if (i >= n_rescue) return 0xFFFFFFFFu;
  const uint32_t ent = queue[i];
  if (!tagged) return ent;
  *hints = ent >> 30;
  return ent & 0x3FFFFFFFu;
}

// List kernel: everything the fast kernel does not finish, in dense waves.  Two work lists,
// claimed 64 reads at a time through one ticket: first the general list (reads with exception
// bytes, or every read for orientation `both` / the forced slow reader — built from the
// exception list before the fast kernel starts; they are the longest poles, so they go first),
// then the rescue queue the fast kernel filled (clean reads whose V or J tag needs the
// half-tag rescue).  One collecting scan per frame (half-tag hits into per-lane LDS lists),
// then the rescue feeds from the lists.
template <bool TABLE_LDS, bool UNIFORM_LEN>
__global__ __launch_bounds__(DCRX_QBLOCK) void decombine_list_kernel(DevTables T0, BatchDev B, CfgDev cfg,
                                                                     dcrx_record_t *__restrict__ records,
                                                                     unsigned long long *__restrict__ counters,
                                                                     const uint32_t *__restrict__ queue,
                                                                     const uint32_t *__restrict__ gqueue,
                                                                     uint32_t *__restrict__ queue_count,
                                                                     unsigned long long *__restrict__ out,
                                                                     unsigned long long read_count, V2SinkCall S) {
  // `out` (behind the v2 kernels, which tally into an accumulator of the handle: `counters`): the call's counters are handed
  // to the caller here and the accumulator is left zeroed for the next call — by block 0 when nothing was handed over (the
  // other blocks leave at once, without an atomic), else by the last block to finish.  read_count (orientation `both`): a
  // read counts once however many frames it was tried in (counts["read_count"], decombine.py:991).
  uint32_t *tile_ticket = queue_count + 3;
  extern __shared__ __align__(16) uint32_t smem[];
  uint32_t *lds_counts = smem;
  uint32_t *lds_slots = smem + DCRX_N_COUNTERS;        // [DCRX_QBLOCK][DCRX_LSLOT]: hit lists + exception copy
  uint32_t *lds_trans = lds_slots + DCRX_QBLOCK * DCRX_LSLOT + DCRX_LSLOT_PAD;
  static_assert(((DCRX_N_COUNTERS + DCRX_QBLOCK * DCRX_LSLOT + DCRX_LSLOT_PAD) % 4) == 0, "DFA rows must stay 16-byte aligned");
  const int tid = threadIdx.x;
  const uint32_t n_rescue = queue_count[0], n_general = queue_count[1];
  const bool tagged = B.n_reads < (1ull << 30);
  // tickets: one per DCRX_GTILE general reads (few lanes per wave: these reads take long, divergent
  // paths, and a wave runs the union of its lanes' paths), one per DCRX_CHUNK*64 rescue reads
  const uint32_t t_general = (n_general + DCRX_GTILE - 1) / DCRX_GTILE;
  const uint32_t t_rescue = (n_rescue + 64 * DCRX_CHUNK - 1) / (64 * DCRX_CHUNK);
  // The last block to finish re-arms the work counters for the next launch (no memset between
  // launches): by then every block has read the counts above.
  auto hand_over = [&](const int c) {      // (a thread per counter, when every tally of the call is in the accumulator)
    const unsigned long long v = atomicExch(&counters[c], 0ull);
    out[c] = (c == DCRX_C_READ_COUNT && read_count != ~0ull) ? read_count : v;
  };
  if (out && n_rescue == 0 && n_general == 0) {      // nothing was handed over: the common case behind the v2 kernels
    if (blockIdx.x == 0 && tid < DCRX_N_COUNTERS) hand_over(tid);
    return;
  }
  auto leave = [&]() {      // (the block's first wave: its lanes hold the block's tallies, lane 0 signs the block off)
    if (tid >= 64) return;
    uint32_t last = 0;
    if (tid == 0) last = atomicAdd(queue_count + 4, 1u) == gridDim.x - 1 ? 1u : 0u;
    last = (uint32_t)__shfl((int)last, 0);
    if (!last) return;
    if (out && tid < DCRX_N_COUNTERS) hand_over(tid);
    if (tid == 0) {
      queue_count[0] = queue_count[1] = queue_count[2] = queue_count[3] = 0;
      __threadfence();
      queue_count[4] = 0;
    }
  };
  if ((uint64_t)blockIdx.x * (DCRX_QBLOCK / 64) >= (uint64_t)t_general + t_rescue) {  // nothing left for this block
    leave();
    return;
  }
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  DevTables T = T0;
  if (TABLE_LDS) {
    const uint32_t lds_addr = dcrx_lds_address(reinterpret_cast<const uint8_t *>(lds_trans));
    stage_lds<DCRX_QBLOCK>(T0.image, lds_trans, T0.lds_image_bytes / 16, T0.dfa_bytes / 16, lds_addr, tid);
    T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_trans), 0);
    T.row0 = lds_addr;
  }
  __syncthreads();
  const Counters C{lds_counts};
  const int lane = tid & 63;
  uint32_t *exc_flag = const_cast<uint32_t *>(B.exc_flag);
  for (;;) {
    uint32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(tile_ticket, 1u);
    ticket = __shfl(ticket, 0);
    if (ticket >= t_general + t_rescue) break;
    if (ticket < t_general) {
      const uint32_t i = ticket * DCRX_GTILE + lane;
      if (lane < DCRX_GTILE && i < n_general) {
        const uint32_t r = gqueue[i];
        decombine_list_one<TABLE_LDS, UNIFORM_LEN>(T, lds_trans, B, cfg, (uint64_t)r, C, records,
                                                   lds_slots + tid * DCRX_LSLOT, true);
        if (S.dev) sink_late(S, records, r);      // tuple sink (behind the v2 kernels): the read's tuple, when it decombined here
        // the read's exception flag (three-launch form: set by the prologue) has served its purpose: cleared here, so that
        // the bitmap is all zero again when the launch ends (no memset per launch)
        if (B.n_exc && ((exc_flag[r >> 5] >> (r & 31)) & 1u)) atomicAnd(&exc_flag[r >> 5], ~(1u << (r & 31)));
      }
    } else {
      for (int c = 0; c < DCRX_CHUNK; c++) {
        uint32_t hints;
        const uint32_t r = rescue_entry(queue, n_rescue, ticket - t_general, 64 * DCRX_CHUNK, (uint32_t)c * 64 + lane, tagged,
                                        &hints);
        if (r != 0xFFFFFFFFu) {
          decombine_list_one<TABLE_LDS, UNIFORM_LEN>(T, lds_trans, B, cfg, (uint64_t)r, C, records,
                                                     lds_slots + tid * DCRX_LSLOT, false);
          if (S.dev) sink_late(S, records, r);
        }
      }
    }
  }
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
  if (out) __threadfence();      // (the first wave's adds have landed before its first lane signs the block off)
  leave();
}

// Rescue kernel (pair-table form of the list kernel's second half): the clean reads the fast
// kernel deferred, 64 per ticket.  LDS holds what the fast kernel holds — the two-bases-per-step
// table and the side tables — plus two hit lists (2 * HH_K dwords) per lane.  Last kernel of a
// launch: re-arms the work counters.
constexpr int DCRX_RBLOCK = 1024;
constexpr int DCRX_RSLOT = (2 * HH_K) | 1;     // two hit lists per lane; odd: conflict-free
constexpr uint32_t DCRX_RESCUE_LDS_EXTRA = ((DCRX_RBLOCK * DCRX_RSLOT * 4 + 63) / 64) * 64;
static_assert(DCRX_GTILE * DCRX_GENERAL_SLOT <= 64 * DCRX_RSLOT, "a wave's rescue slots must hold its general-list slots");
static_assert(HH_STRIDE + DCRX_GSLOT_EXTRA <= DCRX_GENERAL_WORDS_AT, "hit lists and exception copy come before the words");
static_assert((DCRX_N_COUNTERS * 4 + DCRX_RESCUE_LDS_EXTRA) % 64 == 0, "pair-scan rows must be 64-byte aligned");

template <bool UNIFORM_LEN, int NW>
__global__ __launch_bounds__(DCRX_RBLOCK) void decombine_rescue_kernel(DevTables T0, BatchDev B, CfgDev cfg,
                                                                       dcrx_record_t *__restrict__ records,
                                                                       unsigned long long *__restrict__ counters,
                                                                       const uint32_t *__restrict__ queue,
                                                                       const uint32_t *__restrict__ gqueue,
                                                                       uint32_t *__restrict__ queue_count, uint32_t qcap) {
  extern __shared__ __align__(64) uint32_t smem[];
  uint32_t *lds_counts = smem;
  uint32_t *lds_slots = smem + DCRX_N_COUNTERS;                    // [DCRX_RBLOCK][DCRX_RSLOT]
  uint32_t *lds_trans = smem + DCRX_N_COUNTERS + DCRX_RESCUE_LDS_EXTRA / 4;
  const int tid = threadIdx.x;
  const uint32_t n_rescue = queue_count[0], n_general = queue_count[1];
  const bool tagged = B.n_reads < (1ull << 30);
  const uint32_t tickets = (n_rescue + 63) / 64;
  const uint32_t t_general = (n_general + DCRX_GTILE - 1) / DCRX_GTILE;
  auto leave = [&]() {
    if (tid == 0 && atomicAdd(queue_count + 4, 1u) == gridDim.x - 1) {
      queue_count[0] = queue_count[1] = queue_count[2] = queue_count[3] = 0;
      __threadfence();
      queue_count[4] = 0;
    }
  };
  if ((uint64_t)blockIdx.x * (DCRX_RBLOCK / 64) >= (uint64_t)tickets + t_general) { leave(); return; }
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  const uint32_t lds_addr = dcrx_lds_address(reinterpret_cast<const uint8_t *>(lds_trans));
  uint32_t *lds_side = lds_trans + T0.dfa16_bytes / 4;
  stage_lds<DCRX_RBLOCK>(reinterpret_cast<const uint8_t *>(T0.trans16), lds_trans, T0.dfa16_bytes / 16, T0.dfa16_bytes / 16,
                         lds_addr, tid);
  stage_lds<DCRX_RBLOCK>(T0.image + T0.dfa_bytes, lds_side, (T0.lds_image_bytes - T0.dfa_bytes) / 16, 0, 0, tid);
  DevTables T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_side), T0.dfa_bytes);
  T.row16_0 = lds_addr;
  __syncthreads();
  const Counters C{lds_counts};
  const uint32_t nw = B.stride >> 2;
  const int lane = tid & 63;
  const uint32_t wave = blockIdx.x * (DCRX_RBLOCK / 64) + (uint32_t)(tid >> 6), n_waves = gridDim.x * (DCRX_RBLOCK / 64);
  // The general list first (reads with exception bytes), DCRX_GTILE reads per wave (few lanes: a
  // wave runs the union of its lanes' paths); these longer chains run beside the rescue tiles of
  // the other waves.
  if (t_general) {
    uint32_t *exc_flag = const_cast<uint32_t *>(B.exc_flag);
    uint32_t *gslot = lds_slots + (tid >> 6) * (64 * DCRX_RSLOT);      // the wave's slots: room for DCRX_GTILE list-kernel slots
    for (uint32_t g = wave; g < t_general; g += n_waves) {
      const uint32_t i = g * DCRX_GTILE + lane;
      if (lane < DCRX_GTILE && i < n_general) {
        const uint32_t r = gqueue[i];
        decombine_general16_one<true, UNIFORM_LEN, NW>(T, B, cfg, (uint64_t)r, gqueue[qcap + i], nw, C, records,
                                                       gslot + lane * DCRX_GENERAL_SLOT);
        if ((exc_flag[r >> 5] >> (r & 31)) & 1u) atomicAnd(&exc_flag[r >> 5], ~(1u << (r & 31)));
      }
    }
  }
  // Rescue tiles: whole rounds by static striding (a ticket per tile would serialise: one atomic
  // address sustains ~90 tickets/us and there are ~13 000 tiles in 10 M reads), the last, partial
  // round through a ticket, so that waves that come late (a general tile, slower reads) take fewer.
  auto rescue_tile = [&](uint32_t tile) {
    uint32_t hints;
    const uint32_t r = rescue_entry(queue, n_rescue, tile, 64, (uint32_t)lane, tagged, &hints);
    if (r != 0xFFFFFFFFu)
      decombine_rescue16_one<true, UNIFORM_LEN, NW>(T, B, cfg, (uint64_t)r, hints, nw, C, records, lds_slots + tid * DCRX_RSLOT);
  };
  // the waves that took a general tile stay out of the static rounds
  const uint32_t g_waves = t_general < n_waves / 2 ? t_general : 0u, s_waves = n_waves - g_waves;
  const uint32_t static_tiles = (tickets / s_waves) * s_waves;
  if (wave >= g_waves)
    for (uint32_t tile = wave - g_waves; tile < static_tiles; tile += s_waves) rescue_tile(tile);
  uint32_t *tile_ticket = queue_count + 3;
  for (;;) {
    uint32_t tile = 0;
    if (lane == 0) tile = atomicAdd(tile_ticket, 1u);
    tile = static_tiles + __shfl(tile, 0);
    if (tile >= tickets) break;
    rescue_tile(tile);
  }
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
  leave();
}

// Prologue of a launch, one thread per exception entry (or per read when `all`): zeroes the
// caller's counters, sets the exception bit of every read on the exception list and builds the
// general work list — every read when `all` (orientation `both`, forced slow reader), else the
// reads that have exception entries (first entry of each read appends it).
__global__ void prologue_kernel(const uint32_t *__restrict__ exc_read, uint64_t n_exc, int all, int list, uint64_t n_reads,
                                uint32_t *__restrict__ flag, uint32_t *__restrict__ gqueue, uint32_t *__restrict__ gstart,
                                uint32_t *__restrict__ gcount, unsigned long long *__restrict__ counters) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < DCRX_N_COUNTERS) counters[i] = 0;
  if (i < n_exc) {
    const uint32_t r = exc_read[i];
    atomicOr(&flag[r >> 5], 1u << (r & 31));
    if (!all && list && (i == 0 || exc_read[i - 1] != r)) {    // list == 0: the v2 kernels resolve these reads themselves
      const uint32_t slot = atomicAdd(gcount, 1u);
      gqueue[slot] = r;
      gstart[slot] = (uint32_t)i;        // where the read's entries start in the exception list
    }
  }
  if (all) {
    if (i < n_reads) gqueue[i] = (uint32_t)i;
    if (i == 0) *gcount = (uint32_t)n_reads;
  }
}

// ------------------------------------------------------------------------------
// Order-preserving compaction of the decombined (status OK) records: the DCR
// tuples that leave the GPU (gathered to rank 0 in the sharded run).
// ------------------------------------------------------------------------------
constexpr int CP_BLOCK = 256;
constexpr int CP_PER_THREAD = 4;
constexpr int CP_TILE = CP_BLOCK * CP_PER_THREAD;

__global__ __launch_bounds__(CP_BLOCK) void compact_count_kernel(const dcrx_record_t *__restrict__ rec, uint64_t n,
                                                                  uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE;
  uint32_t c = 0;
  for (int k = 0; k < CP_PER_THREAD; k++) {
    const uint64_t i = base + (uint64_t)k * CP_BLOCK + threadIdx.x;
    const bool ok = i < n && rec[i].status == DCRX_S_OK;
    c += (uint32_t)__popcll(__ballot(ok));
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = s_cnt;
}

// Exclusive scan of the tile counts by one block; writes the grand total.
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t *__restrict__ tile_count, uint32_t n_tiles,
                                                             uint64_t *__restrict__ tile_off,
                                                             uint64_t *__restrict__ total) {
  __shared__ uint64_t s_part[1024];
  const uint32_t per = (n_tiles + 1023) / 1024;
  const uint32_t b = threadIdx.x * per;
  uint64_t sum = 0;
  for (uint32_t i = b; i < b + per && i < n_tiles; i++) sum += tile_count[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < 1024; i++) { uint64_t t = s_part[i]; s_part[i] = run; run += t; }
    *total = run;
  }
  __syncthreads();
  uint64_t run = s_part[threadIdx.x];
  for (uint32_t i = b; i < b + per && i < n_tiles; i++) { tile_off[i] = run; run += tile_count[i]; }
}

__global__ __launch_bounds__(CP_BLOCK) void compact_scatter_kernel(const dcrx_record_t *__restrict__ rec, uint64_t n,
                                                                    uint64_t first_index,
                                                                    const uint64_t *__restrict__ tile_off,
                                                                    dcrx_record_t *__restrict__ hits,
                                                                    uint64_t *__restrict__ hit_index,
                                                                    uint64_t *__restrict__ ok_bitmap, int packed12) {
  __shared__ uint32_t s_wave[CP_PER_THREAD][CP_BLOCK / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool ok[CP_PER_THREAD];
  uint32_t rank[CP_PER_THREAD];
  for (int k = 0; k < CP_PER_THREAD; k++) {
    const uint64_t i = base + (uint64_t)k * CP_BLOCK + threadIdx.x;
    ok[k] = i < n && rec[i].status == DCRX_S_OK;
    const unsigned long long m = __ballot(ok[k]);
    rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) {
      s_wave[k][wave] = (uint32_t)__popcll(m);
      // bit (i & 63) of word i >> 6: read i decombined.  A wave covers 64 consecutive reads from a multiple of 64.
      const uint64_t i0 = base + (uint64_t)k * CP_BLOCK + (uint64_t)wave * 64;
      if (ok_bitmap && i0 < n) ok_bitmap[i0 >> 6] = m;
    }
  }
  __syncthreads();
  uint64_t off = tile_off[blockIdx.x];
  for (int k = 0; k < CP_PER_THREAD; k++) {
    uint32_t before = 0;
    for (int kk = 0; kk < k; kk++)
      for (int wv = 0; wv < CP_BLOCK / 64; wv++) before += s_wave[kk][wv];
    for (int wv = 0; wv < wave; wv++) before += s_wave[k][wv];
    if (ok[k]) {
      const uint64_t i = base + (uint64_t)k * CP_BLOCK + threadIdx.x;
      const uint64_t dst = off + before + rank[k];
      if (packed12 == 1) {      // the 12-byte tuple of include/dcrx.h (dcrx_compact_hits_packed_device)
        const dcrx_record_t r = rec[i];
        uint32_t *o = reinterpret_cast<uint32_t *>(hits) + dst * 3;
        o[0] = (uint32_t)r.v | ((uint32_t)r.j << 12) | ((uint32_t)r.vdel << 24);
        o[1] = (uint32_t)r.v_start | ((uint32_t)r.j_end << 9) | ((uint32_t)r.ins_start << 18);
        o[2] = (uint32_t)r.ins_len | ((uint32_t)r.jdel << 9) | ((uint32_t)r.frame << 17);
      } else if (packed12 == 2) {   // the 8-byte tuple (dcrx_compact_hits_packed8_device): ins_start is left out, the receiver re-derives it
        const dcrx_record_t r = rec[i];
        uint2 o;
        o.x = ((uint32_t)r.v & 0x7FFu) | (((uint32_t)r.j & 0x1FFu) << 11) | ((uint32_t)r.vdel << 20) | (((uint32_t)r.jdel & 0xFu) << 28);
        o.y = ((uint32_t)r.jdel >> 4) | (((uint32_t)r.v_start & 0x1FFu) << 4) | (((uint32_t)r.j_end & 0x1FFu) << 13) | (((uint32_t)r.ins_len & 0x1FFu) << 22) |
              (((uint32_t)r.frame & 1u) << 31);
        reinterpret_cast<uint2 *>(hits)[dst] = o;
      } else {
        reinterpret_cast<uint4 *>(hits)[dst] = reinterpret_cast<const uint4 *>(rec)[i];
      }
      if (hit_index) hit_index[dst] = first_index + i;
    }
  }
}

// The narrow tuple of include/dcrx.h (dcrx_tuple_layout): fields least significant first, widths from the tag tables.
// short_end: the record's j_end is not the J tag's end but its start + 2 * j_half_split (J half1 rescue, decombine.py:450-454).
DCRX_DEV uint64_t narrow_tuple(const dcrx_record_t &r, const TupleLayoutDev &L) {
  const int tagpos = (int)r.ins_start + (int)r.ins_len - (int)r.jdel + L.j_jump[r.j];
  const uint64_t short_end = ((int)r.j_end - tagpos) != (int)L.j_tag_len[r.j] ? 1u : 0u;
  uint64_t t = r.v;
  uint32_t sh = L.w_v;
  t |= (uint64_t)r.j << sh; sh += L.w_j;
  t |= (uint64_t)r.vdel << sh; sh += L.w_vdel;
  t |= (uint64_t)r.jdel << sh; sh += L.w_jdel;
  t |= (uint64_t)r.v_start << sh; sh += L.w_pos;
  t |= (uint64_t)r.j_end << sh; sh += L.w_pos;
  t |= short_end << sh; sh += 1;
  t |= (uint64_t)(r.frame & 1u) << sh;
  return t;
}
// tuple k of a message: low 32 bits in plane A, the other bytes in plane B
DCRX_DEV void narrow_store(uint32_t *plane_a, uint8_t *plane_b, const uint32_t hi_bytes, const uint64_t k, const uint64_t t) {
  plane_a[k] = (uint32_t)t;
  uint32_t hi = (uint32_t)(t >> 32);
  for (uint32_t b = 0; b < hi_bytes; b++) { plane_b[k * hi_bytes + b] = (uint8_t)hi; hi >>= 8; }
}

__global__ __launch_bounds__(CP_BLOCK) void compact_scatter_narrow_kernel(const dcrx_record_t *__restrict__ rec, uint64_t n,
                                                                           const uint64_t *__restrict__ tile_off,
                                                                           const uint64_t *__restrict__ total,
                                                                           uint8_t *__restrict__ msg, uint64_t n_slots, TupleLayoutDev L) {
  __shared__ uint32_t s_wave[CP_PER_THREAD][CP_BLOCK / 64];
  const uint64_t base = (uint64_t)blockIdx.x * CP_TILE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t *ok_bitmap = reinterpret_cast<uint64_t *>(msg);
  const uint64_t bm_bytes = ((n_slots + 63) / 64) * 8;      // the bitmap spans n_slots >= n reads: no bit beyond read n
  if (blockIdx.x == 0)
    for (uint64_t wd = (n + 63) / 64 + threadIdx.x; wd < bm_bytes / 8; wd += CP_BLOCK) ok_bitmap[wd] = 0;
  uint32_t *plane_a = reinterpret_cast<uint32_t *>(msg + bm_bytes);
  uint8_t *plane_b = msg + bm_bytes + *total * 4;
  const uint32_t hi_bytes = L.bytes - 4;
  bool ok[CP_PER_THREAD];
  uint32_t rank[CP_PER_THREAD];
  for (int k = 0; k < CP_PER_THREAD; k++) {
    const uint64_t i = base + (uint64_t)k * CP_BLOCK + threadIdx.x;
    ok[k] = i < n && rec[i].status == DCRX_S_OK;
    const unsigned long long m = __ballot(ok[k]);
    rank[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) {
      s_wave[k][wave] = (uint32_t)__popcll(m);
      const uint64_t i0 = base + (uint64_t)k * CP_BLOCK + (uint64_t)wave * 64;
      if (i0 < n) ok_bitmap[i0 >> 6] = m;
    }
  }
  __syncthreads();
  const uint64_t off = tile_off[blockIdx.x];
  for (int k = 0; k < CP_PER_THREAD; k++) {
    uint32_t before = 0;
    for (int kk = 0; kk < k; kk++)
      for (int wv = 0; wv < CP_BLOCK / 64; wv++) before += s_wave[kk][wv];
    for (int wv = 0; wv < wave; wv++) before += s_wave[k][wv];
    if (ok[k]) {
      const uint64_t i = base + (uint64_t)k * CP_BLOCK + threadIdx.x;
      narrow_store(plane_a, plane_b, hi_bytes, off + before + rank[k], narrow_tuple(rec[i], L));
    }
  }
}

// ------------------------------------------------------------------------------
// launchers (called from dcrx_api.cpp)
// ------------------------------------------------------------------------------
// per-device "attribute already set" marks (one process may drive several devices)
// (the mark is set by attributes_set_on_device once every hipFuncSetAttribute of the launcher has succeeded: a failure
// in between must not leave later launches without the LDS attribute)
bool first_use_on_device(bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  return !seen[dev];
}
void attributes_set_on_device(bool (&seen)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) seen[dev] = true;
}

template <bool TABLE_LDS, bool UNIFORM, int NW, int ARITY>
static hipError_t launch_all(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg,
                             dcrx_record_t *rec, uint32_t *queue, uint32_t *gqueue, uint32_t *queue_count,
                             unsigned long long *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  auto kfast = decombine_kernel<TABLE_LDS, UNIFORM, NW, ARITY>;
  constexpr uint32_t FBLOCK = ARITY == 16 ? DCRX_BLOCK16 : DCRX_BLOCK;
  auto klist = decombine_list_kernel<TABLE_LDS, UNIFORM>;
  const uint32_t lds_fast = ARITY == 16 ? P.lds16_bytes + DCRX_FAST16_LDS_EXTRA : P.lds_bytes + DCRX_FAST_LDS_EXTRA;
  const uint32_t lds_list = P.lds_bytes + DCRX_QUEUE_LDS_EXTRA;
  hipError_t e;
  // persistent grids: as many blocks as are resident at once for THIS table size (queried per
  // launch: a host-side call, and tables of different sizes share the kernel instantiations)
  static bool attr_seen[64];
  if (first_use_on_device(attr_seen)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfast), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(klist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attributes_set_on_device(attr_seen);
  }
  int occ_fast = 0, occ_list = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_fast, kfast, FBLOCK, lds_fast);
  if (e != hipSuccess) return e;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_list, klist, DCRX_QBLOCK, lds_list);
  if (e != hipSuccess) return e;
  occ_fast = std::max(occ_fast, 1); occ_list = std::max(occ_list, 1);
  // (the three-launch form holds 320 nt in registers: longer reads all go through its list kernel; the v2 kernels hold 511)
  // (the v2 condition: its entries keep two flags above a 30-bit read index)
  const bool v2_ok_here = B.stride <= 4 * DCRX_V2_NWLONG && v2_applies(P, T, cfg, B.stride) && B.n_reads < (1ull << 30);
  // (`both` as two v2 passes: 0.69 ms per 10 M reads of config 2 against 28 ms through the list kernel)
  const bool v2_both = cfg.orientation == DCRX_ORIENT_BOTH && v2_ok_here;
  const bool all_general = (cfg.orientation == DCRX_ORIENT_BOTH && !v2_both) || (cfg.flags & DCRX_F_FORCE_SLOW_READER) || (B.stride > 4 * DCRX_NWMAX && !v2_ok_here);
  // reserved_cus: compute units left to other streams (an RCCL gather running beside the scan)
  const uint32_t cus = P.n_cu > P.reserved_cus ? P.n_cu - P.reserved_cus : 1u;
  const bool v2 = v2_ok_here && cfg.orientation != DCRX_ORIENT_BOTH;
  const uint32_t grid = all_general ? 0 : std::min<uint32_t>(P.grid, cus * (uint32_t)occ_fast);
  const uint32_t qgrid = std::min<uint32_t>(P.qgrid, cus * (uint32_t)occ_list);
  const uint32_t qcap = (uint32_t)(gqueue - queue);      // capacity of the rescue queue, of the general
                                                         // list behind it, and of the list of exception-list offsets behind that
  // The three-launch form: the prologue zeroes the caller's counters (and marks / lists the reads with exception bytes), the
  // kernels add their tallies with one atomic per counter and block, and the list kernel leaves the work counters and the
  // exception bitmap zeroed for the next batch.  The v2 kernels need no prologue: the scan blocks mark the exception reads of
  // their own ranges, every kernel tallies into the handle's accumulator (zero between calls), and the list kernel — the
  // last launch — hands the counters to the caller and re-arms the accumulator.
  // (which form this call's launches take, for whoever asks: dcrx_tune_state — a fallback to the three-launch form is then seen,
  // not inferred from the clock)
  if (P.tune) P.tune[cfg.orientation == DCRX_ORIENT_FORWARD ? 0 : 1].last_form = (v2 || v2_both) ? 2u : 1u;
  unsigned long long *acc = (v2 || v2_both) ? reinterpret_cast<unsigned long long *>(P.v2_acc) : d_counters;
  if (!(v2 || v2_both)) {
    const uint64_t items = std::max<uint64_t>(all_general ? B.n_reads : 0, std::max<uint64_t>(B.n_exc, 1));
    // (the call's start event, when one is set, rides on this dispatch)
    hipExtLaunchKernelGGL(prologue_kernel, dim3((uint32_t)((items + 255) / 256)), dim3(256), 0, s, P.ev_step_start, nullptr, 0, B.exc_read, B.n_exc,
                          all_general ? 1 : 0, 1, B.n_reads, const_cast<uint32_t *>(B.exc_flag), gqueue, gqueue + qcap,
                          queue_count + 1, d_counters);
  }
  // (v2: the call's start event rides on the scan's dispatch — the first launch — unless the scan's own start event claims that
  // place, or nothing is launched for an empty batch: then it is recorded in front)
  hipEvent_t scan_start = ev_start;
  if ((v2 || v2_both) && P.ev_step_start) {
    if (!ev_start && B.n_reads) scan_start = P.ev_step_start;
    else { e = hipEventRecord(P.ev_step_start, s); if (e != hipSuccess) return e; }
  }
  if (v2_both) {
    // `both` (decombine.py:1005-1010): the reverse frame for every read, then the forward frame for the reads it did not
    // decombine (retry: the scan skips the reads whose record is OK); the failure counters of both attempts add up, as the
    // reference's do.  After each frame the list kernel takes what the v2 kernels handed over (in that frame); the second one
    // hands the counters over, with read_count set once.
    const uint32_t lgrid = std::max<uint32_t>(1u, std::min<uint32_t>(qgrid, cus / 4));
    CfgDev c1 = cfg, c2 = cfg;
    c1.orientation = DCRX_ORIENT_REVERSE; c2.orientation = DCRX_ORIENT_FORWARD;
    e = launch_v2_any(P, T, B, c1, rec, queue, gqueue, qcap, queue_count, acc, s, scan_start, ev_stop, 0u);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(klist, dim3(lgrid), dim3(DCRX_QBLOCK), lds_list, s, T, B, c1, rec, acc, queue, gqueue, queue_count, (unsigned long long *)nullptr, ~0ull, V2SinkCall{});
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_v2_any(P, T, B, c2, rec, queue, gqueue, qcap, queue_count, acc, s, nullptr, nullptr, 1u);
    if (e != hipSuccess) return e;
    hipExtLaunchKernelGGL(klist, dim3(lgrid), dim3(DCRX_QBLOCK), lds_list, s, nullptr, P.ev_step_stop, 0, T, B, c2, rec, acc, queue, gqueue, queue_count, d_counters,
                          (unsigned long long)B.n_reads, V2SinkCall{});
    return hipGetLastError();
  }
  // (a tuple sink, when the call has one and the launch shape serves it: the kernels leave the tuples' items, the place
  // kernel behind the list kernel puts the message together; else the caller compacts the records)
  V2SinkLaunch K;
  if (v2) {
    e = launch_v2_any(P, T, B, cfg, rec, queue, gqueue, qcap, queue_count, acc, s, scan_start, ev_stop, 0u, P.sink.dev ? &K : nullptr);
    if (e != hipSuccess) return e;
  }
  if (v2 && dcrx_debug_env("DCRX_DEBUG_HANDOVER")) {      // developer aid: how many reads the v2 kernels handed over
    uint32_t qc[2] = {0, 0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(qc, queue_count, sizeof qc, hipMemcpyDeviceToHost);
    fprintf(stderr, "dcrx: v2 handed over %u clean reads and %u reads with exception bytes\n", qc[0], qc[1]);
  }
  if (!v2 && ev_start) { e = hipEventRecord(ev_start, s); if (e != hipSuccess) return e; }
  if (!v2 && grid) {
    hipLaunchKernelGGL(kfast, dim3(grid), dim3(FBLOCK), lds_fast, s, T, B, cfg, rec, d_counters, queue, queue_count);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (!v2 && ev_stop) { e = hipEventRecord(ev_stop, s); if (e != hipSuccess) return e; }
  // What the fast kernel left: the general list (reads with exception bytes; every read for
  // `both` / the forced slow reader) goes through the list kernel; the rescue queue through the
  // rescue kernel when the pair table serves it, else through the list kernel as well.
  // (behind the v2 kernels only the few reads they hand over are left: the list kernel, whose blocks
  // leave at once when there is nothing to do)
  const bool rescue16 = !v2 && ARITY == 16 && T.pair_rescue && !(cfg.flags & DCRX_F_LIST_RESCUE) &&
                        P.lds16_bytes + DCRX_RESCUE_LDS_EXTRA <= 160u * 1024u;
  if (!rescue16 || all_general) {
    // (behind the v2 kernels a quarter of the grid: every block signs off with an atomic on one address, and that,
    // not the handful of reads, is what the launch costs there)
    const uint32_t lgrid = v2 ? std::max<uint32_t>(1u, std::min<uint32_t>(qgrid, cus / 4)) : qgrid;
    const bool list_is_last = !(rescue16 && !all_general);
    hipExtLaunchKernelGGL(klist, dim3(lgrid), dim3(DCRX_QBLOCK), lds_list, s, nullptr, list_is_last && !K.S.dev ? P.ev_step_stop : nullptr, 0, T, B, cfg, rec,
                          acc, queue, gqueue, queue_count, v2 ? d_counters : (unsigned long long *)nullptr, ~0ull, K.S);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (K.S.dev) {
      e = launch_v2_place(P, K, B.n_reads, s, P.ev_step_stop);
      if (e != hipSuccess) return e;
      if (P.sink.done) *P.sink.done = true;
    }
  }
  if (rescue16 && !all_general) {
    auto kresc = decombine_rescue_kernel<UNIFORM, NW>;
    static bool rattr_seen[64];
    if (first_use_on_device(rattr_seen)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kresc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attributes_set_on_device(rattr_seen);
    }
    const uint32_t lds_resc = P.lds16_bytes + DCRX_RESCUE_LDS_EXTRA;
    hipExtLaunchKernelGGL(kresc, dim3(cus), dim3(DCRX_RBLOCK), lds_resc, s, nullptr, P.ev_step_stop, 0, T, B, cfg, rec, d_counters, queue, gqueue,
                          queue_count, qcap);
    e = hipGetLastError();
  }
  return e;
}

// Reads of 512 .. 65 535 nt (strides beyond 128 bytes): every read through decombine_long_one, one read per lane, tables in
// global memory (the L2 keeps them); the first launch zeroes the caller's counters (on the stream: a kernel of one block).
__global__ void zero_counters_kernel(unsigned long long *__restrict__ counters) {
  if (threadIdx.x < DCRX_N_COUNTERS) counters[threadIdx.x] = 0;
}
// TABLE_LDS: the tables' image (the one-base rows, then the side tables) staged in the block's LDS as the list kernel stages it.
// A step of the scan is a dependent look-up with 64 different addresses per wave: out of the L2 that is 64 requests through
// one texture-address unit (the launch ran at 0.18 T bases/s), out of LDS a gather over 32 banks.
template <bool UNIFORM_LEN, bool TABLE_LDS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void decombine_long_kernel(DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records,
                                                               unsigned long long *__restrict__ counters, const uint32_t slot_dwords) {
  extern __shared__ __align__(16) uint32_t smem[];
  uint32_t *lds_counts = smem;
  uint32_t *lds_trans = smem + DCRX_N_COUNTERS;
  static_assert(DCRX_N_COUNTERS % 4 == 0, "the rows stay 16-byte aligned");
  const int tid = threadIdx.x;
  // a lane's slot (hit lists, the notes of the scan's first pass) behind the tables' image
  uint32_t *slot = lds_trans + (TABLE_LDS ? T0.lds_image_bytes / 4 : 0u) + (size_t)tid * slot_dwords;
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  DevTables T = T0;
  if (TABLE_LDS) {
    const uint32_t lds_addr = dcrx_lds_address(reinterpret_cast<const uint8_t *>(lds_trans));
    stage_lds<BLOCK>(T0.image, lds_trans, T0.lds_image_bytes / 16, T0.dfa_bytes / 16, lds_addr, tid);
    T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_trans), 0);
    T.row0 = lds_addr;
  }
  __syncthreads();
  const Counters C{lds_counts};
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + tid; r < B.n_reads; r += (uint64_t)gridDim.x * blockDim.x)
    decombine_long_one<UNIFORM_LEN, TABLE_LDS>(T, lds_trans, B, cfg, r, C, records, slot, (int)slot_dwords);
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
}
template <bool TABLE_LDS, int BLOCK>
static hipError_t launch_long_as(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                                 unsigned long long *d_counters, hipStream_t s, const uint32_t grid, const uint32_t lds, const uint32_t slot_dwords) {
  auto ku = decombine_long_kernel<true, TABLE_LDS, BLOCK>;
  auto kr = decombine_long_kernel<false, TABLE_LDS, BLOCK>;
  static bool attr_seen[64];
  if (TABLE_LDS && first_use_on_device(attr_seen)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ku), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attributes_set_on_device(attr_seen);
  }
  if (B.lens) hipExtLaunchKernelGGL(kr, dim3(grid), dim3(BLOCK), lds, s, nullptr, P.ev_step_stop, 0, T, B, cfg, rec, d_counters, slot_dwords);
  else hipExtLaunchKernelGGL(ku, dim3(grid), dim3(BLOCK), lds, s, nullptr, P.ev_step_stop, 0, T, B, cfg, rec, d_counters, slot_dwords);
  return hipGetLastError();
}
static hipError_t launch_long(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                              unsigned long long *d_counters, hipStream_t s) {
  hipExtLaunchKernelGGL(zero_counters_kernel, dim3(1), dim3(64), 0, s, P.ev_step_start, nullptr, 0, d_counters);
  const uint32_t cus = P.n_cu > P.reserved_cus ? P.n_cu - P.reserved_cus : 1u;
  // the tables in LDS where a block's image and its lanes' slots leave room for a block of 1 024 threads per compute unit (long dependent
  // chains: the waves of a unit hide one another's look-ups; 4 waves per SIMD is what the kernel's registers allow), else one of 512;
  // a batch of a few reads keeps the form without staging
  constexpr uint32_t LDS_CU = 160u * 1024u;
  auto lds_of = [&](const uint32_t block, const bool tables, const uint32_t slot) { return DCRX_N_COUNTERS * 4u + (tables ? T.lds_image_bytes : 0u) + block * slot * 4u; };
  // the largest slot (an odd number of dwords, DCRX_LONG_SLOT_MIN .. _MAX: the more notes a lane holds, the fewer times its wave takes them back mid-scan) that `per_cu` blocks leave room for; 0: none
  auto slot_for = [&](const uint32_t block, const uint32_t per_cu) {
    for (uint32_t sl = DCRX_LONG_SLOT_MAX; sl >= (uint32_t)DCRX_LONG_SLOT_MIN; sl -= 2u) if (per_cu * lds_of(block, true, sl) <= LDS_CU) return sl;
    return 0u;
  };
  static const bool no_lds = dcrx_debug_env("DCRX_DEBUG_LONG_GLOBAL_TABLES") != nullptr;      // (A/B)
  if (!no_lds && T.lds_image_bytes && slot_for(512, 1) && B.n_reads >= 4096u) {
    // (experiment, DCRX_DEBUG_LONG_BLOCK=768: three waves per SIMD with 170 registers each instead of four with 128)
    static const int long_block = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_LONG_BLOCK"); return e ? atoi(e) : 0; }();
    if (long_block == 768) {
      if (const uint32_t sl = slot_for(768, 1)) {
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus, (B.n_reads + 767) / 768));
        return launch_long_as<true, 768>(P, T, B, cfg, rec, d_counters, s, grid, lds_of(768, true, sl), sl);
      }
    }
    if (const uint32_t sl = slot_for(1024, 1)) {
      const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus, (B.n_reads + 1023) / 1024));
      return launch_long_as<true, 1024>(P, T, B, cfg, rec, d_counters, s, grid, lds_of(1024, true, sl), sl);
    }
    const uint32_t sl = slot_for(512, 1);
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus, (B.n_reads + 511) / 512));
    return launch_long_as<true, 512>(P, T, B, cfg, rec, d_counters, s, grid, lds_of(512, true, sl), sl);
  }
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)cus * 8, (B.n_reads + 63) / 64));      // (a wave per block where the batch is small: long chains, few reads)
  if (B.n_reads > (uint64_t)grid * 64) return launch_long_as<false, 256>(P, T, B, cfg, rec, d_counters, s, grid, lds_of(256, false, DCRX_LONG_SLOT_MAX), DCRX_LONG_SLOT_MAX);
  return launch_long_as<false, 64>(P, T, B, cfg, rec, d_counters, s, grid, lds_of(64, false, DCRX_LONG_SLOT_MAX), DCRX_LONG_SLOT_MAX);
}

hipError_t launch_decombine(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg,
                            dcrx_record_t *rec, uint32_t *queue, uint32_t *gqueue, uint32_t *queue_count,
                            uint64_t *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
  hipError_t e;
  unsigned long long *ctr = reinterpret_cast<unsigned long long *>(d_counters);
  if (B.stride > DCRX_FAST_MAX_STRIDE) return launch_long(P, T, B, cfg, rec, ctr, s);      // reads of 512 nt and more
  const bool uniform = B.lens == nullptr;
  const bool nw10 = B.stride <= 40;  // 150-nt reads: ten words in registers instead of DCRX_NWMAX
#define DCRX_LAUNCH(TL, UN, NW_, AR_) launch_all<TL, UN, NW_, AR_>(P, T, B, cfg, rec, queue, gqueue, queue_count, ctr, s, ev_start, ev_stop)
  const bool pair_scan = P.table16_in_lds && !(cfg.flags & DCRX_F_ONE_BASE_SCAN);
  if (P.table_in_lds && pair_scan) {
    if (nw10) e = uniform ? DCRX_LAUNCH(true, true, 10, 16) : DCRX_LAUNCH(true, false, 10, 16);
    else e = uniform ? DCRX_LAUNCH(true, true, DCRX_NWMAX, 16) : DCRX_LAUNCH(true, false, DCRX_NWMAX, 16);
  } else if (P.table_in_lds) {
    if (nw10) e = uniform ? DCRX_LAUNCH(true, true, 10, 4) : DCRX_LAUNCH(true, false, 10, 4);
    else e = uniform ? DCRX_LAUNCH(true, true, DCRX_NWMAX, 4) : DCRX_LAUNCH(true, false, DCRX_NWMAX, 4);
  } else {
    if (nw10) e = uniform ? DCRX_LAUNCH(false, true, 10, 4) : DCRX_LAUNCH(false, false, 10, 4);
    else e = uniform ? DCRX_LAUNCH(false, true, DCRX_NWMAX, 4) : DCRX_LAUNCH(false, false, DCRX_NWMAX, 4);
  }
#undef DCRX_LAUNCH
  return e;
}

hipError_t launch_compact(const dcrx_record_t *rec, uint64_t n, uint64_t first_index, dcrx_record_t *hits,
                          uint64_t *hit_index, uint64_t *ok_bitmap, int packed12, uint64_t *d_total, uint32_t *tile_count,
                          uint64_t *tile_off, hipStream_t s) {
  const uint32_t n_tiles = (uint32_t)((n + CP_TILE - 1) / CP_TILE);
  if (n_tiles)
    hipLaunchKernelGGL(compact_count_kernel, dim3(n_tiles), dim3(CP_BLOCK), 0, s, rec, n, tile_count);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, tile_count, n_tiles, tile_off, d_total);
  if (n_tiles)
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(n_tiles), dim3(CP_BLOCK), 0, s, rec, n, first_index, tile_off,
                       hits, hit_index, ok_bitmap, packed12);
  return hipGetLastError();
}

hipError_t launch_compact_narrow(const dcrx_record_t *rec, uint64_t n, uint8_t *msg, uint64_t n_slots, const TupleLayoutDev &L, uint64_t *d_total,
                                 uint32_t *tile_count, uint64_t *tile_off, hipStream_t s) {
  const uint32_t n_tiles = (uint32_t)((n + CP_TILE - 1) / CP_TILE);
  if (n_tiles)
    hipLaunchKernelGGL(compact_count_kernel, dim3(n_tiles), dim3(CP_BLOCK), 0, s, rec, n, tile_count);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, tile_count, n_tiles, tile_off, d_total);
  // (an empty batch still clears the bitmap of its slots: one block)
  if (n_tiles || n_slots)
    hipLaunchKernelGGL(compact_scatter_narrow_kernel, dim3(n_tiles ? n_tiles : 1), dim3(CP_BLOCK), 0, s, rec, n, tile_off, d_total, msg, n_slots, L);
  return hipGetLastError();
}

uint32_t compact_tiles(uint64_t n) { return (uint32_t)((n + CP_TILE - 1) / CP_TILE); }

}  // namespace dcrx
