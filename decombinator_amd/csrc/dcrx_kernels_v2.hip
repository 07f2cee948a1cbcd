// dcrx_kernels_v2.hip — the v2 decombine kernels for gfx950 (CDNA4) and their launcher.
// Per-read code: dcrx_v2_device.h.  The three-launch form (dcrx_kernels.hip) stays for
// orientation `both`, tag sets the v2 tables do not express, and the few reads these kernels
// hand over (more flagged pairs than an entry holds).
//
// Launches per batch: scan2_kernel -> finish2_kernel (the lean rescue, the lean tail and the general
// form over list X as roles of ONE launch on the caller's stream) -> the list kernel (dcrx_kernels.hip: hand-overs, normally
// none; its first block hands the call's counters to the caller).  No prologue launch: the scan blocks mark the reads with
// exception bytes of their own ranges, and the kernels tally into an accumulator the handle owns.
//
//   scan2_kernel    persistent, one 1024-thread block per CU.  LDS holds the frame's 16-bit pair
//                   table (at LDS address 0) and nothing else.  Per tile of 1024 * RPL reads: the
//                   next tile's words are requested, this tile is scanned (RPL independent chains
//                   per lane, one ds_read_u16 + one v_alignbit per two bases), the flag log is
//                   digested, and by ballot a read is
//                     - finished at once (no V tag and no V half tag / several V tags),
//                     - appended to the wave's TAIL list (read, V pair, J pair, words), or
//                     - appended to the wave's EVENT list (read, flag log, words).
//                   Each wave owns a region of both lists (no atomics) and leaves its counts.
//   rescue2_kernel  event entries in straight-line code (rescue2_fast): full-tag pairs as the tail
//                   kernel resolves them, half-tag rescue over up to four flagged pairs per gene.
//   tail2_kernel    tail entries in straight-line code (tail2_fast).
//                   Both lean kernels: 256-thread blocks, side tables and keyword buckets in LDS, the
//                   read in hand in a per-lane LDS strip; what they do not settle goes to the slow list.
//   events2_kernel  the slow list through the general form (dcr_frame3) on a read held in registers;
//                   reads with exception bytes are resolved here against their exception list.
//
// The scan has its registers to itself (no finishing code in that kernel); the lean kernels run at
// 4-5 waves per SIMD; the general form sees a few thousand reads per batch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/dcrx.h"
#include "dcrx_device.h"
#include "dcrx_launch.h"
#include "dcrx_dcr_device.h"
#include "dcrx_v2_device.h"
#include "dcrx_sink_device.h"

namespace dcrx {

bool first_use_on_device(bool (&seen)[64]);
void attributes_set_on_device(bool (&seen)[64]);

constexpr int DCRX_V2_BLOCK = 1024;
#ifdef DCRX_SCAN_STAMPS
__device__ uint32_t *g_scan_stamps;      // instrumented build (tools/): four 100 MHz stamps per wave of the scan kernel
#endif
constexpr int DCRX_V2_FBLOCK = 256;      // (128: the same step; 512: twice as long — profiles/r05/finish_block_size_ab.log)
#ifndef DCRX_V2_TBLOCK
#define DCRX_V2_TBLOCK 256      /* threads of a tail-kernel block ... */
#define DCRX_V2_TWAVES 5        /* ... and the waves per SIMD it is compiled for */
#endif
#ifndef DCRX_V2_RWAVES
#define DCRX_V2_RWAVES 4        /* waves per SIMD the rescue kernel is compiled for */
#endif
#ifndef DCRX_V2_RECORD_STORES
#define DCRX_V2_RECORD_STORES 1   /* 0: streamed record stores everywhere; 1: through the caches in the finishing kernels; 2: in the scan kernel as well */
#endif
#if DCRX_V2_RECORD_STORES >= 1
#define DCRX_STORE_FINISH dcrx_store_record_cached
#else
#define DCRX_STORE_FINISH dcrx_store_record
#endif
#if DCRX_V2_RECORD_STORES >= 2
#define DCRX_STORE_SCAN dcrx_store_record_cached
#else
#define DCRX_STORE_SCAN dcrx_store_record
#endif
#ifndef DCRX_V2_LOOP_PRIO
#define DCRX_V2_LOOP_PRIO 2   /* s_setprio of a scanning wave inside scan2() (0: none; A/B) */
#endif
#ifndef DCRX_V2_FULL_LINE_RECORDS
#define DCRX_V2_FULL_LINE_RECORDS 1   /* the scan kernel writes a record for every read (a placeholder where a later kernel settles it): whole lines leave the wave; A/B: 0.545 against 0.554 ms per step */
#endif
#ifndef DCRX_V2_ROLE_CALLS
#define DCRX_V2_ROLE_CALLS 0   /* 1: the roles of the finishing launch are functions (calls); 0: inlined, each re-reading the launch's arguments behind a barrier the optimiser cannot look through */
#endif
#if DCRX_V2_ROLE_CALLS
#define DCRX_V2_ROLE __device__ __attribute__((noinline))
#else
#define DCRX_V2_ROLE __device__ __forceinline__
#endif
#ifndef DCRX_LEAN_LDS_WORDS
#define DCRX_LEAN_LDS_WORDS 1   /* the lean kernels keep the read in hand in LDS strips (0: in registers, A/B) */
#endif
constexpr uint32_t DCRX_V2_GROUP_MAX = 64;   // regions an event-kernel block can take together

// ---- the lists: per block of the scan kernel one region of each ----
// An entry carries the read's packed words (the scan kernel has them in registers), so that the
// finishing kernels never gather from the read array: their loads are the entries, one lane one slot,
// fully coalesced.  Tail entry: read, digest (tail2_pack), words; event entry: read | flags, the
// read's flag log, words.  Event entries by what they need (rescue2_fast):
//   region `ev`:  E = one gene has its full tag, the other needs the half-tag rescue (V2_SHAPE_ONE): one sweep per wave
//   region `sx`:  C = half-tag rescue of both genes (V2_SHAPE_BOTH)  |  X = reads with exception bytes, flags on an odd read's
//                     last half pair: the general form            (half of the region each)
// (a list that outgrows its room hands the rest to the list kernel; what a lean kernel does not settle it finishes itself,
// behind the launch's last block: v2_left_push, v2_general_role)
enum { V2_L_TAIL = 0, V2_L_E = 1, V2_L_C = 2, V2_L_X = 3, V2_L_RING = 4, V2_L_TICKETS = 5, V2_L_TWHINT = 6, V2_L_COUNTS = 8 };      // (V2_L_TICKETS: the lean rescue's batch tickets of lists E (low half) and C (high half): zero after the scan)      // (V2_L_RING: tail entries that went through the fused scan's ring)
constexpr uint32_t V2_LEFT_CAP = 1024;      // entries of the left list (a handful per 10 M reads; more go to the list kernel)
enum { V2_QC_LEFT = 5, V2_QC_DONE = 6 };    // words of the queue header (DCRX_QUEUE_HEADER): entries of the left list, finishing blocks done
struct V2Lists {
  uint4 *tail;        // [regions][rows_t][tcap]
  uint4 *ev;          // [regions][rows_e][ecap]
  uint4 *sx;          // [regions][rows_e][scap]
  uint4 *left;        // [rows_e][V2_LEFT_CAP]: event entries of what the lean roles did not settle (count: queue_count[V2_QC_LEFT])
  uint32_t *counts;   // [regions][V2_L_COUNTS]: entries of each list (V2_L_*)
  uint32_t tcap, ecap, scap;      // scap: a multiple of 128 (two lists of whole chunks)
};
template <int NW>
struct V2Rows {
  static constexpr int T = (2 + NW + 3) / 4, E = (1 + 2 * NW + 3) / 4;
  static_assert((2 + 2 * NW + 3) / 4 == E, "the digest word of an event entry (put_event) lies in the entry's last row");
};
// an event list of a region: its rows (entry i in slot i) and its capacity
struct V2ListRef { uint4 *rows; uint32_t cap; };
template <int NW>
__device__ __forceinline__ V2ListRef v2_list(const V2Lists &Q, const int which, const size_t region) {
  constexpr size_t E = V2Rows<NW>::E;
  if (which == V2_L_E) return V2ListRef{Q.ev + region * Q.ecap * E, Q.ecap};
  return V2ListRef{Q.sx + (region * Q.scap + (which == V2_L_X ? Q.scap / 2 : 0u)) * E, Q.scap / 2};
}
// dwords x[0 .. N) of slot `at` of a region.  Layout: chunks of 64 slots, inside a chunk structure of arrays
// (row k of slot i at rows[((i / 64) * R + k) * 64 + i % 64], R = rows per entry): a wave's batch of 64
// entries is one contiguous piece of R KB, every load of it fully coalesced, and a region is one stream of
// addresses, not R of them (the lists are hundreds of MB: fewer streams, fewer pages in flight).
template <int N>
__device__ __forceinline__ void v2_put_rows(uint4 *rows, const uint32_t cap, const uint32_t at, const uint32_t (&x)[N]) {
  constexpr uint32_t R = (N + 3) / 4;
  uint4 *p = rows + ((size_t)(at >> 6) * R) * 64 + (at & 63u);
#pragma unroll
  for (int k = 0; k < (N + 3) / 4; k++)
    p[(size_t)k * 64] = make_uint4(x[4 * k], 4 * k + 1 < N ? x[4 * k + 1] : 0u, 4 * k + 2 < N ? x[4 * k + 2] : 0u,
                                   4 * k + 3 < N ? x[4 * k + 3] : 0u);
}
template <int N>
__device__ __forceinline__ void v2_get_rows(const uint4 *rows, const uint32_t cap, const uint32_t at, const bool live, uint32_t (&x)[N]) {
  constexpr uint32_t R = (N + 3) / 4;
  const uint4 *p = rows + ((size_t)(at >> 6) * R) * 64 + (at & 63u);
#pragma unroll
  for (int k = 0; k < (N + 3) / 4; k++) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (live) v = p[(size_t)k * 64];
    x[4 * k] = v.x;
    if (4 * k + 1 < N) x[4 * k + 1] = v.y;
    if (4 * k + 2 < N) x[4 * k + 2] = v.z;
    if (4 * k + 3 < N) x[4 * k + 3] = v.w;
  }
}

// The batch in hand has landed, and the compiler knows: an empty statement takes and gives back each of its registers, and the
// index the next batch's loads are addressed with goes through the last of them, so that no load of the next batch can be
// scheduled in front.  Without this the compiler renames the copy `x = x1` away, the batch in hand IS the destination of loads
// issued one trip earlier, and its first use gets `s_waitcnt vmcnt(0)` — behind the loads of the NEXT batch, issued a moment
// before (they are conditional: the pass has no count to wait for): the look-ahead then hides nothing, every wave sits out one
// memory round trip per batch with nothing in flight (round 6: found in the ISA of the scan loop and of both lean roles).
template <int N>
__device__ __forceinline__ uint32_t v2_landed(uint32_t (&x)[N], uint32_t next_index) {
#pragma unroll
  for (int k = 0; k < N; k++) asm volatile("" : "+v"(x[k]));
  asm volatile("" : "+v"(next_index) : "v"(x[N - 1]));
  return next_index;
}

// a read the v2 kernels hand to the three-launch form: clean reads to its rescue queue, reads
// with exception bytes to its general list (with the offset of their exception entries)
__device__ __forceinline__ void v2_hand_over(const BatchDev &B, uint32_t *queue, uint32_t *gqueue, uint32_t qcap, uint32_t *queue_count,
                                             const bool tagged, const uint32_t r, const bool exc) {
  if (!exc) {
    const uint32_t at = atomicAdd(queue_count, 1u);
    queue[at] = tagged ? (r | (3u << 30)) : r;
    return;
  }
  uint64_t lo = 0, hi = B.n_exc;
  while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (B.exc_read[mid] < r) lo = mid + 1; else hi = mid; }
  const uint32_t at = atomicAdd(queue_count + 1, 1u);
  gqueue[at] = r;
  gqueue[qcap + at] = (uint32_t)lo;
}

// counters of a wave's lean-tail statuses: one ballot per status, one LDS atomic per counter and wave
__device__ __forceinline__ void v2_tally(uint32_t *lds_counts, const int lane, const int status, const bool forward) {
  const unsigned long long m_ok = __ballot(status == DCRX_S_OK), m_jn = __ballot(status == DCRX_S_J_NONE),
                           m_jm = __ballot(status == DCRX_S_J_MULTI), m_tl = __ballot(status == DCRX_S_F_TOOLONG),
                           m_im = __ballot(status == DCRX_S_F_IMPOSS_DEL), m_ov = __ballot(status == DCRX_S_F_OVERLAP),
                           m_jw = __ballot(status == DCRX_S_J_WALK_FAIL), m_all = __ballot(status >= 0);
  if (lane == 0) {
    const uint32_t n_ok = (uint32_t)__popcll(m_ok), n_jn = (uint32_t)__popcll(m_jn), n_jm = (uint32_t)__popcll(m_jm),
                   n_tl = (uint32_t)__popcll(m_tl), n_im = (uint32_t)__popcll(m_im), n_ov = (uint32_t)__popcll(m_ov);
    const uint32_t all = (uint32_t)__popcll(m_all);      // (a walk that failed has counted its own counter: V_WALK_FAIL adds nothing else)
    if (m_jw) atomicAdd(&lds_counts[DCRX_C_VJ_ASSIGNMENT_FAILED], (uint32_t)__popcll(m_jw));
    if (all) atomicAdd(&lds_counts[DCRX_C_READ_COUNT], all);
    if (n_ok) { atomicAdd(&lds_counts[DCRX_C_VJ_COUNT], n_ok); if (forward) atomicAdd(&lds_counts[DCRX_C_FRAME_FORWARD], n_ok); }
    if (n_jn) atomicAdd(&lds_counts[DCRX_C_NO_J_ASSIGNED], n_jn);
    if (n_jm) atomicAdd(&lds_counts[DCRX_C_MULTIPLE_J_MATCHES], n_jm);
    if (n_jn + n_jm) atomicAdd(&lds_counts[DCRX_C_VJ_ASSIGNMENT_FAILED], n_jn + n_jm);
    if (n_tl) atomicAdd(&lds_counts[DCRX_C_DCRFILTER_TOOLONG_INTERTAG], n_tl);
    if (n_im) atomicAdd(&lds_counts[DCRX_C_DCRFILTER_IMPOSS_DELETION], n_im);
    if (n_ov) atomicAdd(&lds_counts[DCRX_C_DCRFILTER_TAG_OVERLAP], n_ov);
  }
}

// counters of a wave's lean-rescue statuses (rescue2_count, by ballot)
__device__ __forceinline__ void v2_tally_rescue(uint32_t *lds_counts, const int lane, const int status, const uint32_t errs, const bool forward) {
  const bool done = status >= 0;
  const unsigned long long m_all = __ballot(done);
  if (!m_all) return;
  auto cnt = [&](const bool c) { return (uint32_t)__popcll(__ballot(done && c)); };
  const uint32_t n_ok = cnt(status == DCRX_S_OK), n_v1 = cnt(status == DCRX_S_V_HALF1_EXHAUSTED), n_v2 = cnt(status == DCRX_S_V_HALF2_EXHAUSTED),
                 n_vn = cnt(status == DCRX_S_V_NONE), n_jm = cnt(status == DCRX_S_J_MULTI), n_jn = cnt(status == DCRX_S_J_NONE),
                 n_j1 = cnt(status == DCRX_S_J_HALF1_EXHAUSTED), n_j2 = cnt(status == DCRX_S_J_HALF2_EXHAUSTED),
                 n_tl = cnt(status == DCRX_S_F_TOOLONG), n_im = cnt(status == DCRX_S_F_IMPOSS_DEL), n_ov = cnt(status == DCRX_S_F_OVERLAP),
                 n_jw = cnt(status == DCRX_S_J_WALK_FAIL),       // (a full-tag walk that failed has counted its own counter)
                 e_v1 = cnt((errs & 1u) != 0u), e_v2 = cnt((errs & 2u) != 0u), e_j1 = cnt((errs & 4u) != 0u), e_j2 = cnt((errs & 8u) != 0u);
  if (lane == 0) {
    auto add = [&](const int c, const uint32_t k) { if (k) atomicAdd(&lds_counts[c], k); };
    add(DCRX_C_READ_COUNT, (uint32_t)__popcll(m_all));
    add(DCRX_C_VJ_COUNT, n_ok);
    if (forward) add(DCRX_C_FRAME_FORWARD, n_ok);
    add(DCRX_C_FOUNDV1NOTV2, n_v1);
    add(DCRX_C_FOUNDV2NOTV1, n_v2 + n_j2);            // the reference bumps the V key for the exhausted J half-2 list (:526)
    add(DCRX_C_NO_VTAGS_FOUND, n_vn);
    add(DCRX_C_MULTIPLE_J_MATCHES, n_jm);
    add(DCRX_C_NO_J_ASSIGNED, n_jn);
    add(DCRX_C_FOUNDJ1NOTJ2, n_j1);
    add(DCRX_C_VJ_ASSIGNMENT_FAILED, n_jm + n_jn + n_j1 + n_j2 + n_jw);
    add(DCRX_C_DCRFILTER_TOOLONG_INTERTAG, n_tl);
    add(DCRX_C_DCRFILTER_IMPOSS_DELETION, n_im);
    add(DCRX_C_DCRFILTER_TAG_OVERLAP, n_ov);
    add(DCRX_C_VERR1, e_v1); add(DCRX_C_VERR2, e_v2); add(DCRX_C_JERR1, e_j1); add(DCRX_C_JERR2, e_j2);
  }
}

// One wave's item: 64 * RPL consecutive reads from `first`, lane l holding reads first + 64 q + l (q < RPL) while they
// lie below `hi` (the end of the block's range).  With the words comes the wave's slice of the exception bitmap
// (64-read groups start on multiples of 64: two words per group, the same for every lane).
template <int NW, int RPL>
__device__ __forceinline__ void v2_load_item(const BatchDev &B, const uint32_t nw, const uint64_t first, const uint64_t hi, const int lane,
                                             uint32_t (&w)[RPL][NW], unsigned long long (&xm)[RPL]) {
#pragma unroll
  for (int q = 0; q < RPL; q++) {
    const uint64_t r = first + (uint64_t)q * 64 + lane;
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(B.packed + (r < hi ? r : 0) * B.stride);
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
      uint2 t = make_uint2(0u, 0u);
      if ((uint32_t)(2 * k) < nw && r < hi) t = wp2[k];
      w[q][2 * k] = t.x; w[q][2 * k + 1] = t.y;
    }
    xm[q] = 0ull;
    if (B.n_exc) {
      const uint64_t g = (first + (uint64_t)q * 64) >> 5;       // (wave-uniform: the compiler may fetch it through the scalar cache)
      xm[q] = (unsigned long long)B.exc_flag[g] | ((unsigned long long)B.exc_flag[g + 1] << 32);
    }
  }
}


// The reads with exception bytes of a scan block's own range, marked by the block itself (no prologue launch): the block's
// slice of the sorted exception list — [lower_bound(blk_lo), lower_bound(blk_hi)) — is found by all its threads together
// (k-ary search: one probe per thread and round, half the block per bound; two or three rounds), then one thread per entry
// sets (or, when the block has scanned its range, clears) the read's bit in the workspace bitmap.  A block's range is a
// multiple of 512 reads, so its bits fill whole 64-byte lines that no other block reads or writes: neither a vector nor a
// scalar cache can hold a line of them from before the marks.  s: 6 words of LDS.
__device__ __forceinline__ void v2_exc_slice(const BatchDev &B, const uint64_t blk_lo, const uint64_t blk_hi, uint32_t *s, const int tid) {
  // A short list (the usual case: one read in two thousand carries such a byte, 5 000 entries for 10 M reads) in ONE pass: the
  // list is sorted, so a bound is the number of entries below it — every thread counts its share against both keys, one LDS
  // atomic per wave and key, one barrier.  The k-ary search below took three rounds of a dependent load between barriers:
  // 4-5 of the 7.5 us a scan block spent before its first look-up (round 6, profiles/r06/scan_wave_stamps.log).
  if (B.n_exc <= 16u * blockDim.x) {
    if (tid < 4) s[tid] = 0u;
    __syncthreads();
    uint32_t below_lo = 0, below_hi = 0;
    for (uint64_t i = (uint32_t)tid; i < B.n_exc; i += blockDim.x) {
      const uint64_t r = B.exc_read[i];
      below_lo += r < blk_lo ? 1u : 0u;
      below_hi += r < blk_hi ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { below_lo += __shfl_xor(below_lo, o); below_hi += __shfl_xor(below_hi, o); }
    if ((tid & 63) == 0) { if (below_lo) atomicAdd(&s[0], below_lo); if (below_hi) atomicAdd(&s[2], below_hi); }
    __syncthreads();
    return;      // s[0] = the slice's first entry, s[2] = one behind its last (what the caller reads)
  }
  const uint32_t half = blockDim.x >> 1;
  const int which = (uint32_t)tid >= half ? 1 : 0;
  const uint32_t j = (uint32_t)tid - (which ? half : 0u);
  const uint64_t key = which ? blk_hi : blk_lo;             // (read indices are < 2^32: check_batch)
  if (tid < 2) { s[2 * tid] = 0u; s[2 * tid + 1] = (uint32_t)B.n_exc; }
  for (;;) {
    __syncthreads();
    const uint32_t lo = s[2 * which], hi = s[2 * which + 1], span = hi - lo;
    const bool done = s[1] == s[0] && s[3] == s[2];
    __syncthreads();                                         // (everyone has read the bounds before anyone moves them)
    if (done) break;
    if (tid < 2) s[4 + tid] = 0u;
    __syncthreads();
    const uint32_t step = span ? (span + half - 1) / half : 1u;
    const uint64_t idx = (uint64_t)lo + (uint64_t)j * step;
    const bool below = span && idx < hi && (uint64_t)B.exc_read[idx] < key;
    const unsigned long long m = __ballot(below);
    if ((tid & 63) == 0 && m) atomicAdd(&s[4 + which], (uint32_t)__popcll(m));
    __syncthreads();
    if (j == 0 && span) {                                    // the probes below the key are a prefix of the probes
      const uint32_t c = s[4 + which], probes = (span + step - 1) / step;
      s[2 * which] = c ? lo + (c - 1) * step + 1 : lo;
      s[2 * which + 1] = c < probes ? lo + c * step : hi;
    }
  }
}
__device__ __forceinline__ void v2_exc_marks(const BatchDev &B, const uint32_t e_lo, const uint32_t e_hi, const bool set, const int tid) {
  uint32_t *flag = const_cast<uint32_t *>(B.exc_flag);
  for (uint64_t i = (uint64_t)e_lo + (uint32_t)tid; i < e_hi; i += blockDim.x) {
    const uint32_t r = B.exc_read[i];
    // (a read's first entry marks it: an all-N read has 150 entries, and as many atomics on one word serialise)
    if (set) { if (i == e_lo || B.exc_read[i - 1] != r) atomicOr(&flag[r >> 5], 1u << (r & 31)); }
    else flag[r >> 5] = 0u;                                  // (all of a word's bits belong to this block)
  }
}

// What a lean role does not settle (one read in two million) becomes an event entry of the launch's left list; one wave of the
// launch — of the first block of list X's role, which stays for it — takes the entries through the general form as they
// arrive (v2_general_role, mode 1) and leaves when every other block has signed off.  Neither a call inside the lean loops — the
// general form as a call there kept 58 vector and 80 scalar registers of the tail loop spilled around a call site that one
// batch in ten thousand reaches — nor a pass of its own behind them (a launch and one read's latency on every step's path).
// (behind the rows of the list: one word per entry, set when the entry is whole)
template <int NW>
__device__ __forceinline__ uint32_t *v2_left_valid(uint4 *left_rows) { return reinterpret_cast<uint32_t *>(left_rows + (size_t)V2_LEFT_CAP * V2Rows<NW>::E); }
template <int NW>
__device__ __forceinline__ bool v2_left_push(uint4 *left_rows, uint32_t *__restrict__ queue_count, const uint32_t x0, const uint32_t (&lg)[NW], const uint32_t (&w)[NW]) {
  const uint32_t at = atomicAdd(queue_count + V2_QC_LEFT, 1u);
  if (at >= V2_LEFT_CAP) return false;      // (the count runs on; its reader clamps it)
  uint32_t x[1 + 2 * NW];
  x[0] = x0;
#pragma unroll
  for (int k = 0; k < NW; k++) { x[1 + k] = lg[k]; x[1 + NW + k] = w[k]; }
  v2_put_rows<1 + 2 * NW>(left_rows, 0u, at, x);
  __threadfence();                          // the entry is in memory ...
  __hip_atomic_store(v2_left_valid<NW>(left_rows) + at, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... before it is called valid
  return true;
}

// LDS of the scan kernel behind the pair table and the counters: the block's work counters
// next item; entries of each list (V2_WK_LIST + V2_L_*)
enum { V2_WK_NEXT = 0, V2_WK_LIST = 1, V2_WK_EXC = 8, V2_WK_HEAD = 16, V2_WK_CLAIM = 17, V2_WK_SCANNED = 18, V2_WK_SINK = 19, V2_WK_HEAD2 = 20, V2_WK_CLAIM2 = 21,
       V2_WK_FILLED = 32, V2_WK_GEN = 48, V2_WK_FILLED2 = 64, V2_WK_GEN2 = 80, V2_WK_WORDS = 96 };      // (…2: the event ring of FUSE_E, same protocol as the tail ring)      // (V2_WK_EXC: six words of v2_exc_slice; from V2_WK_HEAD on: the tail ring of the fused form; V2_WK_SINK: decombined reads of the tail waves, tuple sink)
// The fused form (FUSE >= 0 = the frame): the block's last waves — four of the sixteen, give or take (below) — do not scan: they take the tail entries
// the scanning waves produce, in batches of 64, out of a ring in LDS, and finish them (tail2_fast) while the scan goes on.  The
// scan is bound by its LDS look-ups, the tail by instruction issue: on one SIMD the two share what neither uses up, where the
// tail as a kernel of its own ran behind the scan and beside the rescue (bound by issue as well).  The 168 MB of tail entries
// of a 10 M-read batch never leave the compute unit.
//   ring slot (15 words, odd: the lanes of a wave spread over the banks): the read's NW words, two zero words (a window may
//   read two words past the read), the read's index, its digest (tail2_pack), one word of padding
//   HEAD     slots drawn so far (a wave draws the slots of its tail lanes with one LDS atomic)
//   FILLED   per ring batch of 64 slots: slots written (a wave adds its share behind its writes: LDS keeps a wave's order)
//   CLAIM    next batch to finish (a tail wave takes batch c when FILLED[c % NB] == 64, or what is left once every scanning
//            wave has signed off in SCANNED)
//   GEN      per ring batch: times it has been finished — a scanning wave writes into occurrence k of a ring batch when GEN == k
//   How many waves take the tail is a block's own choice per launch (the roles differ by wave index only): the share of its
//   region's reads that were tail reads in the handle's PREVIOUS launch (V2_L_TWHINT in the counts, left there by the block
//   itself) picks 2 to 6 — (round 4's figures) config 2 (35 % tail reads) runs 3.7 % faster on 5 than on 4 and 10 % slower on 3, the mouse chains
//   of config 5 (half the reads are the other chain's: 17 %) 2 % faster on 3 and 3 % slower on 5, a library with 70 % rearranged
//   reads (54 % tail reads) 5 % faster on 6 than on 5 (profiles/r04/tail_waves_*.log); a handle's first launch takes
//   DCRX_V2_FUSE_TAILWAVES.
#ifndef DCRX_V2_FUSE_TAILWAVES
#define DCRX_V2_FUSE_TAILWAVES 3
#endif
#ifndef DCRX_V2_PRIO_TAIL
#define DCRX_V2_PRIO_TAIL 1   /* s_setprio of the fused form's tail waves (a scanning wave: DCRX_V2_LOOP_PRIO inside its look-up loop, 0 between two scans) */
#endif
constexpr int V2_FUSE_TAILWAVES = DCRX_V2_FUSE_TAILWAVES;
// (round 5, with the keyword tables' look-ups at one wait each a tail wave finishes a batch sooner and fewer of them keep up —
// profiles/r05/tail_waves_by_share.log, ms per step on 2 / 3 / 4 / 5 / 6 tail waves: 12 % tail reads 0.291 / 0.295 / 0.305 / .. ;
// 23 %: 0.335 / 0.322 / 0.329; 35 % (config 2): 0.423 / 0.362 / 0.356 / 0.363 / 0.374; 47 %: .. 0.426 / 0.389 / 0.391 / 0.400;
// 58 %: .. 0.443 / 0.417 / 0.426; 70 %: .. 0.502 / 0.462 / 0.450.  Waves that scan AND take tail batches between two items
// (one loop for both kinds of work, so that no wave slot idles) were built and measured: the loop with both bodies in it runs
// 8 % slower before the first batch changes hands, and the exchange gives nothing back — profiles/r05/flexible_waves_experiment.log)
// With the tail waves one priority level above a scanning wave's work between two scans (and one below its look-up loop: DCRX_V2_PRIO_TAIL
// = 1) a batch finishes sooner again, three waves serve config 2 (0.343-0.345 against 0.347-0.351 ms on four at priority 0, one box:
// tail_wave_priority_ab.log), and the shares move once more — tail_waves_by_share_tail_priority_1.log, 2 / 3 / 4 / 5 tail waves:
// 12 %: 0.294 / 0.299; 23 %: 0.323 / 0.326 / 0.336; 35 %: 0.375 / 0.351 / 0.360; 47 %: .. 0.382 / 0.388 / 0.395; 58 %: .. 0.440 /
// 0.417 / 0.425; 70 %: .. 0.496 / 0.446 / 0.455.
constexpr uint32_t V2_TW6_FRAC256 = 243, V2_TW5_FRAC256 = 205, V2_TW4_FRAC256 = 136, V2_TW3_FRAC256 = 64;      // 95 % / 80 % / 53 % / 25 % of a region's reads
constexpr int V2_RING_STRIDE = 15;
constexpr uint32_t V2_RING_MAXBATCHES = 16;
// FUSE_E (round 6): the event entries of list E — one gene to rescue: a tenth of the reads, a third of a step's vector work, and a
// finishing launch of 90 us that runs alone on the chip — go through a second ring in LDS and are finished by `rescue_waves` further
// waves of the scan block (rescue2_fast_to, V2_SHAPE_ONE) while the scan goes on, as the tail entries are since round 4.
//   event ring slot (odd stride): the read's NW words, two zero words, read | flags, the flag log's NW words, the digest word, padding
template <int NW> constexpr int v2_ering_stride() { return (2 * NW + 4) | 1; }      // 25 words for NW = 10
constexpr uint32_t V2_ERING_BATCHES = 8;
#ifndef DCRX_V2_HELP_DRAIN
#define DCRX_V2_HELP_DRAIN 0      // 1: scanning waves that have left their loop help to empty the rings — measured level (profiles/r06/scanning_waves_help_to_drain_ab.log): off
#endif
constexpr float V2_FUSE_E_MAX_SHARE = 0.25f;      // list E's share of the reads up to which finishing its entries inside the scan kernel is tried (and timed)

// (SINK: the fused form's tail waves also leave the tuple sink's items — an instantiation of its own, so that the kernel of a
// call without a sink carries none of that code: 0.6 % of the step, measured)
template <bool UNIFORM_LEN, int NW, int RPL, bool NARROW, bool PREFETCH = true, int FUSE = -1, bool SINK = false, bool FUSE_E = false>
__global__ __launch_bounds__(DCRX_V2_BLOCK) void scan2_kernel(
    DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records, unsigned long long *__restrict__ counters,
    V2Lists Q, uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue, uint32_t qcap, uint32_t *__restrict__ queue_count,
    uint64_t per_block, uint32_t retry, const DevTables *__restrict__ Tmem, uint32_t ring_batches, V2SinkCall S, uint32_t tail_waves, uint32_t rescue_waves) {
  static_assert(!FUSE_E || (FUSE >= 0 && !SINK), "the event ring rides on the fused form, without a tuple sink");
  extern __shared__ __align__(64) uint32_t smem[];
  if constexpr (FUSE >= 0 && !SINK) S.dev = nullptr;      // (the scanning form, FUSE < 0, only flushes a count: one kernel for both)
  const int o = FUSE >= 0 ? FUSE : (cfg.orientation == DCRX_ORIENT_FORWARD ? 0 : 1);
  V2Ori V0 = T0.v2[0];
  if (o) V0 = T0.v2[1];
  uint32_t *lds_trans = smem;                                         // the pair table, at LDS address 0
  uint32_t *lds_counts = smem + V0.trans_bytes / 4;
  uint32_t *lds_work = lds_counts + DCRX_N_COUNTERS;
  const int tid = threadIdx.x;
#ifdef DCRX_SCAN_STAMPS
  const unsigned long long stamp_r0 = __builtin_amdgcn_s_memrealtime();      // instrumented build (tools/): per wave, 100 MHz ticks
  unsigned long long stamp_setup = 0, stamp_loop_end = 0;
#endif
  // (v2_entry reads the table at absolute LDS addresses: the dynamic segment starts at 0 because this kernel declares no static
  // LDS — launch_v2 checks that once per device, hipFuncGetAttributes, and refuses the launch otherwise: nothing traps here)
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  if (tid < V2_WK_WORDS) lds_work[tid] = 0;
  {      // (the block may be launched with fewer threads than it is compiled for: the stride is the block's own size)
    const uint4 *src = reinterpret_cast<const uint4 *>(V0.trans);
    uint4 *dst = reinterpret_cast<uint4 *>(lds_trans);
    for (uint32_t i = tid; i < V0.trans_bytes / 16; i += blockDim.x) dst[i] = src[i];
  }
  const V2Tab tab{};
  // the fused form: side tables and keyword buckets behind the work words (what the lean tail reads), then the ring
  uint32_t *lds_side = lds_work + V2_WK_WORDS;
  uint32_t *lds_bk = lds_side + (T0.lds_image_bytes - T0.dfa_bytes) / 4;
  uint32_t *ring = lds_bk + V0.bk_bytes / 4;
  const uint32_t ring_mask = 64u * ring_batches - 1u;
  constexpr uint32_t ES = (uint32_t)v2_ering_stride<NW>(), ering_mask = 64u * V2_ERING_BATCHES - 1u;
  uint32_t *ring2 = ring + (size_t)(ring_mask + 1u) * V2_RING_STRIDE;            // (FUSE_E) the event ring, then a block of counters nobody reads
  uint32_t *lds_dry = ring2 + (size_t)(ering_mask + 1u) * ES;
  if (FUSE >= 0) {
    {      // (the stride is the block's own size, as above)
      const uint4 *s1 = reinterpret_cast<const uint4 *>(T0.image + T0.dfa_bytes), *s2 = reinterpret_cast<const uint4 *>(V0.bk);
      uint4 *d1 = reinterpret_cast<uint4 *>(lds_side), *d2 = reinterpret_cast<uint4 *>(lds_bk);
      for (uint32_t i = tid; i < (T0.lds_image_bytes - T0.dfa_bytes) / 16; i += blockDim.x) d1[i] = s1[i];
      for (uint32_t i = tid; i < V0.bk_bytes / 16; i += blockDim.x) d2[i] = s2[i];
    }
    for (uint32_t i = tid; i <= ring_mask; i += blockDim.x) { ring[i * V2_RING_STRIDE + NW] = 0u; ring[i * V2_RING_STRIDE + NW + 1] = 0u; }
    if (FUSE_E) for (uint32_t i = tid; i <= ering_mask; i += blockDim.x) { ring2[i * ES + NW] = 0u; ring2[i * ES + NW + 1] = 0u; }
  }
  __syncthreads();
  const uint32_t nw = B.stride >> 2;
  const int lane = tid & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const bool tagged = B.n_reads < (1ull << 30);

  // The block owns the reads [blk_lo, blk_hi) and one region of each list; its waves draw items of 64 * RPL reads
  // from a counter in LDS.  (The four waves that share a SIMD are served oldest first: with the same reads for each, the
  // first ended a quarter earlier than the last — 148 against 193 us — and idled until the launch was over.)
  const uint64_t blk_lo = (uint64_t)blockIdx.x * per_block;
  const uint64_t blk_hi = blk_lo + per_block < B.n_reads ? blk_lo + per_block : B.n_reads;
  // the reads with exception bytes of this range: marked here, cleared below (the bitmap is all zero between launches)
  uint32_t e_lo = 0u, e_hi = 0u;
  if (B.n_exc) {
    v2_exc_slice(B, blk_lo < blk_hi ? blk_lo : blk_hi, blk_hi, lds_work + V2_WK_EXC, tid);
    e_lo = lds_work[V2_WK_EXC]; e_hi = lds_work[V2_WK_EXC + 2];
    v2_exc_marks(B, e_lo, e_hi, true, tid);
    // the marks are atomics at the L2 this unit reads through: they need only have been performed before a wave of this
    // block fetches them with its first item (no write-back of the L2, no invalidation: __threadfence() here cost the scan
    // a quarter of its time, every wave of the chip writing back an L2 full of the previous step's records)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  uint32_t tw = tail_waves;      // (0: the block's own choice, from what its region held last time)
  if (FUSE >= 0 && tw == 0u) {
    const uint32_t h = Q.counts[V2_L_COUNTS * blockIdx.x + V2_L_TWHINT + (o ? 0 : 1)];      // (a hint per frame: the two passes of orientation `both` meet different shares)
    tw = (h >= 2u && h <= 8u) ? h : (uint32_t)V2_FUSE_TAILWAVES;
    if (FUSE_E && tw > 2u) tw -= 1u;      // (fewer waves scan: the tail entries come in more slowly — 0.3158 against 0.3291 ms on two instead of three)
  }
  const uint32_t rw = FUSE_E ? rescue_waves : 0u;      // (FUSE_E) the block's last waves take the event ring
  const uint32_t n_scan_waves = (blockDim.x >> 6) - (FUSE >= 0 ? tw : 0u) - rw;
  const size_t region = blockIdx.x;
#ifdef DCRX_SCAN_STAMPS
  stamp_setup = __builtin_amdgcn_s_memrealtime();
#endif
  // The block's waves by role: scanning waves first; where the fused forms are compiled in, the last `rw` waves take the event ring and
  // the `tw` before them the tail ring from the start.  A scanning wave that has left its loop (and signed off) does not idle at the
  // barrier while the rings still hold entries: it goes through the rescue loop and then the tail loop as well (round 6: the rings'
  // last batches otherwise wait for the three and two waves of those roles, one batch after the other, while eleven waves stand by).
  const uint32_t wv = (uint32_t)(tid >> 6);
  const bool scans = wv < n_scan_waves;
  if (scans) {
#ifdef DCRX_V2_PRIO_SCAN
  if (FUSE >= 0) __builtin_amdgcn_s_setprio(DCRX_V2_PRIO_SCAN);
#endif
  constexpr uint32_t WT = 64u * RPL;
  const uint32_t n_items = blk_lo < blk_hi ? (uint32_t)((blk_hi - blk_lo + WT - 1) / WT) : 0u;
  uint4 *tq = Q.tail + region * Q.tcap * V2Rows<NW>::T;
  uint4 *eq = Q.ev + region * Q.ecap * V2Rows<NW>::E;
  auto draw = [&]() -> uint32_t {
    uint32_t i = 0;
    if (lane == 0) i = atomicAdd(&lds_work[V2_WK_NEXT], 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
  };

  const int npairs = UNIFORM_LEN ? (int)((B.read_len + 1u) >> 1) : 8 * (int)(nw < (uint32_t)NW ? nw : (uint32_t)NW);
  uint32_t w[RPL][NW];
  unsigned long long xm[RPL];
  uint32_t item = draw();
  if (PREFETCH && item < n_items) v2_load_item<NW, RPL>(B, nw, blk_lo + (uint64_t)item * WT, blk_hi, lane, w, xm);
  if (PREFETCH) {
    // The first item's words are waited for HERE, and the compiler is told so (an empty statement that takes and gives back each
    // register).  Without it the loop below saw "w may still be in flight" on its entry edge and put `s_waitcnt vmcnt(0)` in front
    // of the first look-up of EVERY item — behind the loads of the next item, issued a moment before: each wave then sat out a whole
    // memory round trip per item with nothing in flight, and the item "in flight during the scan" was in flight during nothing
    // (round 6; the counter counts loads in order and the conditional loads in between leave the pass no count to wait for).
#pragma unroll
    for (int q = 0; q < RPL; q++) {
#pragma unroll
      for (int k = 0; k < NW; k++) asm volatile("" : "+v"(w[q][k]));
      asm volatile("" : "+v"(xm[q]));
    }
  }
  while (item < n_items) {
    uint32_t wn[RPL][NW];
    unsigned long long xn[RPL];
    const uint32_t next = draw();
    const bool more = next < n_items;
    if (!PREFETCH) v2_load_item<NW, RPL>(B, nw, blk_lo + (uint64_t)item * WT, blk_hi, lane, w, xm);
    if (PREFETCH) { if (more) v2_load_item<NW, RPL>(B, nw, blk_lo + (uint64_t)next * WT, blk_hi, lane, wn, xn); }     // in flight while this item is scanned
    uint32_t lg[RPL][NW];
    // (the look-up loop is bound by the LDS; what a wave does between two scans — digest, pushes, records — is bound by
    // issue: a wave inside the loop goes first on its SIMD, so that the block's look-ups stay in flight: - 2 % of the step)
#if DCRX_V2_LOOP_PRIO
    __builtin_amdgcn_s_setprio(DCRX_V2_LOOP_PRIO);
#endif
    if (UNIFORM_LEN && NW == 10 && npairs == 75) scan2<NW, RPL, NARROW, (UNIFORM_LEN && NW == 10) ? 75 : 0>(tab, w, lg, npairs);      // (150 nt: the count known to the compiler)
    else scan2<NW, RPL, NARROW>(tab, w, lg, npairs);
#if DCRX_V2_LOOP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // (round 6: the next item's words landed HERE instead — a whole scan behind their loads, in front of this item's stores, so
    // that the copy w = wn at the loop's end does not sit behind the stores' acknowledgements — measured 0.6 % slower, one register
    // more: profiles/r06/scan_early_landing_ab.log)
    unsigned long long tmask[RPL];      // (fused form) the tail lanes of each of the item's reads, and their digests
    uint32_t tdg[RPL];
#pragma unroll
    for (int q = 0; q < RPL; q++) { tmask[q] = 0ull; tdg[q] = 0u; }
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      const uint64_t r = blk_lo + (uint64_t)item * WT + (uint64_t)q * 64 + lane;
      bool live = r < blk_hi;
      // orientation `both`, second attempt (decombine.py:1005-1010): only the reads the first frame did not decombine are
      // tried again; the others keep their records and count nothing
      if (retry && live) live = records[r].status != DCRX_S_OK;
      const bool exc = live && ((xm[q] >> lane) & 1ull);
      const int n = UNIFORM_LEN ? (int)B.read_len : (live ? (int)B.lens[r] : 0);
      if (!UNIFORM_LEN) mask_log2<NW>(lg[q], n);
      const Digest2 d = digest2_scan<NW>(lg[q]);
      if (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) {        // profiling aid: price the scan alone (records are NOT results)
        if (live) {
          __align__(16) dcrx_record_t rec;
          rec.v = (uint16_t)d.any; rec.j = (uint16_t)d.vf_n; rec.v_start = (uint16_t)d.vf_pair; rec.j_end = (uint16_t)d.jf_n;
          rec.ins_start = (uint16_t)d.jf_pair; rec.ins_len = 0; rec.vdel = rec.jdel = 0; rec.status = 254; rec.frame = 0;
          DCRX_STORE_SCAN(records + r, rec);
        }
        continue;
      }
      // A read with exception bytes was scanned with those bytes packed as A: its flags hold every
      // keyword that really occurs (and possibly more), so "no V flag" still means "no V keyword";
      // anything else is resolved from its events against its exception list.
      const uint32_t bnd = (n & 1) ? log_nibble<NW>(lg[q], n >> 1) : 0u;
      int what = live ? classify2(d, bnd) : -1;
      if (exc && what != V2_VNONE) what = V2_EVENTS;
      const bool vnone = what == V2_VNONE, vmulti = what == V2_VMULTI;
#if DCRX_V2_FULL_LINE_RECORDS
      // every lane writes: whole lines of records leave the wave (reads that go on get a placeholder, rewritten by the kernel that
      // settles them) — but for the tail reads of the fused form: a tail wave of this block writes their records, and two stores to
      // one address from two waves have no order
      if (live && !(FUSE >= 0 && what == V2_TAIL)) {
#else
      if (vnone || vmulti) {
#endif
        __align__(16) dcrx_record_t rec;
        rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
        rec.vdel = rec.jdel = 0;
        rec.status = (uint8_t)(vnone ? DCRX_S_V_NONE : (vmulti ? DCRX_S_V_MULTI : DCRX_S_DEFER));
        rec.frame = (uint8_t)(o == 0 ? 1 : 0);
        DCRX_STORE_SCAN(records + r, rec);
      }
      const unsigned long long mn = __ballot(vnone), mm = __ballot(vmulti);
      if (lane == 0) {
        if (mn) atomicAdd(&lds_counts[DCRX_C_NO_VTAGS_FOUND], (uint32_t)__popcll(mn));
        if (mm) atomicAdd(&lds_counts[DCRX_C_MULTIPLE_V_MATCHES], (uint32_t)__popcll(mm));
        if (mn | mm) atomicAdd(&lds_counts[DCRX_C_READ_COUNT], (uint32_t)__popcll(mn | mm));
      }
      if (cfg.flags & DCRX_F_PROFILE_NO_FINISH) continue;
      bool to_tail = what == V2_TAIL;
      const unsigned long long mt0 = __ballot(to_tail);
      if (FUSE >= 0) {             // the fused form: the item's tail entries go into the block's ring together, behind the loop over its reads
        tmask[q] = mt0; tdg[q] = tail2_pack(d);
      } else if (mt0) {           // the block's tail list: one LDS atomic per wave and group of 64 reads
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&lds_work[V2_WK_LIST + V2_L_TAIL], (uint32_t)__popcll(mt0));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t at = base + (uint32_t)__popcll(mt0 & lt_mask);
        if (to_tail && at >= Q.tcap) { v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, (uint32_t)r, false); to_tail = false; }
        if (to_tail) {
          uint32_t x[2 + NW];
          x[0] = (uint32_t)r; x[1] = tail2_pack(d);
#pragma unroll
          for (int k = 0; k < NW; k++) x[2 + k] = w[q][k];
          v2_put_rows<2 + NW>(tq, Q.tcap, at, x);
        }
      }
      // event entries: list E (one gene to rescue: nine in ten), or one of the rare lists C (both genes) and X (exception
      // bytes, a flag on the last half pair of an odd read: the general form)
      int lst = 0;
      if (what == V2_EVENTS) lst = (exc || bnd) ? V2_L_X : (shape2(d.vf_n, d.jf_n, d.any) == V2_SHAPE_BOTH ? V2_L_C : V2_L_E);
      auto put_event = [&](uint4 *rows, const uint32_t at) {      // (behind the words: the digest of the log, for the lean rescue)
        uint32_t x[2 + 2 * NW];
        x[0] = (uint32_t)r | (exc ? V2_R_EXC : 0u);
#pragma unroll
        for (int k = 0; k < NW; k++) { x[1 + k] = lg[q][k]; x[1 + NW + k] = w[q][k]; }
        x[1 + 2 * NW] = rescue2_digest_pack(d);
        v2_put_rows<2 + 2 * NW>(rows, 0u, at, x);
      };
      bool to_ev = lst == V2_L_E;
      const unsigned long long me0 = __ballot(to_ev);
      if (FUSE_E) {
        // list E's entries of this read go into the block's event ring (one transaction: the slots of the wave's E lanes drawn with
        // one LDS atomic, every ring batch's GEN fetched in the same wait; the entries; the batches' FILLED counts behind them)
        if (me0) {
          constexpr uint32_t nb_mask = V2_ERING_BATCHES - 1u, nb_shift = (uint32_t)__builtin_ctz(V2_ERING_BATCHES);
          const uint32_t total = (uint32_t)__popcll(me0);
          const uint32_t genv = __hip_atomic_load(&lds_work[V2_WK_GEN2 + ((uint32_t)lane & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&lds_work[V2_WK_HEAD2], total);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          const uint32_t gb0 = base >> 6, gbl = (base + total - 1u) >> 6;
          bool room = true;
          for (uint32_t g = gb0; g <= gbl; g++) room = room && (uint32_t)__builtin_amdgcn_readlane((int)genv, (int)(g & nb_mask)) >= (g >> nb_shift);
          if (!room && lane == 0) {      // a ring batch's last occupants are still being finished: wait for them
            bool ok = false;
            for (uint32_t spins = 0; spins < (1u << 24) && !ok; spins++) {
              ok = true;
              for (uint32_t g = gb0; g <= gbl; g++)
                ok = ok && __hip_atomic_load(&lds_work[V2_WK_GEN2 + (g & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (g >> nb_shift);
              if (!ok) __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) atomicAdd(&counters[DCRX_C_DEVICE_ERRORS], 1ull);
          }
          asm volatile("" ::: "memory");
          if (to_ev) {
            uint32_t *sl = ring2 + ((base + (uint32_t)__popcll(me0 & lt_mask)) & ering_mask) * ES;
#pragma unroll
            for (int k = 0; k < NW; k++) { sl[k] = w[q][k]; sl[NW + 3 + k] = lg[q][k]; }
            sl[NW + 2] = (uint32_t)r; sl[2 * NW + 3] = rescue2_digest_pack(d);
          }
          asm volatile("" ::: "memory");       // (the entries before the counts that announce them, in program order: LDS keeps it)
          if (lane == 0) {
            for (uint32_t g = gb0; g <= gbl; g++) {
              const uint32_t lo = max(base, g << 6), hi = min(base + total, (g + 1u) << 6);
              atomicAdd(&lds_work[V2_WK_FILLED2 + (g & nb_mask)], hi - lo);
            }
          }
        }
      } else if (me0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&lds_work[V2_WK_LIST + V2_L_E], (uint32_t)__popcll(me0));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t at = base + (uint32_t)__popcll(me0 & lt_mask);
        if (to_ev && at >= Q.ecap) {                 // a full region (its last lanes): the three-launch form
          v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, (uint32_t)r, exc);
          to_ev = false;
        }
        if (to_ev) put_event(eq, at);
      }
      if (__ballot(lst > V2_L_E)) {
#pragma unroll 1
        for (int which = V2_L_C; which <= V2_L_X; which++) {
          const bool mine = lst == which;
          const unsigned long long mr = __ballot(mine);
          if (!mr) continue;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&lds_work[V2_WK_LIST + which], (uint32_t)__popcll(mr));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (mine) {
            const V2ListRef l = v2_list<NW>(Q, which, region);
            const uint32_t at = base + (uint32_t)__popcll(mr & lt_mask);
            if (at >= l.cap) v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, (uint32_t)r, exc);
            else put_event(l.rows, at);
          }
        }
      }
    }
    if constexpr (FUSE >= 0) {
      // One ring transaction per item: the slots of all its tail lanes drawn with one LDS atomic, and with it — the same wait —
      // every ring batch's GEN (lane l reads GEN[l % NB]: a value read a moment early only errs towards waiting); the entries;
      // the batches' FILLED counts behind them (LDS keeps a wave's order: nothing to wait for).  Per read of the item the same
      // cost 43 us of a 10 M-read step: three dependent LDS round trips behind the look-ups of fifteen other waves, twice per item.
      // (List E's draw — one round trip per read of the item, above — moved into this transaction as well: the scan 4-6 us LONGER,
      // the masks and digests of both reads kept alive for it; profiles/r05/list_e_draw_in_the_ring_transaction_ab.log.)
      uint32_t cnts[RPL], total = 0;
#pragma unroll
      for (int q = 0; q < RPL; q++) { cnts[q] = (uint32_t)__popcll(tmask[q]); total += cnts[q]; }
      if (total) {
        const uint32_t nb_mask = ring_batches - 1u, nb_shift = (uint32_t)__builtin_ctz(ring_batches);
        const uint32_t genv = __hip_atomic_load(&lds_work[V2_WK_GEN + ((uint32_t)lane & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&lds_work[V2_WK_HEAD], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t gb0 = base >> 6, gbl = (base + total - 1u) >> 6;
        bool room = true;
        for (uint32_t g = gb0; g <= gbl; g++) room = room && (uint32_t)__builtin_amdgcn_readlane((int)genv, (int)(g & nb_mask)) >= (g >> nb_shift);
        if (!room && lane == 0) {      // a ring batch's last occupants are still being finished: wait for them
          bool ok = false;
          for (uint32_t spins = 0; spins < (1u << 24) && !ok; spins++) {      // (the bound is never reached: a tail wave that does not come back)
            ok = true;
            for (uint32_t g = gb0; g <= gbl; g++)
              ok = ok && __hip_atomic_load(&lds_work[V2_WK_GEN + (g & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (g >> nb_shift);
            if (!ok) __builtin_amdgcn_s_sleep(2);
          }
          if (!ok) atomicAdd(&counters[DCRX_C_DEVICE_ERRORS], 1ull);
        }
        asm volatile("" ::: "memory");
        uint32_t at0 = base;
#pragma unroll
        for (int q = 0; q < RPL; q++) {
          if ((tmask[q] >> lane) & 1ull) {
            uint32_t *sl = ring + ((at0 + (uint32_t)__popcll(tmask[q] & lt_mask)) & ring_mask) * V2_RING_STRIDE;
#pragma unroll
            for (int k = 0; k < NW; k++) sl[k] = w[q][k];
            sl[NW + 2] = (uint32_t)(blk_lo + (uint64_t)item * WT + (uint64_t)q * 64 + lane); sl[NW + 3] = tdg[q];
          }
          at0 += cnts[q];
        }
        asm volatile("" ::: "memory");       // (the entries before the counts that announce them, in program order: LDS keeps it)
        if (lane == 0) {
          for (uint32_t g = gb0; g <= gbl; g++) {
            const uint32_t lo = max(base, g << 6), hi = min(base + total, (g + 1u) << 6);
            atomicAdd(&lds_work[V2_WK_FILLED + (g & nb_mask)], hi - lo);
          }
        }
      }
    }
    if (PREFETCH && more) {
#pragma unroll
      for (int q = 0; q < RPL; q++) {
#pragma unroll
        for (int k = 0; k < NW; k++) w[q][k] = wn[q][k];
        xm[q] = xn[q];
      }
    }
    item = next;
  }
  if (FUSE >= 0 && lane == 0) {      // this wave's last entries are in the ring
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    atomicAdd(&lds_work[V2_WK_SCANNED], 1u);
  }
  }
  if (FUSE_E && (wv >= n_scan_waves + tw || (DCRX_V2_HELP_DRAIN && scans))) {
   if constexpr (FUSE_E) {
    // ---- a rescue wave (FUSE_E): batches of 64 event entries of list E out of the event ring, as the scanning waves fill them ----
    constexpr bool REV = FUSE == 1;
#if DCRX_V2_PRIO_TAIL
    __builtin_amdgcn_s_setprio(DCRX_V2_PRIO_TAIL);
#endif
    uint32_t kw_base[K_NCLASS];
#pragma unroll
    for (int c = 0; c < K_NCLASS; c++) kw_base[c] = T0.kw_base[c];
    const Rescue2Tabs rt = rescue2_tabs(T0, V0, reinterpret_cast<const uint8_t *>(lds_side), reinterpret_cast<const uint8_t *>(lds_bk), REV, kw_base);
    const Counters C{lds_counts}, Cdry{lds_dry};
    constexpr uint32_t nb_mask = V2_ERING_BATCHES - 1u, nb_shift = (uint32_t)__builtin_ctz(V2_ERING_BATCHES);
    constexpr uint32_t V2_RING_EXIT = 0xFFFFFFFFu;
    for (uint32_t spins = 0;;) {
      uint32_t c = 0, nvalid = 0;
      if (lane == 0) {      // (the tail ring's protocol on the event ring's words)
        c = __hip_atomic_load(&lds_work[V2_WK_CLAIM2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool mine = __hip_atomic_load(&lds_work[V2_WK_GEN2 + (c & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (c >> nb_shift);
        const uint32_t f = __hip_atomic_load(&lds_work[V2_WK_FILLED2 + (c & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!mine) nvalid = 0;
        else if (f == 64u) nvalid = 64u;
        else if (__hip_atomic_load(&lds_work[V2_WK_SCANNED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == n_scan_waves) {
          const uint32_t h = __hip_atomic_load(&lds_work[V2_WK_HEAD2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t rem = h - 64u * c;
          nvalid = (int32_t)rem <= 0 ? V2_RING_EXIT : min(rem, 64u);
        }
        if (nvalid && nvalid != V2_RING_EXIT) {
          uint32_t expect = c;
          if (!__hip_atomic_compare_exchange_strong(&lds_work[V2_WK_CLAIM2], &expect, c + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) nvalid = 0;
        }
      }
      c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
      nvalid = (uint32_t)__builtin_amdgcn_readfirstlane((int)nvalid);
      if (nvalid == V2_RING_EXIT) break;
      if (!nvalid) {
        if (++spins > (1u << 24)) { if (lane == 0) atomicAdd(&counters[DCRX_C_DEVICE_ERRORS], 1ull); break; }
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint32_t *sl = ring2 + ((64u * c + (uint32_t)lane) & ering_mask) * ES;
      int status = -2;
      uint32_t errs = 0, x0 = 0;
      uint32_t lg[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) lg[k] = 0u;
      if ((uint32_t)lane < nvalid) {
        x0 = sl[NW + 2];
#pragma unroll
        for (int k = 0; k < NW; k++) lg[k] = sl[NW + 3 + k];
        const uint32_t dg = sl[2 * NW + 3];
        const uint32_t r = x0 & V2_R_MASK;
        const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
        const LdsWords lw{dcrx_ldsaddr_of(sl)};
        auto on_ok = [&](dcrx_record_t rec, const uint32_t) {      // a decombined read's record, where its fields are known
          rec.frame = (uint8_t)(o ? 0 : 1);
          DCRX_STORE_FINISH(records + r, rec);
        };
        status = rescue2_fast_to<REV, NW, V2_SHAPE_ONE>(rt, lw, lg, n, cfg, on_ok, errs, *Tmem, C, Cdry, dg);
        if (status > 0) {      // settled, not decombined: the status alone
          dcrx_record_t rec;
          rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0; rec.vdel = rec.jdel = 0;
          rec.status = (uint8_t)status; rec.frame = (uint8_t)(o ? 0 : 1);
          DCRX_STORE_FINISH(records + r, rec);
        } else if (status < 0) errs = 0;
      }
      v2_tally_rescue(lds_counts, lane, status, errs, o == 0);
      if (__builtin_expect(status == RESCUE2_SLOW, 0)) {      // what the lean form does not settle: the launch's left list (its placeholder record stands)
        uint32_t ww[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) ww[k] = sl[k];
        if (!v2_left_push<NW>(Q.left, queue_count, x0, lg, ww)) v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, x0 & V2_R_MASK, false);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) {
        __hip_atomic_store(&lds_work[V2_WK_FILLED2 + (c & nb_mask)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        atomicAdd(&lds_work[V2_WK_GEN2 + (c & nb_mask)], 1u);
      }
      spins = 0;
    }
   }
  }
  if (FUSE >= 0 && (DCRX_V2_HELP_DRAIN || (wv >= n_scan_waves && wv < n_scan_waves + tw))) {
   if constexpr (FUSE >= 0) {
    // ---- a tail wave of the fused form: batches of 64 tail entries out of the ring, as the scanning waves fill them ----
    constexpr bool REV = FUSE == 1;
#if DCRX_V2_PRIO_TAIL
    __builtin_amdgcn_s_setprio(DCRX_V2_PRIO_TAIL);
#endif
    const Tail2Tabs tt = tail2_tabs(T0, V0, reinterpret_cast<const uint8_t *>(lds_side), reinterpret_cast<const uint8_t *>(lds_bk), REV);
    const Counters C{lds_counts};
    const uint32_t nb_mask = ring_batches - 1u, nb_shift = (uint32_t)__builtin_ctz(ring_batches);
    constexpr uint32_t V2_RING_EXIT = 0xFFFFFFFFu;
    for (uint32_t spins = 0;;) {
      uint32_t c = 0, nvalid = 0;
      if (lane == 0) {
        c = __hip_atomic_load(&lds_work[V2_WK_CLAIM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (FILLED speaks of batch c only once the ring batch's earlier occupants have been finished: with a short ring a wave may
        // still be at work on batch c - NB, its count not yet taken back)
        const bool mine = __hip_atomic_load(&lds_work[V2_WK_GEN + (c & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (c >> nb_shift);
        const uint32_t f = __hip_atomic_load(&lds_work[V2_WK_FILLED + (c & nb_mask)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!mine) nvalid = 0;
        else if (f == 64u) nvalid = 64u;
        else if (__hip_atomic_load(&lds_work[V2_WK_SCANNED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == n_scan_waves) {
          // every scanning wave has signed off (its last entries and its share of FILLED before that): what is left is final
          const uint32_t h = __hip_atomic_load(&lds_work[V2_WK_HEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t rem = h - 64u * c;
          nvalid = (int32_t)rem <= 0 ? V2_RING_EXIT : min(rem, 64u);
        }
        if (nvalid && nvalid != V2_RING_EXIT) {
          uint32_t expect = c;
          if (!__hip_atomic_compare_exchange_strong(&lds_work[V2_WK_CLAIM], &expect, c + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) nvalid = 0;
        }
      }
      c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
      nvalid = (uint32_t)__builtin_amdgcn_readfirstlane((int)nvalid);
      if (nvalid == V2_RING_EXIT) break;
      if (!nvalid) {
        if (++spins > (1u << 24)) {           // (never seen: a scanning wave that does not sign off) — said in the call's counters, not passed over in silence
          if (lane == 0) atomicAdd(&counters[DCRX_C_DEVICE_ERRORS], 1ull);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the entries were written before FILLED said so: nothing is read early)
      uint32_t *sl = ring + ((64u * c + (uint32_t)lane) & ring_mask) * V2_RING_STRIDE;
      int status = -2;
      uint32_t r = 0, dg = 0;
      uint64_t tup = 0;
      if ((uint32_t)lane < nvalid) {
        r = sl[NW + 2]; dg = sl[NW + 3];
        const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
        dcrx_record_t rec;
        rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0; rec.vdel = rec.jdel = 0;
        const LdsWords lw{dcrx_ldsaddr_of(sl)};
        if (cfg.flags & DCRX_F_PROFILE_TAIL_STREAM_ONLY) { status = DCRX_S_J_NONE; rec.v = (uint16_t)(sl[0] ^ sl[NW - 1]); }      // profiling: the ring without the arithmetic
        else
        status = tail2_fast<REV>(tt, lw, n, dg, cfg, rec, *Tmem, C);
        rec.frame = (uint8_t)(o ? 0 : 1);
        if (status >= 0) { rec.status = (uint8_t)status; DCRX_STORE_FINISH(records + r, rec); }
        if (S.dev && status == DCRX_S_OK) tup = sink_tuple_lean(rec, S.wpack, false);      // (a tail read's J tag is whole)
      }
      v2_tally(lds_counts, lane, status, o == 0);
      // tuple sink: the item of every entry of the batch, at the entry's place in the ring's sequence
      if (S.dev) sink_put(S, (uint32_t)region, 0u, 64u * c + (uint32_t)lane, (uint32_t)lane < nvalid, status == DCRX_S_OK, r, tup, lane, &lds_work[V2_WK_SINK]);
      if (__builtin_expect(status == TAIL2_SLOW, 0)) {
        // what the lean form does not settle (one read in millions): an event entry of the launch's left list (the finishing
        // launch's polling wave takes it), behind a placeholder record
        __align__(16) dcrx_record_t rec;
        rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0; rec.vdel = rec.jdel = 0;
        rec.status = (uint8_t)DCRX_S_DEFER; rec.frame = (uint8_t)(o ? 0 : 1);
        DCRX_STORE_FINISH(records + r, rec);
        const uint32_t vp = dg & 0xFFu, jp = (dg >> 8) & 0xFFu, jc = (dg >> 16) & 3u;
        uint32_t lg[NW], ww[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) {
          uint32_t l = (vp >> 3) == (uint32_t)k ? (V2_F_VF << (4 * (vp & 7u))) : 0u;
          if (jc == 1u && (jp >> 3) == (uint32_t)k) l |= V2_F_JF << (4 * (jp & 7u));
          lg[k] = l; ww[k] = sl[k];
        }
        if (!v2_left_push<NW>(Q.left, queue_count, r | (jc == 2u ? V2_R_JMULTI : 0u), lg, ww)) v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, r, false);
      }
      // the ring batch is free again: FILLED back to zero, then GEN (a scanning wave looks at GEN first)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) {
        __hip_atomic_store(&lds_work[V2_WK_FILLED + (c & nb_mask)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        atomicAdd(&lds_work[V2_WK_GEN + (c & nb_mask)], 1u);
      }
      spins = 0;
    }
   }
  }
#ifdef DCRX_SCAN_STAMPS
  stamp_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();
  if (B.n_exc) v2_exc_marks(B, e_lo, e_hi, false, tid);      // the marks have served (an entry carries V2_R_EXC from here on)
#ifdef DCRX_SCAN_STAMPS
  if (lane == 0) {      // instrumented build (tools/): per wave start, set-up done, main loop left, end — 100 MHz ticks — into a buffer of the build's own
    uint32_t *c = g_scan_stamps + 4 * ((size_t)blockIdx.x * (DCRX_V2_BLOCK / 64) + (size_t)(tid >> 6));
    c[0] = (uint32_t)stamp_r0; c[1] = (uint32_t)stamp_setup; c[2] = (uint32_t)stamp_loop_end; c[3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
  }
#else
  if (tid < V2_L_COUNTS) {      // (entries beyond a list's capacity were handed over, not stored; L starts empty)
    uint32_t c = tid <= V2_L_X ? lds_work[V2_WK_LIST + tid] : 0u;
    c = min(c, tid == V2_L_TAIL ? Q.tcap : (tid == V2_L_E ? Q.ecap : Q.scap / 2));
    if (tid == V2_L_RING) c = lds_work[V2_WK_HEAD];      // (fused form: entries that went through the ring — the tuple sink's tail section)
    if ((tid == V2_L_TWHINT || tid == V2_L_TWHINT + 1) && FUSE >= 0) c = Q.counts[V2_L_COUNTS * region + tid];      // (the other frame's hint stays)
    if (tid == V2_L_TWHINT + (o ? 0 : 1) && FUSE >= 0 && blk_hi > blk_lo) {      // the next launch's tail waves in this frame, by this one's share of tail reads
      const uint32_t frac256 = (uint32_t)(((uint64_t)lds_work[V2_WK_HEAD] << 8) / (blk_hi - blk_lo));
      c = frac256 >= V2_TW6_FRAC256 ? 6u : (frac256 >= V2_TW5_FRAC256 ? 5u : (frac256 >= V2_TW4_FRAC256 ? 4u : (frac256 >= V2_TW3_FRAC256 ? 3u : 2u)));
    }
    Q.counts[V2_L_COUNTS * region + tid] = c;
  }
  if (S.dev && tid == 0 && lds_work[V2_WK_SINK]) (void)__hip_atomic_fetch_add(S.hits + region, lds_work[V2_WK_SINK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
}

// LDS of the finishing kernel: counters, side tables, the frame's keyword buckets
static uint32_t v2_finish_lds_bytes(const DevTables &T, int o) {
  return DCRX_N_COUNTERS * 4 + (T.lds_image_bytes - T.dfa_bytes) + T.v2[o].bk_bytes;
}

// One entry through the general form (dcr_frame3): the read's event list from its flag log, its slice of the exception list,
// finish2_words; what even that form does not hold (more flagged pairs than an event list, more exception bytes than the
// register frame, more hits than a hit list) is handed to the list kernel.
template <bool UNIFORM_LEN, int NW, int ORI>
__device__ __forceinline__ void v2_general_entry(const DevTables &T, const V2Ori &V, const BatchDev &B, const CfgDev &cfg, const uint32_t x0,
                                                 const uint32_t (&lg)[NW], const uint32_t (&w)[NW], const Counters &C, dcrx_record_t *__restrict__ records,
                                                 uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue, const uint32_t qcap,
                                                 uint32_t *__restrict__ queue_count, const bool tagged) {
  const uint32_t r = x0 & V2_R_MASK;
  const bool exc = (x0 & V2_R_EXC) != 0u;
  const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  // the read's event list, from its flag log (a read with exception bytes keeps every flag: none is certain)
  const Digest2 d = digest2<NW>(lg);
  const uint32_t bnd = (n & 1) ? log_nibble<NW>(lg, n >> 1) : 0u;
  uint32_t ev[3];
  const bool fits = events2<NW>(lg, d, exc ? 0xFu : bnd, ev);
  if (!fits) { v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, r, exc); return; }   // more flagged pairs than a list holds: the three-launch form
  int x0e = 0, x1e = 0;
  if (exc) {                      // the read's slice of the (sorted) exception list
    uint64_t lo = 0, hi = B.n_exc;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (B.exc_read[mid] < r) lo = mid + 1; else hi = mid; }
    x0e = (int)lo;
    while (lo < B.n_exc && B.exc_read[lo] == r) lo++;
    x1e = (int)lo;
    ExcLayout xl;
    if (x1e - x0e > V2_MAX_EXC && !exc_layout(B.exc_pos, B.exc_chr, x0e, x1e, xl)) {      // more than a run of Ns and four single bytes: the list kernel
      v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, r, true); return;
    }
  }
  if (!finish2_words<UNIFORM_LEN, NW, ORI>(T, V, B, cfg, (uint64_t)r, w, ev, (x0 & V2_R_JMULTI) != 0u, x0e, x1e, C, records))
    v2_hand_over(B, queue, gqueue, qcap, queue_count, tagged, r, exc);
}

// ---- the finishing roles ---------------------------------------------------------------------------
// What a block of a finishing kernel holds in LDS: counters, side tables, the frame's keyword buckets, a strip per lane for
// the read in hand, the scratch counters of the lean rescue; per wave the slots of the entries its lean form left.
constexpr int V2_LEFT_SLOTS = 15;      // entries a wave may note per job; more (never seen) go to the list kernel
struct V2FinishLds {
  uint32_t *counts, *side, *bk, *strip, *dry, *left;      // left: this wave's 1 + V2_LEFT_SLOTS words
};
// where those pieces lie in the block's dynamic LDS (addresses only) ...
template <int NW, int BLOCK>
__device__ __forceinline__ V2FinishLds v2_finish_layout(const DevTables &T0, const V2Ori &V, uint32_t *smem, const int tid) {
  V2FinishLds L;
  L.counts = smem;
  L.side = smem + DCRX_N_COUNTERS;
  L.bk = L.side + (T0.lds_image_bytes - T0.dfa_bytes) / 4;
  // each lane's strip of LDS for the read in hand
  L.strip = L.bk + V.bk_bytes / 4 + (uint32_t)tid * lds_words_stride<NW>();
  // behind the strips: a block of counters nobody reads (what a walk counts that turns out not to be final)
  L.dry = L.bk + V.bk_bytes / 4 + (uint32_t)BLOCK * lds_words_stride<NW>();
  L.left = L.dry + DCRX_N_COUNTERS + (uint32_t)(tid >> 6) * (1 + V2_LEFT_SLOTS);
  return L;
}
// ... and their staging (the caller synchronises)
template <int NW, int BLOCK>
__device__ __forceinline__ V2FinishLds v2_finish_stage(const DevTables &T0, const V2Ori &V, uint32_t *smem, const int tid) {
  const V2FinishLds L = v2_finish_layout<NW, BLOCK>(T0, V, smem, tid);
  if (tid < DCRX_N_COUNTERS) L.counts[tid] = 0;
  stage_lds<BLOCK>(T0.image + T0.dfa_bytes, L.side, (T0.lds_image_bytes - T0.dfa_bytes) / 16, 0, 0, tid);
  stage_lds<BLOCK>(V.bk, L.bk, V.bk_bytes / 16, 0, 0, tid);
  L.strip[NW] = 0u; L.strip[NW + 1] = 0u;      // (the strip's two zero words are written once)
  if ((tid & 63) == 0) L.left[0] = 0u;
  return L;
}
// (bytes of that image: the launcher's size of the dynamic LDS)
template <int NW>
static uint32_t v2_finish_block_lds(const DevTables &T, int o, int block) {
  return DCRX_N_COUNTERS * 4 + (T.lds_image_bytes - T.dfa_bytes) + T.v2[o].bk_bytes + (uint32_t)block * lds_words_stride<NW>() * 4 + DCRX_N_COUNTERS * 4 +
         (uint32_t)(block / 64) * (1 + V2_LEFT_SLOTS) * 4;
}

// a lane notes the slot of an entry its lean form did not settle (LDS, per wave)
__device__ __forceinline__ void v2_note_left(uint32_t *left, const uint32_t slot, const BatchDev &B, uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue,
                                             const uint32_t qcap, uint32_t *__restrict__ queue_count, const uint32_t r, const bool exc) {
  const uint32_t at = atomicAdd(&left[0], 1u);
  if (at < (uint32_t)V2_LEFT_SLOTS) left[1 + at] = slot;
  else v2_hand_over(B, queue, gqueue, qcap, queue_count, B.n_reads < (1ull << 30), r, exc);
}
template <int NW>
struct V2EntryWords { uint32_t lg[NW], w[NW]; };
// The lean tail: wave `gwave` of `n_gwaves` takes the jobs (region, part) gwave, gwave + n_gwaves, ...: `split` waves share a
// region (a scan block's list), wave k of them its batches k, k + split, ...  Software pipeline over the batches of 64: the
// entries are read one batch ahead, the read in hand sits in the lane's LDS strip.  What the lean form does not settle is
// noted per wave and pushed to the launch's left list behind the job's loop (v2_left_push).
template <bool UNIFORM_LEN, int NW, int ORI>
__device__ __forceinline__ void v2_tail_jobs(const Tail2Tabs &tt, const V2FinishLds &L, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *__restrict__ records,
                                             const V2Lists &Q, const uint32_t n_regions, const uint32_t split, uint32_t *__restrict__ queue,
                                             uint32_t *__restrict__ gqueue, const uint32_t qcap, uint32_t *__restrict__ queue_count,
                                             const DevTables *__restrict__ Tmem, const uint32_t gwave, const uint32_t n_gwaves, const int tid, const V2SinkCall &S) {
  constexpr int o = ORI;
  const Counters C{L.counts};
  const LdsWords lw{dcrx_ldsaddr_of(L.strip)};
  uint32_t *strip = L.strip, *lds_counts = L.counts;
  uint32_t *s_left = L.left;
  const int lane = tid & 63;
  for (uint32_t job = gwave; job < n_regions * split; job += n_gwaves) {
    const uint32_t region = job / split, part = job % split;
    const uint32_t tn = Q.counts[V2_L_COUNTS * region + V2_L_TAIL];
    const uint4 *tq = Q.tail + (size_t)region * Q.tcap * V2Rows<NW>::T;
    const uint32_t STEP = 64 * split;
    constexpr bool AHEAD = NW <= DCRX_NWMAX;      // (the longest reads: no look-ahead, the registers do not hold two entries)
    uint32_t x1[2 + NW];
    if constexpr (AHEAD) v2_get_rows<2 + NW>(tq, Q.tcap, 64 * part + lane, 64 * part + lane < tn, x1);
    for (uint32_t first = 64 * part; first < tn && !(cfg.flags & DCRX_F_PROFILE_NO_TAIL); first += STEP) {
      uint32_t x[2 + NW];
      if constexpr (AHEAD) {
#pragma unroll
        for (int k = 0; k < 2 + NW; k++) x[k] = x1[k];
        const uint32_t nf = v2_landed(x, first + STEP);
        v2_get_rows<2 + NW>(tq, Q.tcap, nf + lane, nf + lane < tn, x1);     // the next batch, in flight during this one
      } else {
        v2_get_rows<2 + NW>(tq, Q.tcap, first + lane, first + lane < tn, x);
      }
      uint32_t w[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) w[k] = x[2 + k];
      int status = -2;
      uint64_t tup = 0;
      const uint32_t r = x[0], dg = x[1];
      if (first + lane < tn) {
        const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
        dcrx_record_t rec;
        rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0; rec.vdel = rec.jdel = 0;
#if DCRX_LEAN_LDS_WORDS
#pragma unroll
        for (int k = 0; k < NW; k++) strip[k] = w[k];
        if (cfg.flags & DCRX_F_PROFILE_TAIL_STREAM_ONLY) { status = DCRX_S_J_NONE; rec.v = (uint16_t)(w[0] ^ w[NW - 1]); }
        else
        status = tail2_fast<ORI == 1>(tt, lw, n, dg, cfg, rec, *Tmem, C);      // (the tables through memory: only a walk that leaves its window reads them)
#else
        const RegWords<NW> rw{w};
        status = tail2_fast<ORI == 1>(tt, rw, n, dg, cfg, rec, *Tmem, C);
#endif
        if (status >= 0) { rec.status = (uint8_t)status; rec.frame = (uint8_t)(o ? 0 : 1); DCRX_STORE_FINISH(records + r, rec); }
        if (S.dev && status == DCRX_S_OK) tup = sink_tuple_lean(rec, S.wpack, false);
      }
      v2_tally(lds_counts, lane, status, o == 0);
      if (S.dev) sink_put(S, region, 0u, first + (uint32_t)lane, first + lane < tn, status == DCRX_S_OK, r, tup, lane, nullptr);      // tuple sink: the entry's item
      if (__builtin_expect(status == TAIL2_SLOW, 0)) v2_note_left(s_left, first + (uint32_t)lane, B, queue, gqueue, qcap, queue_count, r, false);
    }
    const uint32_t n_left = min(s_left[0], (uint32_t)V2_LEFT_SLOTS);
    if (__builtin_expect(n_left != 0u, 0)) {
      const bool live = (uint32_t)lane < n_left;
      const uint32_t slot = live ? s_left[1 + lane] : 0u;
      uint32_t x[2 + NW];
      v2_get_rows<2 + NW>(tq, Q.tcap, slot, live, x);
      if (live) {
        // the entry's flag log: the V pair and (when one pair holds a J tag) the J pair, as the scan saw them
        const uint32_t dg = x[1], vp = dg & 0xFFu, jp = (dg >> 8) & 0xFFu, jc = (dg >> 16) & 3u;
        V2EntryWords<NW> e;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          uint32_t l = (vp >> 3) == (uint32_t)k ? (V2_F_VF << (4 * (vp & 7u))) : 0u;
          if (jc == 1u && (jp >> 3) == (uint32_t)k) l |= V2_F_JF << (4 * (jp & 7u));
          e.lg[k] = l; e.w[k] = x[2 + k];
        }
        if (!v2_left_push<NW>(Q.left, queue_count, x[0] | (jc == 2u ? V2_R_JMULTI : 0u), e.lg, e.w)) v2_hand_over(B, queue, gqueue, qcap, queue_count, B.n_reads < (1ull << 30), x[0], false);
      }
      if (lane == 0) s_left[0] = 0u;
    }
  }
}

// The lean rescue: the scan kernel's event lists E and C in straight-line code (rescue2_fast compiled for the list's shape),
// `split` waves per region and list, entries read one batch ahead; jobs (list, region, part), the lists one after the other,
// so that the waves in flight at one time run the same code.  Leftovers as in the lean tail.
template <bool UNIFORM_LEN, int NW, int ORI>
__device__ __forceinline__ void v2_rescue_jobs(const Rescue2Tabs &rt, const V2FinishLds &L, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *__restrict__ records,
                                               const V2Lists &Q, const uint32_t n_regions, const uint32_t split_ec, uint32_t *__restrict__ queue,
                                               uint32_t *__restrict__ gqueue, const uint32_t qcap, uint32_t *__restrict__ queue_count,
                                               const DevTables *__restrict__ Tmem, const uint32_t gwave, const uint32_t n_gwaves, const int tid, const V2SinkCall &S,
                                               uint32_t *__restrict__ tickets) {
  constexpr int o = ORI;
  const Counters C{L.counts}, Cdry{L.dry};
  const LdsWords lw{dcrx_ldsaddr_of(L.strip)};
  uint32_t *strip = L.strip, *lds_counts = L.counts;
  uint32_t *s_left = L.left;
  const int lane = tid & 63;
  // (`split_ec`: the waves that share a region's list E in its low byte, list C's in the next)
  const uint32_t split_e = split_ec & 255u, split_c = (split_ec >> 8) ? (split_ec >> 8) : split_e;
  for (uint32_t job = gwave; job < n_regions * (split_e + split_c) && !(cfg.flags & DCRX_F_PROFILE_NO_EVENTS); job += n_gwaves) {
    const int which = job < n_regions * split_e ? V2_L_E : V2_L_C;
    const uint32_t split = which == V2_L_E ? split_e : split_c, jrel = which == V2_L_E ? job : job - n_regions * split_e;
    const uint32_t region = jrel / split, part = jrel % split;
    const V2ListRef l = v2_list<NW>(Q, which, region);
    const uint32_t en = min(Q.counts[V2_L_COUNTS * region + which], l.cap);
    // The `split` waves of a region and list take its batches of 64 as they come: a wave's first batch is its own (`part`), every
    // further one a ticket from the region's counter (tickets only grow: a wave that draws one beyond the list's end holds none
    // inside it).  Dealt out in fixed strides, 61 batches over 16 waves made 4 for thirteen of them and 3 for the rest, and the
    // launch took what the 4 take.  Two tickets are in flight per wave: the batch after next is drawn before this one is worked
    // on, so that its entries can be requested a batch ahead without a wait for the atomic.
    auto draw = [&]() -> uint32_t {
      uint32_t t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(tickets + V2_L_COUNTS * region + V2_L_TICKETS, which == V2_L_E ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      return split + (which == V2_L_E ? (t & 0xFFFFu) : (t >> 16));
    };
    constexpr bool AHEAD = NW <= 10;       // long reads: no look-ahead (the registers do not hold two entries, and a spill reload waits for the loads in flight)
    uint32_t x1[2 + 2 * NW];
    uint32_t first = 64 * part, next_first = 64 * part + 64 * en + 64;      // (past the end until a ticket says otherwise)
    if (first < en) next_first = 64 * draw();
    if constexpr (AHEAD) v2_get_rows<2 + 2 * NW>(l.rows, l.cap, first + lane, first + lane < en, x1);
    for (; first < en; ) {
      uint32_t x[2 + 2 * NW];
      // (the ticket of the batch after next: the atomic is in flight during this batch, its result read behind it)
      const bool more = next_first < en;
      uint32_t after_raw = 0;
      uint32_t nf = next_first;
      if constexpr (AHEAD) {      // (the batch in hand lands first: the ticket's atomic below is then in flight beside the next batch's loads, not waited for here)
#pragma unroll
        for (int k = 0; k < 2 + 2 * NW; k++) x[k] = x1[k];
        nf = v2_landed(x, next_first);
      }
      if (more && lane == 0) after_raw = __hip_atomic_fetch_add(tickets + V2_L_COUNTS * region + V2_L_TICKETS, which == V2_L_E ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (AHEAD) {
        v2_get_rows<2 + 2 * NW>(l.rows, l.cap, nf + lane, nf + lane < en, x1);     // the next batch, in flight during this one
      } else {
        v2_get_rows<2 + 2 * NW>(l.rows, l.cap, first + lane, first + lane < en, x);
      }
      uint32_t lg[NW], w[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) { lg[k] = x[1 + k]; w[k] = x[1 + NW + k]; }
      int status = -2;
      uint32_t errs = 0;
      uint64_t tup = 0;
      const uint32_t r = x[0] & V2_R_MASK;
      if (first + lane < en) {
        status = RESCUE2_SLOW;
        if (!(x[0] & V2_R_EXC)) {
          const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
#pragma unroll
          for (int k = 0; k < NW; k++) strip[k] = w[k];
          // a decombined read's record is stored where its fields are known (and its tuple made there: j_end of a J found through its
          // first half is the half's start + 2 * split, decombine.py:450-454: errs bit 3); every other settled read's below
          auto on_ok = [&](dcrx_record_t rec, const uint32_t e) {
            rec.frame = (uint8_t)(o ? 0 : 1);
            DCRX_STORE_FINISH(records + r, rec);
            if (S.dev) tup = sink_tuple_lean(rec, S.wpack, (e & 8u) != 0u && 2 * rt.split[1] != (int)rt.t.L[1]);
          };
          if (which == V2_L_E) status = rescue2_fast_to<ORI == 1, NW, V2_SHAPE_ONE>(rt, lw, lg, n, cfg, on_ok, errs, *Tmem, C, Cdry, x[1 + 2 * NW]);
          else status = rescue2_fast_to<ORI == 1, NW, V2_SHAPE_BOTH>(rt, lw, lg, n, cfg, on_ok, errs, *Tmem, C, Cdry, x[1 + 2 * NW]);
          if (status > 0) {      // settled, not decombined: the status alone
            dcrx_record_t rec;
            rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0; rec.vdel = rec.jdel = 0;
            rec.status = (uint8_t)status; rec.frame = (uint8_t)(o ? 0 : 1);
            DCRX_STORE_FINISH(records + r, rec);
          } else if (status < 0) errs = 0;
        }
      }
      v2_tally_rescue(lds_counts, lane, status, errs, o == 0);
      if (S.dev) sink_put(S, region, which == V2_L_E ? S.e_off : S.c_off, first + (uint32_t)lane, first + lane < en, status == DCRX_S_OK, r, tup, lane, nullptr);      // tuple sink
      if (__builtin_expect(status == RESCUE2_SLOW, 0)) v2_note_left(s_left, first + (uint32_t)lane, B, queue, gqueue, qcap, queue_count, r, (x[0] & V2_R_EXC) != 0u);
      first = next_first;
      if (more) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)after_raw);
        next_first = 64u * (split + (which == V2_L_E ? (t & 0xFFFFu) : (t >> 16)));
      }
    }
    const uint32_t n_left = min(s_left[0], (uint32_t)V2_LEFT_SLOTS);      // (as in the lean tail)
    if (__builtin_expect(n_left != 0u, 0)) {
      const bool live = (uint32_t)lane < n_left;
      const uint32_t slot = live ? s_left[1 + lane] : 0u;
      uint32_t x[1 + 2 * NW];
      v2_get_rows<1 + 2 * NW>(l.rows, l.cap, slot, live, x);
      if (live) {
        V2EntryWords<NW> e;
#pragma unroll
        for (int k = 0; k < NW; k++) { e.lg[k] = x[1 + k]; e.w[k] = x[1 + NW + k]; }
        if (!v2_left_push<NW>(Q.left, queue_count, x[0], e.lg, e.w)) v2_hand_over(B, queue, gqueue, qcap, queue_count, B.n_reads < (1ull << 30), x[0] & V2_R_MASK, (x[0] & V2_R_EXC) != 0u);
      }
      if (lane == 0) s_left[0] = 0u;
    }
  }
}

// The general form (dcr_frame3) as a role of the finishing launch, a function of its own like the lean roles:
//   mode bit 0  list X (reads with exception bytes whose flag log is not empty, flags on an odd read's last half pair: a few
//               thousand of a 10 M-read batch).  `width` lanes of a wave take entries — the general form costs a wave the longest
//               of its lanes' loops, and a short list is better spread over many waves than packed into a few —, `bsplit` blocks
//               share a region; block `vblock` of n_regions * bsplit;
//   mode bit 1  then the launch's left list, as it fills, by one wave (block 0 of the launch).
template <bool UNIFORM_LEN, int NW, int ORI, bool SINK>
DCRX_V2_ROLE void v2_general_role(const uint32_t mode_, const uint32_t vblock_, const uint32_t ka_lo, const uint32_t ka_hi);

// ONE launch for everything behind the scan, on the caller's stream — no side streams, no fork and no join (each wait of one
// queue for another cost the waiting queue 8-10 us).  A block's role follows from its index: first the blocks of list X
// (single reads with long dependent chains: they start at once and run under everything else), then blocks of the lean
// rescue and of the lean tail in turn, so that every compute unit holds waves of both at any time — the rescue is bound by
// instruction issue, the tail by its stream of entries.
// The lean roles are functions of their own (not inlined): each keeps the register allocation it has as a kernel — inlined side
// by side, the scalars of both stayed live across the role switch and the launch spilled twice as many as either kernel (311
// against 156; 287 us for what the two kernels did in 185).  A role reads the launch's arguments where the kernel found
// them, in the kernarg segment (its address is the one thing handed down, made scalar again by two v_readfirstlane):
// uniform scalar loads, nothing else travels through vector registers or the stack.
struct V2Roles { uint32_t xgrid, rgrid, tgrid, rsplit, tsplit, bsplit, width; };
struct V2FinishArgs {
  DevTables T0; BatchDev B; CfgDev cfg; dcrx_record_t *records; unsigned long long *counters; V2Lists Q; uint32_t n_regions; V2Roles R;
  uint32_t *queue, *gqueue; uint32_t qcap; uint32_t *queue_count; const DevTables *Tmem; V2SinkCall S;
};
// (every field through a pointer typed for the constant address space, dword by dword: scalar loads by construction — a
// reference into the segment through a generic pointer left half of the accesses as vector loads inside the hot loops,
// where their waits also drained the entry loads in flight: the tail role took 162 us for the kernel's 93)
typedef const __attribute__((address_space(4))) uint32_t *dcrx_kwords;
template <class T>
__device__ __forceinline__ T v2_karg(const dcrx_kwords base, const size_t offset) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  T out;
  uint32_t *d = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; i++) d[i] = base[offset / 4 + i];
  return out;
}
template <class T>
__device__ __forceinline__ T *v2_global(T *p) {      // (through the integer: a cast there and back between pointer types folds away)
  return (T *)(__attribute__((address_space(1))) T *)(uintptr_t)p;
}
// ... and what no kernel of the launch writes — the tables in device memory, the lists' counts the scan left — is constant
// memory to a role: uniform loads of it are scalar loads (a function, unlike a kernel, cannot show that nothing clobbers global
// memory before its loads, and would fetch the tables' fields of the rare paths with vector loads in the middle of the loop,
// where the wait for them also drains the entry loads in flight)
template <class T>
__device__ __forceinline__ T *v2_constant(T *p) { return (T *)(__attribute__((address_space(4))) T *)(uintptr_t)p; }
struct V2FinishLocals {      // a role's copy of the launch's arguments (what it does not use is never loaded)
  DevTables T0; BatchDev B; CfgDev cfg; dcrx_record_t *records; V2Lists Q; uint32_t n_regions; V2Roles R;
  uint32_t *queue, *gqueue; uint32_t qcap; uint32_t *queue_count; const DevTables *Tmem; V2SinkCall S;
  uint32_t *tickets;      // the lists' counts as plain global memory: the lean rescue draws its batch tickets there (V2_L_TICKETS)
};
__device__ __forceinline__ V2FinishLocals v2_finish_args(const uint32_t lo, const uint32_t hi) {
  uint32_t slo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo), shi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
  asm volatile("" : "+s"(slo), "+s"(shi));      // (inlined roles: each reads the arguments for itself, none kept live across the role switch)
  const uint64_t p = (uint64_t)slo | ((uint64_t)shi << 32);
  const dcrx_kwords k = reinterpret_cast<dcrx_kwords>(p);
  V2FinishLocals A;
#define DCRX_KARG(F) A.F = v2_karg<decltype(A.F)>(k, offsetof(V2FinishArgs, F))
  DCRX_KARG(T0); DCRX_KARG(B); DCRX_KARG(cfg); DCRX_KARG(records); DCRX_KARG(Q); DCRX_KARG(R);
  DCRX_KARG(queue); DCRX_KARG(gqueue); DCRX_KARG(queue_count); DCRX_KARG(Tmem); DCRX_KARG(S);
#undef DCRX_KARG
  A.n_regions = k[offsetof(V2FinishArgs, n_regions) / 4];
  A.qcap = k[offsetof(V2FinishArgs, qcap) / 4];
  // (a kernel knows that its pointer arguments are global memory; a pointer put together from two dwords is a generic one, and
  // every access through it a flat one, until it is told)
#define DCRX_GLOBAL(P) P = v2_global(P)
  DCRX_GLOBAL(A.records); DCRX_GLOBAL(A.Q.tail); DCRX_GLOBAL(A.Q.ev); DCRX_GLOBAL(A.Q.sx); DCRX_GLOBAL(A.queue);
  DCRX_GLOBAL(A.gqueue); DCRX_GLOBAL(A.queue_count); DCRX_GLOBAL(A.B.packed); DCRX_GLOBAL(A.B.lens); DCRX_GLOBAL(A.B.exc_read);
  DCRX_GLOBAL(A.B.exc_pos); DCRX_GLOBAL(A.B.exc_chr); DCRX_GLOBAL(A.B.exc_flag); DCRX_GLOBAL(A.T0.image);
#undef DCRX_GLOBAL
  A.tickets = v2_global(A.Q.counts);
  A.Tmem = v2_constant(A.Tmem); A.Q.counts = v2_constant(A.Q.counts); A.T0.kw_base = v2_constant(A.T0.kw_base);
  A.S.dev = v2_constant(A.S.dev);      // (the sink's descriptor: nothing of a launch writes it)
  A.S.items = v2_global(A.S.items); A.S.hits = v2_global(A.S.hits);
  return A;
}
template <bool UNIFORM_LEN, int NW, int ORI, bool SINK>
DCRX_V2_ROLE void v2_rescue_role(const uint32_t vblock_, const uint32_t ka_lo, const uint32_t ka_hi) {
  extern __shared__ __align__(64) uint32_t smem[];
  V2FinishLocals A = v2_finish_args(ka_lo, ka_hi);
  if constexpr (!SINK) A.S.dev = nullptr;
  const uint32_t vblock = (uint32_t)__builtin_amdgcn_readfirstlane((int)vblock_);
  const int tid = threadIdx.x;
  const V2Ori V = A.T0.v2[ORI];
  const V2FinishLds L = v2_finish_layout<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
  uint32_t kw_base[K_NCLASS];
#pragma unroll
  for (int c = 0; c < K_NCLASS; c++) kw_base[c] = A.T0.kw_base[c];
  const Rescue2Tabs rt = rescue2_tabs(A.T0, V, reinterpret_cast<const uint8_t *>(L.side), reinterpret_cast<const uint8_t *>(L.bk), ORI == 1, kw_base);
  constexpr uint32_t WPB = DCRX_V2_FBLOCK / 64;
  v2_rescue_jobs<UNIFORM_LEN, NW, ORI>(rt, L, A.B, A.cfg, A.records, A.Q, A.n_regions, A.R.rsplit, A.queue, A.gqueue, A.qcap, A.queue_count, A.Tmem,
                                       vblock * WPB + (uint32_t)(tid >> 6), A.R.rgrid * WPB, tid, A.S, A.tickets);
}
template <bool UNIFORM_LEN, int NW, int ORI, bool SINK>
DCRX_V2_ROLE void v2_tail_role(const uint32_t vblock_, const uint32_t ka_lo, const uint32_t ka_hi) {
  extern __shared__ __align__(64) uint32_t smem[];
  V2FinishLocals A = v2_finish_args(ka_lo, ka_hi);
  if constexpr (!SINK) A.S.dev = nullptr;
  const uint32_t vblock = (uint32_t)__builtin_amdgcn_readfirstlane((int)vblock_);
  const int tid = threadIdx.x;
  const V2Ori V = A.T0.v2[ORI];
  const V2FinishLds L = v2_finish_layout<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
  const Tail2Tabs tt = tail2_tabs(A.T0, V, reinterpret_cast<const uint8_t *>(L.side), reinterpret_cast<const uint8_t *>(L.bk), ORI == 1);
  constexpr uint32_t WPB = DCRX_V2_FBLOCK / 64;
  v2_tail_jobs<UNIFORM_LEN, NW, ORI>(tt, L, A.B, A.cfg, A.records, A.Q, A.n_regions, A.R.tsplit, A.queue, A.gqueue, A.qcap, A.queue_count, A.Tmem,
                                     vblock * WPB + (uint32_t)(tid >> 6), A.R.tgrid * WPB, tid, A.S);
}
template <bool UNIFORM_LEN, int NW, int ORI, bool SINK>
DCRX_V2_ROLE void v2_general_role(const uint32_t mode_, const uint32_t vblock_, const uint32_t ka_lo, const uint32_t ka_hi) {
  extern __shared__ __align__(64) uint32_t smem[];
  V2FinishLocals A = v2_finish_args(ka_lo, ka_hi);
  if constexpr (!SINK) A.S.dev = nullptr;
  const uint32_t vblock = (uint32_t)__builtin_amdgcn_readfirstlane((int)vblock_), mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)mode_);
  const int tid = threadIdx.x, lane = tid & 63;
  V2Ori V = A.T0.v2[ORI];
  const V2FinishLds L = v2_finish_layout<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
  const DevTables T = tables_in_lds(A.T0, reinterpret_cast<const uint8_t *>(L.side), A.T0.dfa_bytes, false);      // (the side tables and buckets the block staged)
  V.bk = reinterpret_cast<const uint8_t *>(L.bk);
  const Counters C{L.counts};
  const bool tagged = A.B.n_reads < (1ull << 30);
  uint32_t width = A.R.width;
  // one loop for both lists (one copy of the general form's code): entries of list X by (region, slot), `width` lanes of a wave
  // at a time; then — mode bit 1, one wave — the left list as its entries arrive, a lane per entry, in order; every other block
  // signs off in queue_count[V2_QC_DONE] when its role is done (its pushes before that)
  const uint32_t region = vblock / A.R.bsplit, bpart = vblock % A.R.bsplit;
  const bool do_x = (mode & 1u) && region < A.n_regions && !(A.cfg.flags & DCRX_F_PROFILE_NO_EVENTS);
  const V2ListRef lx = v2_list<NW>(A.Q, V2_L_X, do_x ? region : 0u);
  const uint32_t x_total = do_x ? min(A.Q.counts[V2_L_COUNTS * region + V2_L_X], lx.cap) : 0u;
  // (a long list — a run with many N tails — fills its waves: few lanes per wave pay only while the list is a handful per region)
  if (x_total > 64u * A.R.bsplit * (DCRX_V2_FBLOCK / 64)) width = 64u;
  else if (x_total > 16u * A.R.bsplit * (DCRX_V2_FBLOCK / 64)) width = 16u;
  uint32_t first = width * ((uint32_t)(tid >> 6) + (DCRX_V2_FBLOCK / 64) * bpart);
  const uint32_t step = width * (DCRX_V2_FBLOCK / 64) * A.R.bsplit;
  uint4 *lrows = A.Q.left;
  uint32_t *valid = v2_left_valid<NW>(lrows);
  const uint32_t others = A.R.xgrid + A.R.rgrid + A.R.tgrid - 1u;
  bool polling = false;
  uint32_t consumed = 0, spins = 0;
  for (;;) {
    const uint4 *rows;
    uint32_t i;
    bool live;
    if (!polling) {
      if (first >= x_total) {
        if (!(mode & 2u) || tid >= 64) break;
        polling = true;
        continue;
      }
      rows = lx.rows; i = first + (uint32_t)lane; live = (uint32_t)lane < width && i < x_total;
      first += step;
    } else {
      const uint32_t done = __hip_atomic_load(A.queue_count + V2_QC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t n = min(__hip_atomic_load(A.queue_count + V2_QC_LEFT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), V2_LEFT_CAP);
      // the valid entries in front (an entry becomes valid a moment after its slot was drawn)
      i = consumed + (uint32_t)lane;
      const bool ok = i < n && __hip_atomic_load(valid + (i < V2_LEFT_CAP ? i : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      const unsigned long long m = __ballot(ok);
      const uint32_t run = ~m ? (uint32_t)__builtin_ctzll(~m) : 64u;   // lanes 0 .. run-1 hold valid entries
      if (!run) {
        if (done >= others && consumed >= n) break;                    // every pusher has signed off (its pushes before that) and nothing is left
        if (++spins > (1u << 22)) break;                               // (never seen: a block that does not sign off; the records it owed keep status 255)
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");               // (this unit's cache may hold older bytes of the entries' lines)
      rows = lrows; live = (uint32_t)lane < run;
      consumed += run;
    }
    uint32_t x[1 + 2 * NW];
    v2_get_rows<1 + 2 * NW>(rows, 0u, i, live, x);
    uint32_t lg[NW], w[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { lg[k] = x[1 + k]; w[k] = x[1 + NW + k]; }
    if (live) {
      v2_general_entry<UNIFORM_LEN, NW, ORI>(T, V, A.B, A.cfg, x[0], lg, w, C, A.records, A.queue, A.gqueue, A.qcap, A.queue_count, tagged);
      if (A.S.dev) sink_late(A.S, A.records, x[0] & V2_R_MASK);      // tuple sink: the read's tuple, when this form decombined it
      if (polling) valid[i] = 0u;                                      // (re-armed for the next launch)
    }
  }
  if (polling && lane == 0) { A.queue_count[V2_QC_LEFT] = 0u; A.queue_count[V2_QC_DONE] = 0u; }
}

template <bool UNIFORM_LEN, int NW, int ORI, bool TAIL_ROLE = true, bool SINK = false>
__global__ __launch_bounds__(DCRX_V2_FBLOCK, DCRX_V2_RWAVES) void finish2_kernel(const V2FinishArgs A) {
  extern __shared__ __align__(64) uint32_t smem[];
  const int tid = threadIdx.x;
  const V2Roles R = A.R;
  const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();      // (where this launch's V2FinishArgs lies)
  // the role of this block
  uint32_t b = blockIdx.x;
  int role;                 // 0 list X, 1 rescue, 2 tail
  bool work = true;
  if (b < R.xgrid) {
    role = 0;
    // (most blocks of a short list's pass find their share empty: they stage nothing)
    const uint32_t g = b / R.bsplit;
    work = g < A.n_regions && R.width * (DCRX_V2_FBLOCK / 64) * (b % R.bsplit) < A.Q.counts[V2_L_COUNTS * g + V2_L_X];
  } else {
    b -= R.xgrid;
    const uint32_t paired = 2u * min(R.rgrid, R.tgrid);
    if (b < paired) { role = (b & 1u) ? 2 : 1; b >>= 1; }
    else { b -= paired; role = R.rgrid > R.tgrid ? 1 : 2; b += min(R.rgrid, R.tgrid); }
  }
  const V2Ori V = A.T0.v2[ORI];
  V2FinishLds L = v2_finish_layout<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
  // block 0 — of list X's role — stays to take the left list as it fills and leaves last; every other block signs off when its
  // role is done (its left-list pushes are in memory, each behind a fence of its own)
  const bool staged = work || blockIdx.x == 0;
  if (staged) {
    L = v2_finish_stage<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
    __syncthreads();
    if (role == 1) v2_rescue_role<UNIFORM_LEN, NW, ORI, SINK>(b, (uint32_t)ka, (uint32_t)(ka >> 32));
    else if (TAIL_ROLE && role == 2) v2_tail_role<UNIFORM_LEN, NW, ORI, SINK>(b, (uint32_t)ka, (uint32_t)(ka >> 32));      // (TAIL_ROLE false: the scan kernel has taken the tail, the launch holds no such block)
    else v2_general_role<UNIFORM_LEN, NW, ORI, SINK>(blockIdx.x == 0 ? 3u : 1u, b, (uint32_t)ka, (uint32_t)(ka >> 32));
  }
  __syncthreads();
  if (blockIdx.x != 0 && tid == 0) atomicAdd(A.queue_count + V2_QC_DONE, 1u);
  __syncthreads();
  if (staged && tid < DCRX_N_COUNTERS && L.counts[tid]) atomicAdd(&A.counters[tid], (unsigned long long)L.counts[tid]);
}

// (the A/B forms below launch the roles as kernels of their own: their left list is taken by one block of this kernel at the end)
template <bool UNIFORM_LEN, int NW, int ORI>
__global__ __launch_bounds__(DCRX_V2_FBLOCK, DCRX_V2_RWAVES) void left2_kernel(const V2FinishArgs A) {
  extern __shared__ __align__(64) uint32_t smem[];
  const int tid = threadIdx.x;
  if (A.queue_count[V2_QC_LEFT] == 0u) return;
  if (tid == 0) A.queue_count[V2_QC_DONE] = A.R.xgrid + A.R.rgrid + A.R.tgrid;      // (every pusher ended with its kernel)
  __syncthreads();
  const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  const V2Ori V = A.T0.v2[ORI];
  const V2FinishLds L = v2_finish_stage<NW, DCRX_V2_FBLOCK>(A.T0, V, smem, tid);
  __syncthreads();
  v2_general_role<UNIFORM_LEN, NW, ORI, false>(2u, 0u, (uint32_t)ka, (uint32_t)(ka >> 32));
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && L.counts[tid]) atomicAdd(&A.counters[tid], (unsigned long long)L.counts[tid]);
}

// The roles as launches of their own (A/B: DCRX_F_V2_SIDE_STREAMS, the tail kernel beside the rescue kernel on a side stream of
// the handle; DCRX_F_V2_LEAN_SERIAL, one after the other on the caller's stream; tests run all three forms).
template <bool UNIFORM_LEN, int NW, int ORI>
__global__ __launch_bounds__(DCRX_V2_TBLOCK, DCRX_V2_TWAVES) void tail2_kernel(
    DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records, unsigned long long *__restrict__ counters,
    V2Lists Q, uint32_t n_regions, uint32_t split, uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue, uint32_t qcap,
    uint32_t *__restrict__ queue_count, const DevTables *__restrict__ Tmem) {
  extern __shared__ __align__(64) uint32_t smem[];
  const int tid = threadIdx.x;
  const V2Ori V = T0.v2[ORI];
  const V2FinishLds L = v2_finish_stage<NW, DCRX_V2_TBLOCK>(T0, V, smem, tid);
  const Tail2Tabs tt = tail2_tabs(T0, V, reinterpret_cast<const uint8_t *>(L.side), reinterpret_cast<const uint8_t *>(L.bk), ORI == 1);
  __syncthreads();
  v2_tail_jobs<UNIFORM_LEN, NW, ORI>(tt, L, B, cfg, records, Q, n_regions, split, queue, gqueue, qcap, queue_count, Tmem,
                                     blockIdx.x * (DCRX_V2_TBLOCK / 64) + (uint32_t)(tid >> 6), gridDim.x * (DCRX_V2_TBLOCK / 64), tid, V2SinkCall{});      // (A/B form: no tuple sink)
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && L.counts[tid]) atomicAdd(&counters[tid], (unsigned long long)L.counts[tid]);
}

template <bool UNIFORM_LEN, int NW, int ORI>
__global__ __launch_bounds__(DCRX_V2_FBLOCK, DCRX_V2_RWAVES) void rescue2_kernel(
    DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records, unsigned long long *__restrict__ counters,
    V2Lists Q, uint32_t n_regions, uint32_t split, uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue, uint32_t qcap,
    uint32_t *__restrict__ queue_count, const DevTables *__restrict__ Tmem) {
  extern __shared__ __align__(64) uint32_t smem[];
  const int tid = threadIdx.x;
  const V2Ori V = T0.v2[ORI];
  const V2FinishLds L = v2_finish_stage<NW, DCRX_V2_FBLOCK>(T0, V, smem, tid);
  uint32_t kw_base[K_NCLASS];
#pragma unroll
  for (int c = 0; c < K_NCLASS; c++) kw_base[c] = T0.kw_base[c];
  const Rescue2Tabs rt = rescue2_tabs(T0, V, reinterpret_cast<const uint8_t *>(L.side), reinterpret_cast<const uint8_t *>(L.bk), ORI == 1, kw_base);
  __syncthreads();
  v2_rescue_jobs<UNIFORM_LEN, NW, ORI>(rt, L, B, cfg, records, Q, n_regions, split, queue, gqueue, qcap, queue_count, Tmem,
                                       blockIdx.x * (DCRX_V2_FBLOCK / 64) + (uint32_t)(tid >> 6), gridDim.x * (DCRX_V2_FBLOCK / 64), tid, V2SinkCall{}, Q.counts);
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && L.counts[tid]) atomicAdd(&counters[tid], (unsigned long long)L.counts[tid]);
}

// The event kernel: the general form (dcr_frame3) on reads held in registers.  Reads with exception bytes
// are resolved against their slice of the exception list.  A block takes `group` consecutive regions at a
// time and its waves the batches of 64 of their entries taken together (lane -> region and slot through the
// prefix sums of the regions' counts): the slow list holds a handful of entries per region, and the
// general form costs a wave the same whether 5 or 64 of its lanes are at work.
template <bool UNIFORM_LEN, int NW, int ORI>
__global__ __launch_bounds__(DCRX_V2_FBLOCK, 4) void events2_kernel(
    DevTables T0, BatchDev B, CfgDev cfg, dcrx_record_t *__restrict__ records, unsigned long long *__restrict__ counters,
    V2Lists Q, int which, uint32_t group, uint32_t width, uint32_t bsplit, uint32_t ext, uint32_t n_regions, uint32_t *__restrict__ queue, uint32_t *__restrict__ gqueue, uint32_t qcap,
    uint32_t *__restrict__ queue_count) {
  extern __shared__ __align__(64) uint32_t smem[];
  V2Ori V = T0.v2[ORI];
  // which: the list (V2_L_E .. V2_L_X)
  const uint32_t lcap = v2_list<NW>(Q, which, 0).cap;
  // ext: the packed germline regions are staged behind the side tables (the launcher found room for them)
  const uint32_t side_bytes = (ext ? T0.lds_image2_bytes : T0.lds_image_bytes) - T0.dfa_bytes;
  uint32_t *lds_counts = smem;
  uint32_t *lds_side = smem + DCRX_N_COUNTERS;
  uint32_t *lds_bk = lds_side + side_bytes / 4;
  const int tid = threadIdx.x;
  // a short list's pass (one region to a group, the grid covers every region): a block whose share of its region's list
  // is empty leaves before it stages anything — most blocks of such a pass
  if (group == 1u && gridDim.x / bsplit >= n_regions) {
    const uint32_t g = blockIdx.x / bsplit;
    if (g >= n_regions || width * (DCRX_V2_FBLOCK / 64) * (blockIdx.x % bsplit) >= min(Q.counts[V2_L_COUNTS * g + which], lcap)) return;
  }
  if (tid < DCRX_N_COUNTERS) lds_counts[tid] = 0;
  stage_lds<DCRX_V2_FBLOCK>(T0.image + T0.dfa_bytes, lds_side, side_bytes / 16, 0, 0, tid);
  stage_lds<DCRX_V2_FBLOCK>(V.bk, lds_bk, V.bk_bytes / 16, 0, 0, tid);
  const DevTables T = tables_in_lds(T0, reinterpret_cast<const uint8_t *>(lds_side), T0.dfa_bytes, ext != 0u);
  V.bk = reinterpret_cast<const uint8_t *>(lds_bk);
  __syncthreads();
  const Counters C{lds_counts};
  const int lane = tid & 63;
  const bool tagged = B.n_reads < (1ull << 30);
  __shared__ uint32_t pref[DCRX_V2_GROUP_MAX + 1];
  // `bsplit` blocks share a group of regions (a short list spread over many waves): block b of them takes the rounds b, b + bsplit, ...
  const uint32_t bpart = blockIdx.x % bsplit, bsteps = width * (DCRX_V2_FBLOCK / 64) * bsplit;
  for (uint32_t g0 = (blockIdx.x / bsplit) * group; g0 < n_regions; g0 += (gridDim.x / bsplit) * group) {
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (uint32_t k = 0; k < group; k++) {
        pref[k] = acc;
        if (g0 + k < n_regions) acc += min(Q.counts[V2_L_COUNTS * (g0 + k) + which], lcap);     // (appends may have run past a region's end)
      }
      pref[group] = acc;
    }
    __syncthreads();
    const uint32_t total = pref[group];
    // (no look-ahead here: this kernel needs its registers, and a spilled one would make every reload wait for
    // the loads in flight; the other waves of the CU cover the entry loads)
    // `width` lanes of a wave take entries (64, or fewer for the slow list: the general form costs a wave the longest
    // of its lanes' loops, and a short list is better spread over many waves than packed into a few)
    for (uint32_t first = width * ((uint32_t)(tid >> 6) + (DCRX_V2_FBLOCK / 64) * bpart); first < total && !(cfg.flags & DCRX_F_PROFILE_NO_EVENTS); first += bsteps) {
      const uint32_t i = first + lane;
      const bool live = (uint32_t)lane < width && i < total;
      uint32_t g = 0;
      for (uint32_t k = 1; k < group; k++) g += i >= pref[k] ? 1u : 0u;
      const uint32_t slot = i - pref[g];
      const V2ListRef l = v2_list<NW>(Q, which, (size_t)(g0 + g));
      uint32_t x[1 + 2 * NW];
      v2_get_rows<1 + 2 * NW>(l.rows, l.cap, slot, live, x);
      uint32_t lg[NW], w[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) { lg[k] = x[1 + k]; w[k] = x[1 + NW + k]; }
      if (live) v2_general_entry<UNIFORM_LEN, NW, ORI>(T, V, B, cfg, x[0], lg, w, C, records, queue, gqueue, qcap, queue_count, tagged);
    }
  }
  __syncthreads();
  if (tid < DCRX_N_COUNTERS && lds_counts[tid]) atomicAdd(&counters[tid], (unsigned long long)lds_counts[tid]);
}

// ---- the tuple sink's last step (dcrx_sink_device.h) ------------------------------------------------
// One block per region of the scan, behind the list kernel: the region's bitmap of decombined reads is built in LDS from the
// items the kernels left, the items are ranked by it, and the message (include/dcrx.h, dcrx_tuple_message_bytes: bitmap |
// low words | high bytes, in read order) gets the region's words and tuples — at the offset the regions in front of it fill,
// known from their counts of decombined reads.  The block that reads the counts last re-arms them.
constexpr int V2_PLACE_BLOCK = 1024;
constexpr int V2_PLACE_KEEP = 24;      // items a thread holds in registers between the passes (24 K items per region: a 10 M-read step has 18 K)
// sums over a block of V2_PLACE_BLOCK threads by wave shuffles (s16: 16 words of LDS; two barriers each)
__device__ __forceinline__ uint32_t v2_wave_sum(uint32_t x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d);
  return x;
}
__device__ __forceinline__ uint32_t v2_block_sum(uint32_t x, uint32_t *s16, const int tid) {
  x = v2_wave_sum(x);
  __syncthreads();
  if ((tid & 63) == 0) s16[tid >> 6] = x;
  __syncthreads();
  uint32_t t = 0;
#pragma unroll
  for (int w = 0; w < V2_PLACE_BLOCK / 64; w++) t += s16[w];
  return t;
}
// exclusive prefix of x over the block's threads
__device__ __forceinline__ uint32_t v2_block_exclusive(const uint32_t x, uint32_t *s16, const int tid) {
  uint32_t inc = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, d); if ((tid & 63) >= d) inc += v; }
  __syncthreads();
  if ((tid & 63) == 63) s16[tid >> 6] = inc;
  __syncthreads();
  uint32_t off = 0;
#pragma unroll
  for (int w = 0; w < V2_PLACE_BLOCK / 64; w++) off += w < (tid >> 6) ? s16[w] : 0u;
  return off + inc - x;
}
__global__ __launch_bounds__(V2_PLACE_BLOCK) void v2_place_kernel(const V2SinkCall S, const uint32_t *__restrict__ counts, const uint32_t n_regions,
                                                                  const uint32_t tcap, const uint32_t ecap, const uint32_t ccap, const uint32_t fused,
                                                                  const uint64_t n_reads, const uint64_t n_slots, uint8_t *__restrict__ msg,
                                                                  const uint32_t hi_bytes, uint64_t *__restrict__ d_total, const uint32_t hcap) {
  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ uint32_t s16[V2_PLACE_BLOCK / 64];
  __shared__ uint32_t s_last, s_mine;
  const V2SinkDev D = *S.dev;
  const int tid = threadIdx.x;
  const uint32_t region = blockIdx.x;
  const uint32_t wpr = S.per_block >> 5;                 // words of the region's bitmap (a region is a multiple of 512 reads)
  uint32_t *bm = smem, *pre = smem + wpr;
  // the region's tuples in read order, staged in LDS when they fit (hcap of them) and copied out in whole lines: a tuple
  // written straight to its place is a 4-byte and a 1-byte store to an address of its own — 64 lines per wave and store, 17 us
  // of a 21-us kernel
  uint32_t *stage_a = smem + 2 * wpr;
  uint8_t *stage_b = reinterpret_cast<uint8_t *>(stage_a + hcap);
  // the region's sections as one run of items: item f of the run lies at items[addr(f)]
  const uint32_t n_late = D.late[region];
  const uint32_t n_tail = fused ? counts[V2_L_COUNTS * region + V2_L_RING] : min(counts[V2_L_COUNTS * region + V2_L_TAIL], tcap);
  const uint32_t n_e = min(counts[V2_L_COUNTS * region + V2_L_E], ecap), n_c = min(counts[V2_L_COUNTS * region + V2_L_C], ccap);
  const uint32_t c0 = n_tail, c1 = c0 + n_e, c2 = c1 + n_c, n_items = c2 + min(n_late, S.late_cap);
  const uint64_t blk_lo = (uint64_t)region * S.per_block;
  const uint2 *items = S.items + (size_t)region * S.stride;
  auto addr = [&](const uint32_t f) -> uint32_t {
    return f < c0 ? f : (f < c1 ? S.e_off + (f - c0) : (f < c2 ? S.c_off + (f - c1) : S.late_off + (f - c2)));
  };
  // A thread's first V2_PLACE_KEEP items stay in registers between the two passes, their loads in flight together (one load at
  // a time, each behind the one before, cost the kernel 34 us for a 10 M-read step: eighteen round trips to the L2 per pass);
  // what a larger region holds beyond them is read twice.  The loads go first: the counts' sums below run under them.
  constexpr int KEEP = V2_PLACE_KEEP;
  uint2 it[KEEP];
#pragma unroll
  for (int k = 0; k < KEEP; k++) {
    const uint32_t f = (uint32_t)tid + (uint32_t)k * V2_PLACE_BLOCK;
    it[k] = make_uint2(0u, V2_SINK_EMPTY);
    if (f < n_items) it[k] = items[addr(f)];
  }
  // decombined reads of the regions in front, and of all
  uint32_t before = 0, all = 0;
  // (the region's own count is kept from this pass: the block with the last ticket zeroes the counts, and a block that looked
  // its count up again behind its ticket could find the zero)
  for (uint32_t q = (uint32_t)tid; q < n_regions; q += V2_PLACE_BLOCK) { const uint32_t h = S.hits[q]; all += h; if (q < region) before += h; if (q == region) s_mine = h; }
  for (uint32_t i = (uint32_t)tid; i < wpr; i += V2_PLACE_BLOCK) bm[i] = 0u;
  const uint64_t base = v2_block_sum(before, s16, tid);
  const uint64_t total = v2_block_sum(all, s16, tid);
  if (tid == 0) s_last = atomicAdd(D.ticket, 1u) == n_regions - 1u ? 1u : 0u;      // (this block has read every count)
  auto mark = [&](const uint32_t y) { if (y != V2_SINK_EMPTY) { const uint32_t k = y & 0xFFFFFFu; atomicOr(&bm[k >> 5], 1u << (k & 31u)); } };
#pragma unroll
  for (int k = 0; k < KEEP; k++) mark(it[k].y);
  for (uint32_t f = (uint32_t)tid + (uint32_t)KEEP * V2_PLACE_BLOCK; f < n_items; f += V2_PLACE_BLOCK) mark(items[addr(f)].y);
  __syncthreads();
  // exclusive prefix of the words' populations: a run of words per thread, then the runs
  const uint32_t per = (wpr + V2_PLACE_BLOCK - 1) / V2_PLACE_BLOCK;
  uint32_t run = 0;
  for (uint32_t i = (uint32_t)tid * per; i < min(((uint32_t)tid + 1u) * per, wpr); i++) run += (uint32_t)__popc(bm[i]);
  uint32_t acc = v2_block_exclusive(run, s16, tid);
  for (uint32_t i = (uint32_t)tid * per; i < min(((uint32_t)tid + 1u) * per, wpr); i++) { pre[i] = acc; acc += (uint32_t)__popc(bm[i]); }
  __syncthreads();
  // the tuples, in read order
  const uint64_t bm_bytes = ((n_slots + 63) / 64) * 8;
  uint32_t *plane_a = reinterpret_cast<uint32_t *>(msg + bm_bytes);
  uint8_t *plane_b = msg + bm_bytes + total * 4;
  const uint32_t mine = s_mine;      // (written before the sums' barriers)
  const bool staged = mine <= hcap;
  auto place = [&](const uint2 v) {
    if (v.y == V2_SINK_EMPTY) return;
    const uint32_t k = v.y & 0xFFFFFFu;
    const uint32_t rank = pre[k >> 5] + (uint32_t)__popc(bm[k >> 5] & ((1u << (k & 31u)) - 1u));
    if (staged) {
      if (rank < hcap) { stage_a[rank] = v.x; stage_b[rank] = (uint8_t)(v.y >> 24); }      // (rank < the region's count by construction)
    } else {
      plane_a[base + rank] = v.x;
      if (hi_bytes) plane_b[base + rank] = (uint8_t)(v.y >> 24);
    }
  };
#pragma unroll
  for (int k = 0; k < KEEP; k++) place(it[k]);
  for (uint32_t f = (uint32_t)tid + (uint32_t)KEEP * V2_PLACE_BLOCK; f < n_items; f += V2_PLACE_BLOCK) place(items[addr(f)]);
  if (staged) {
    __syncthreads();
    for (uint32_t i = (uint32_t)tid; i < mine; i += V2_PLACE_BLOCK) plane_a[base + i] = stage_a[i];
    if (hi_bytes)
      for (uint32_t i = (uint32_t)tid; i < mine; i += V2_PLACE_BLOCK) plane_b[base + i] = stage_b[i];
  }
  // the region's words of the bitmap (pairs of LDS words), and — the last region's block — the words behind the batch's reads
  uint64_t *out_bm = reinterpret_cast<uint64_t *>(msg);
  const uint64_t w_lo = blk_lo >> 6, w_all = bm_bytes / 8;
  for (uint32_t i = (uint32_t)tid; i < (wpr >> 1); i += V2_PLACE_BLOCK)
    if (w_lo + i < w_all) out_bm[w_lo + i] = (uint64_t)bm[2 * i] | ((uint64_t)bm[2 * i + 1] << 32);
  if (region == n_regions - 1u)
    for (uint64_t w = w_lo + (wpr >> 1) + (uint32_t)tid; w < w_all; w += V2_PLACE_BLOCK) out_bm[w] = 0ull;
  if (region == 0 && tid == 0) *d_total = total;
  if (tid == 0) D.late[region] = 0u;
  if (s_last) {      // every block has its offsets: the counts are zero again for the next call
    for (uint32_t q = (uint32_t)tid; q < n_regions; q += V2_PLACE_BLOCK) S.hits[q] = 0u;
    if (tid == 0) *D.ticket = 0u;
  }
}

// ---- launcher ------------------------------------------------------------------------------------
// LDS the scan kernel needs for frame o: the pair table and the block's counters
static uint32_t v2_scan_lds_bytes(const DevTables &T, int o) { return T.v2[o].trans_bytes + DCRX_N_COUNTERS * 4 + V2_WK_WORDS * 4; }

// The v2 kernels serve one frame; the A/B switches of the three-launch form, the forced slow
// reader and orientation `both` keep that form.
// The v2 kernels serve one frame per pass: `reverse` and `forward` are one pass, `both` (decombine.py:1005-1010) the reverse
// frame and then the forward frame for the reads it did not decombine (both frames' tables must fit).
bool v2_applies(const LaunchPlan &P, const DevTables &T, const CfgDev &cfg, const uint32_t stride) {
  if (!T.v2_ok || !P.v2_events || !P.v2_slow || !P.v2_acc || !P.v2_left) return false;      // (the tail list: only where a launch needs it, launch_v2)
  if (cfg.flags & (DCRX_F_V1_KERNELS | DCRX_F_FORCE_SLOW_READER | DCRX_F_ONE_BASE_SCAN | DCRX_F_LIST_RESCUE | DCRX_F_PROFILE_LIST_SCAN_ONLY)) return false;
  // (the lean kernels add a strip of LDS per lane: priced at the register shape the batch's stride takes — at the long reads'
  // shape whatever the batch, the extended alpha set's larger keyword tables sent its 150-nt batches to the three-launch form)
  auto fits = [&](int o) {
    const uint32_t fin = stride <= 40 ? v2_finish_block_lds<10>(T, o, DCRX_V2_FBLOCK)
                                      : (stride <= 4 * DCRX_NWMAX ? v2_finish_block_lds<DCRX_NWMAX>(T, o, DCRX_V2_FBLOCK) : v2_finish_block_lds<DCRX_V2_NWLONG>(T, o, DCRX_V2_FBLOCK));
    return v2_scan_lds_bytes(T, o) <= 160u * 1024u && fin <= 64u * 1024u;
  };
  if (cfg.orientation == DCRX_ORIENT_BOTH) return fits(0) && fits(1) && !(cfg.flags & DCRX_F_PROFILE_MASK);
  return fits(cfg.orientation == DCRX_ORIENT_FORWARD ? 0 : 1);
}

template <bool UNIFORM, int NW, int RPL, bool NARROW, bool PREFETCH = true>
static hipError_t launch_v2(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                            uint32_t *queue, uint32_t *gqueue, uint32_t qcap, uint32_t *queue_count,
                            unsigned long long *d_counters, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t retry, V2SinkLaunch *sink) {
  const int o = cfg.orientation == DCRX_ORIENT_FORWARD ? 0 : 1;
  // The fused form (the tail inside the scan kernel, through a ring in LDS: scan2_kernel, FUSE) for the 150-nt shape when the
  // frame's pair table leaves room for the side tables, the buckets and a ring of at least four batches; the A/B forms and the
  // profiling switches keep the tail a launch of its own.
  constexpr bool CAN_FUSE = NW == 10 && ((RPL == 2 && PREFETCH) || (RPL == 3 && !PREFETCH));
  uint32_t ring_batches = 0;
  // (measured, profiles/r04: config 2's 57 KB table 0.413 ms per step fused against 0.429 with the tail as a role; the extended
  // beta set's 76 KB table 0.752 against 0.726 for config 3's two chains: pair tables of up to 64 KB fuse)
  static const uint32_t fuse_limit = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_FUSE_LIMIT_KB"); const int v = e ? atoi(e) : 0; return v > 0 ? (uint32_t)v * 1024u : 64u * 1024u; }();      // (A/B)
  if (CAN_FUSE && T.v2[o].trans_bytes <= fuse_limit && !(cfg.flags & (DCRX_F_V2_NO_FUSE | DCRX_F_V2_SIDE_STREAMS | DCRX_F_V2_LEAN_SERIAL | DCRX_F_V2_NO_LEAN_RESCUE | (DCRX_F_PROFILE_MASK & ~DCRX_F_PROFILE_TAIL_STREAM_ONLY)))) {
    const uint32_t fixed = v2_scan_lds_bytes(T, o) + (T.lds_image_bytes - T.dfa_bytes) + T.v2[o].bk_bytes;
    static const uint32_t nb_max = [] {      // (tests: DCRX_DEBUG_RING_BATCHES=4 forces the shortest ring)
      const char *e = dcrx_debug_env("DCRX_DEBUG_RING_BATCHES");
      const uint32_t v = e ? (uint32_t)atoi(e) : V2_RING_MAXBATCHES;
      return (v == 4u || v == 8u || v == 16u) ? v : V2_RING_MAXBATCHES;
    }();
    // (the whole ring or none: with the extended alpha set's 119 KB table only four batches fit, the scanning waves wait for
    // room, and config 3 took 7.26 ms per 100 M reads against 5.91 with the tail as a role of the finishing launch;
    // tests force shorter rings through DCRX_DEBUG_RING_BATCHES)
    // (round 5: a ring of 8 batches serves config 2 as well as one of 16 — 0.366 / 0.373 against 0.374 / 0.377 ms per step —, one
    // of 4 does not: 0.447; profiles/r05/ring_batches_ab.log)
    const uint32_t nb_min = dcrx_debug_env("DCRX_DEBUG_RING_BATCHES") ? 4u : 8u;
    for (uint32_t nb = nb_max; nb >= nb_min; nb >>= 1)
      if (fixed + nb * 64u * V2_RING_STRIDE * 4u <= 160u * 1024u) { ring_batches = nb; break; }
  }
  auto ks = scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH>;
  if constexpr (CAN_FUSE) {
    if (ring_batches) ks = o ? scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1> : scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0>;
  }
  // FUSE_E: list E through an event ring inside the scan kernel as well (the two-reads-per-lane shape of uniform 150-nt batches,
  // one pass, no tuple sink, no A/B or profiling switch), where the block's LDS holds the event ring beside a tail ring of eight
  // batches or more; DCRX_DEBUG_FUSE_E=0 keeps list E a role of the finishing launch (A/B), DCRX_DEBUG_FUSE_E_WAVES the rescue waves
  constexpr bool CAN_FUSE_E = CAN_FUSE && UNIFORM && RPL == 2;
  auto ks_e = ks;
  bool fuse_e = false;
  uint32_t ring_batches_e = 0, scan_lds_e = 0;
  static const int fuse_e_env = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_FUSE_E"); return e ? atoi(e) : -1; }();
  static const uint32_t rescue_waves_fused = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_FUSE_E_WAVES"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 8) ? (uint32_t)v : 3u; }();
  V2TuneSlot *eslot = nullptr;      // (the size class's slot, where the handle keeps its decision)
  hipEvent_t e_ev_start = nullptr, e_ev_stop = nullptr;      // this call's pair, when it is one of the two timed samples
  if constexpr (CAN_FUSE_E) {
    const int ec = V2Tune::size_class(B.n_reads);
    if (P.tune && ec >= 0 && !retry && !cfg.flags && ring_batches) eslot = &P.tune[o].slot[ec];
    if (eslot && eslot->fuse_e == -1 && eslot->e_sampling && hipEventQuery(eslot->ev_counts) == hipSuccess) {
      // the first launch's lists have been counted: list E's share of the reads decides (config 2: 10 % — inside the scan, three
      // rescue waves per block, the step 0.316 ms whatever state the box is in against 0.31-0.35 as a role; config 5's mouse chains
      // at 2 % substitutions: 30 % — a role: 0.609 against 0.691 ms; profiles/r06/list_e_in_the_scan_ab.log)
      uint64_t e_entries = 0;
      for (uint32_t r = 0; r < eslot->e_regions; r++) e_entries += eslot->h_counts[(size_t)V2_L_COUNTS * r + V2_L_E];
      eslot->e_share = eslot->e_reads ? (float)((double)e_entries / (double)eslot->e_reads) : 0.f;
      eslot->fuse_e = eslot->e_share <= V2_FUSE_E_MAX_SHARE ? -2 : 0;
      eslot->e_sampling = false;
      static const bool say = dcrx_debug_env("DCRX_DEBUG_TUNE") != nullptr;
      if (say) fprintf(stderr, "dcrx tune: list E holds %.1f %% of %llu reads, frame %d: %s\n", 100.0 * eslot->e_share, (unsigned long long)eslot->e_reads, o,
                       eslot->fuse_e ? "both forms will be timed" : "stays a role of the finishing launch");
    }
    (void)hipGetLastError();
    bool want = fuse_e_env >= 0 ? fuse_e_env != 0 : (eslot && eslot->fuse_e == 1);
    // The share allows it: three launches as a role and three fused under a pair of events each (start on the scan's dispatch, stop on
    // the finishing launch's), from the class's ninth eligible launch on (the clocks have come up), once the handle's rescue waves
    // are settled and on launches that carry no events of the caller's; the faster form stays.  What decides is not the share alone: config 5's mouse chains hold 10 % of list-E entries per chain as
    // config 2 does and lose 13-28 % fused (their entries take a rescue wave half as long again), and on a box whose scan runs at
    // its faster pace the two forms of config 2 are within 2 % of each other (profiles/r06/list_e_in_the_scan_ab.log).
    if (fuse_e_env < 0 && eslot && eslot->fuse_e == -2 && eslot->choice != 0u && !ev_start && !ev_stop && !P.ev_step_start && !P.ev_step_stop && !(sink && P.sink.dev)) {
      bool ok = true;
      constexpr int NP = V2TuneSlot::E_PAIRS, FIRST = V2TuneSlot::E_FIRST;
      if (!eslot->ev_e[0][0])
        for (int a = 0; a < 2 * NP && ok; a++) for (int b = 0; b < 2 && ok; b++) ok = hipEventCreate(&eslot->ev_e[a][b]) == hipSuccess;
      const int k = eslot->e_phase - FIRST;      // index of this launch among the timed ones
      if (!ok) { (void)hipGetLastError(); eslot->fuse_e = 0; }
      else if (k < 0) eslot->e_phase++;
      // NP launches with list E a role, one fused launch that is not timed (the other kernel's code is cold, the blocks' hints are the
      // role form's), NP fused ones: timed in turns the two forms paid for each switch and read within 1 % of each other on a box
      // where the fused form is 5 % faster in the steady state
      else if (k < NP) { e_ev_start = eslot->ev_e[k][0]; e_ev_stop = eslot->ev_e[k][1]; want = false; eslot->e_phase++; }
      else if (k == NP) { want = true; eslot->e_phase++; }
      else if (k <= 2 * NP) { e_ev_start = eslot->ev_e[k - 1][0]; e_ev_stop = eslot->ev_e[k - 1][1]; want = true; eslot->e_phase++; }
      else {
        bool ready = true;
        for (int a = 0; a < 2 * NP && ready; a++) ready = hipEventQuery(eslot->ev_e[a][1]) == hipSuccess;
        want = true;      // (the fused form runs on while its samples are read: one switch fewer if it stays)
        if (ready) {
          float ms[2] = {0.f, 0.f};
          for (int a = 0; a < 2 * NP && ok; a++) { float t = 0.f; ok = hipEventElapsedTime(&t, eslot->ev_e[a][0], eslot->ev_e[a][1]) == hipSuccess; ms[a < NP ? 0 : 1] += t; }
          eslot->us_e[0] = 1e3f * ms[0] / NP; eslot->us_e[1] = 1e3f * ms[1] / NP;
          eslot->fuse_e = (ok && ms[1] > 0.f && ms[1] < 0.985f * ms[0]) ? 1 : 0;
          want = eslot->fuse_e == 1;
          static const bool say = dcrx_debug_env("DCRX_DEBUG_TUNE") != nullptr;
          if (say) fprintf(stderr, "dcrx tune: scan + finishing launch of %llu reads, frame %d: %.1f us with list E a role, %.1f inside the scan -> %s\n",
                           (unsigned long long)B.n_reads, o, eslot->us_e[0], eslot->us_e[1], eslot->fuse_e ? "inside the scan" : "a role");
        }
      }
      (void)hipGetLastError();
    }
    if (ring_batches && want && !retry && !(cfg.flags & ~(DCRX_F_V2_SHAPE(3)))) {
      const uint32_t fixed = v2_scan_lds_bytes(T, o) + (T.lds_image_bytes - T.dfa_bytes) + T.v2[o].bk_bytes + 64u * V2_ERING_BATCHES * (uint32_t)v2_ering_stride<NW>() * 4u +
                             DCRX_N_COUNTERS * 4u;
      for (uint32_t nb = ring_batches; nb >= 8u; nb >>= 1)
        if (fixed + nb * 64u * V2_RING_STRIDE * 4u <= 160u * 1024u) { ring_batches_e = nb; scan_lds_e = fixed + nb * 64u * V2_RING_STRIDE * 4u; break; }
      if (!ring_batches_e && eslot && fuse_e_env < 0) eslot->fuse_e = 0;      // (no room for the event ring beside this table: a role it stays)
      if (ring_batches_e) {
        fuse_e = true;
        ks_e = o ? scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1, false, true> : scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0, false, true>;
      }
    }
  }
  if (P.tune && ring_batches) P.tune[o].last_form = 3u;      // (launch_decombine has said 2; 4 below where list E rides inside the scan as well)
  const uint32_t scan_lds = v2_scan_lds_bytes(T, o) + (ring_batches ? (T.lds_image_bytes - T.dfa_bytes) + T.v2[o].bk_bytes + ring_batches * 64u * V2_RING_STRIDE * 4u : 0u);
  auto ke = o ? events2_kernel<UNIFORM, NW, 1> : events2_kernel<UNIFORM, NW, 0>;
  auto kt = o ? tail2_kernel<UNIFORM, NW, 1> : tail2_kernel<UNIFORM, NW, 0>;
  auto kr = o ? rescue2_kernel<UNIFORM, NW, 1> : rescue2_kernel<UNIFORM, NW, 0>;
  auto kf = o ? finish2_kernel<UNIFORM, NW, 1> : finish2_kernel<UNIFORM, NW, 0>;
  auto kf0 = o ? finish2_kernel<UNIFORM, NW, 1, false> : finish2_kernel<UNIFORM, NW, 0, false>;      // ... without the tail role's code
  // the kernels that also serve a tuple sink (the 150-nt shapes only; other shapes' calls compact their records)
  constexpr bool CAN_SINK = NW == 10;
  auto ks_sink = ks;
  auto kf_sink = kf, kf0_sink = kf0;
  if constexpr (CAN_SINK) {
    kf_sink = o ? finish2_kernel<UNIFORM, NW, 1, true, true> : finish2_kernel<UNIFORM, NW, 0, true, true>;
    kf0_sink = o ? finish2_kernel<UNIFORM, NW, 1, false, true> : finish2_kernel<UNIFORM, NW, 0, false, true>;
    if constexpr (CAN_FUSE) {
      if (ring_batches) ks_sink = o ? scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1, true> : scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0, true>;
    }
  }
  auto kl = o ? left2_kernel<UNIFORM, NW, 1> : left2_kernel<UNIFORM, NW, 0>;
  static bool seen[64];
  hipError_t e;
  if (first_use_on_device(seen)) {
    {      // the scan kernels address their pair table from LDS address 0: no static LDS may sit in front of the dynamic segment
      const void *scans[] = {reinterpret_cast<const void *>(ks), reinterpret_cast<const void *>(ks_sink)};
      for (const void *k : scans) {
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, k);
        if (e != hipSuccess) return e;
        if (fa.sharedSizeBytes != 0) return hipErrorNotSupported;      // (dcrx_api.cpp: DCRX_E_HIP with the runtime's text; a toolchain that moves the LDS base)
      }
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if constexpr (CAN_FUSE_E) {
      const void *ke2[] = {reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0, false, true>),
                           reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1, false, true>)};
      for (const void *k : ke2) {
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, k);
        if (e != hipSuccess) return e;
        if (fa.sharedSizeBytes != 0) return hipErrorNotSupported;
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
    }
    if constexpr (CAN_FUSE) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(ke), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf0), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    if constexpr (CAN_SINK) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf_sink), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(kf0_sink), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      if (e != hipSuccess) return e;
      if constexpr (CAN_FUSE) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan2_kernel<UNIFORM, NW, RPL, NARROW, PREFETCH, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
    }
    attributes_set_on_device(seen);
  }
#ifdef DCRX_SCAN_STAMPS
  {
    static uint32_t *stamps = nullptr;
    if (!stamps) { (void)hipMalloc(&stamps, (size_t)4096 * 16 * 16); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_stamps), &stamps, sizeof(stamps)); }
  }
#endif
  if (B.n_reads == 0) return hipSuccess;       // (the tallies stay zero; the list kernel hands them over)
  const uint32_t cus = P.n_cu > P.reserved_cus ? P.n_cu - P.reserved_cus : 1u;
  // One scan block per compute unit, each with a contiguous range of the reads — a multiple of 512: whole items of its waves and
  // whole 64-byte lines of the exception bitmap (v2_exc_slice) — and one region of every list, sized for the reads the block can
  // meet; a small batch takes fewer blocks, 16 items each at least.
  const uint64_t wt = 64ull * RPL;
  const uint64_t n_items = (B.n_reads + wt - 1) / wt;
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cus, (n_items + 15) / 16));
  const uint64_t per_block = (((B.n_reads + grid - 1) / grid + 511) / 512) * 512;
  V2Lists Q;
  Q.tail = P.v2_tail; Q.ev = P.v2_events; Q.sx = P.v2_slow; Q.left = P.v2_left; Q.counts = P.v2_counts;
  const uint32_t n_regions = grid;
  const uint64_t pb128 = (per_block + 255) & ~127ull;           // (a block's reads, rounded up to whole chunks)
  Q.tcap = (uint32_t)std::min<uint64_t>(pb128, P.v2_tail_rows / V2Rows<NW>::T / n_regions) & ~63u;      // (whole chunks of 64 slots)
  Q.ecap = (uint32_t)std::min<uint64_t>(pb128, P.v2_event_rows / V2Rows<NW>::E / n_regions) & ~127u;    // (region `sx`, of the same size, holds two lists of whole chunks)
  Q.scap = (uint32_t)std::min<uint64_t>(Q.ecap, P.v2_slow_rows / V2Rows<NW>::E / n_regions) & ~127u;
  // (a launch that keeps the tail a role of the finishing launch needs the tail list, which a handle that has fused so far does
  // not hold: hipErrorNotReady before anything is launched — dcrx_api.cpp allocates it and comes back)
  if (!ring_batches && (!P.v2_tail || Q.tcap < 64)) return B.n_reads ? hipErrorNotReady : hipSuccess;
  if (ring_batches) Q.tcap = (uint32_t)pb128 & ~63u;      // (the fused form: the ring's sequence stands in for the list — the tuple sink's tail section is as long)
  if (Q.ecap < 128 || Q.scap < 128) return hipErrorInvalidValue;      // the workspace was not sized for this batch (dcrx_api.cpp sizes it)
  // Launch order, all on the caller's stream: scan -> finish2 (the lean rescue over lists E and C, the lean tail and the general
  // form over list X as roles of one launch) -> (dcrx_kernels.hip) the list kernel.  A/B and tests: DCRX_F_V2_SIDE_STREAMS puts the
  // tail kernel and the X pass on two side streams of the handle beside the rescue kernel (round 3's shape: forked from the scan
  // dispatch's stop event, joined through the side kernels' own stop events), DCRX_F_V2_LEAN_SERIAL runs the three as launches of
  // their own one after the other.
  const bool finish = !(cfg.flags & (DCRX_F_PROFILE_SCAN_ONLY | DCRX_F_PROFILE_NO_FINISH));
  const bool separate = (cfg.flags & (DCRX_F_V2_SIDE_STREAMS | DCRX_F_V2_LEAN_SERIAL | DCRX_F_V2_NO_LEAN_RESCUE)) != 0u;
  const bool side = finish && (cfg.flags & DCRX_F_V2_SIDE_STREAMS) && !(cfg.flags & DCRX_F_V2_LEAN_SERIAL) && P.v2_side && P.v2_side2 && P.v2_ev_fork &&
                    P.v2_ev_join && P.v2_ev_join2;
  // (the caller's stop event for the scan, when there is one — timing, or a caller that orders other work behind the scan —
  // serves as the fork event as well: one signal on the dispatch, no marker packet)
  const hipEvent_t fork_ev = ev_stop ? ev_stop : P.v2_ev_fork;
  static const uint32_t tail_waves_forced = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_TAIL_WAVES"); const int v = e ? atoi(e) : 0; return (v >= 2 && v <= 8) ? (uint32_t)v : 0u; }();      // (tests, A/B)
  // The call's tuple sink (dcrx_sink_device.h), when this launch serves it: the shipped shape (one finishing launch), one pass,
  // no profiling switch, and room: a slab per region with a section per list and one for the late items, a region's bitmap and
  // its ranks in the place kernel's LDS.
  V2SinkCall S{};
  if (CAN_SINK && sink && P.sink.dev && finish && !separate && !retry && !(cfg.flags & DCRX_F_PROFILE_MASK)) {
    const uint64_t stride = (uint64_t)Q.tcap + Q.ecap + Q.scap / 2 + pb128;
    if (n_regions <= P.sink.regions_cap && stride * n_regions <= P.sink.items_cap && stride < (1ull << 32) && per_block / 32 * 8 <= 144u * 1024u &&
        per_block < 0xFFFFFFull) {
      S.dev = P.sink.dev; S.items = P.sink.items; S.hits = P.sink.hits; S.stride = (uint32_t)stride; S.e_off = Q.tcap; S.c_off = Q.tcap + Q.ecap; S.late_off = Q.tcap + Q.ecap + Q.scap / 2;
      S.late_cap = (uint32_t)pb128; S.per_block = (uint32_t)per_block; S.wpack = P.sink.wpack;
      sink->S = S; sink->n_regions = n_regions; sink->tcap = Q.tcap; sink->ecap = Q.ecap; sink->ccap = Q.scap / 2; sink->fused = ring_batches ? 1u : 0u;
      sink->counts = Q.counts;
    }
  }
  if (S.dev) fuse_e = false;      // (a call that leaves a tuple sink's message keeps list E a role: its items are placed by list position)
  if (P.tune && fuse_e) P.tune[o].last_form = 4u;
  // (experiment, DCRX_DEBUG_SCAN_THREADS=768: three waves per SIMD in a scan block, so that a wave of another stream's finishing launch fits beside it)
  static const uint32_t scan_threads = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_SCAN_THREADS"); const int v = e ? atoi(e) : 0; return (v >= 512 && v <= DCRX_V2_BLOCK && v % 64 == 0) ? (uint32_t)v : (uint32_t)DCRX_V2_BLOCK; }();
  hipExtLaunchKernelGGL(S.dev ? ks_sink : (fuse_e ? ks_e : ks), dim3(grid), dim3(scan_threads), fuse_e ? scan_lds_e : scan_lds, s, ev_start ? ev_start : e_ev_start, side ? fork_ev : ev_stop, 0, T, B, cfg, rec,
                        d_counters, Q, queue, gqueue, qcap, queue_count, per_block, retry, P.dev_tables, fuse_e ? ring_batches_e : ring_batches, S, tail_waves_forced,
                        fuse_e ? rescue_waves_fused : 0u);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if constexpr (CAN_FUSE_E) {
    if (eslot && eslot->fuse_e == -1 && !eslot->e_sampling && !fuse_e && fuse_e_env < 0 && !S.dev && finish) {
      // the class's first launch: its regions' list counts to pinned memory behind the scan (once per handle, frame and size class)
      bool ok = eslot->h_counts || hipHostMalloc(reinterpret_cast<void **>(&eslot->h_counts), (size_t)4096 * V2_L_COUNTS * 4, hipHostMallocDefault) == hipSuccess;
      ok = ok && (eslot->ev_counts || hipEventCreateWithFlags(&eslot->ev_counts, hipEventDisableTiming) == hipSuccess);
      ok = ok && grid <= 4096u && hipMemcpyAsync(eslot->h_counts, Q.counts, (size_t)grid * V2_L_COUNTS * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
           hipEventRecord(eslot->ev_counts, s) == hipSuccess;
      if (ok) { eslot->e_sampling = true; eslot->e_regions = grid; eslot->e_reads = B.n_reads; }
      else { (void)hipGetLastError(); eslot->fuse_e = 0; }
    }
  }
  if (finish) {
    // waves of the finishing roles that share a region (a scan block's list): as many as keep 8192 waves on the tail list and
    // 4096 on each rescue list of a full-size launch
    // (measured, profiles/r04/rescue_waves_*.log, config3_finish_waves_*.log: where the launch also holds the tail role — the
    // extended sets — 4 096 tail waves and 3 072 rescue waves per list run config 3's step 6 % faster than 8 192 and 4 096: fewer
    // blocks wait for a slot; where the scan has taken the tail, config 2 would take 3 072 rescue waves (- 2 %) and config 5 loses
    // 4 % on them: the 4 096 stay there)
    static const uint32_t rescue_waves_env = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_RESCUE_WAVES"); const int v = e ? atoi(e) : 0; return v >= 256 ? (uint32_t)v : 0u; }();      // (A/B)
    // (batches of 2^25 reads and more, fused form: 8 192 — config 5 at 100 M reads per launch 18.4 against 17.4 G reads/s, config 2
    // 32.0 against 31.4; at 30 M + 2 % / none: profiles/r05/rescue_waves_by_batch_size.log)
    const bool big = B.n_reads >= V2Tune::BIG_BATCH;
    const uint32_t waves_first = (ring_batches && !separate && big) ? 8192u : 4096u, waves_second = (ring_batches && !separate && big) ? 4096u : 3072u;
    // (the tail as a role of this launch — the extended sets of config 3 —: 3 072, and 4 096 for big batches: 19.3 against 18.8 G reads/s
    // at 100 M reads, 16.3 against 16.6 at 10 M; profiles/r05/config3_waves_by_batch_size.log)
    uint32_t rescue_waves = rescue_waves_env ? rescue_waves_env : ((ring_batches || separate) ? waves_first : (big ? 4096u : 3072u));
    // the fused form: the handle's own choice between 4 096 and 3 072 (V2Tune), timed on its first launches of this batch size
    hipEvent_t tune_start = nullptr, tune_stop = nullptr;
    static const bool tune_off = dcrx_debug_env("DCRX_DEBUG_NO_TUNE") != nullptr;      // (tests, A/B)
    const int tune_class = V2Tune::size_class(B.n_reads);
    if (ring_batches && !separate && !rescue_waves_env && !tune_off && P.tune && !retry && !cfg.flags && tune_class >= 0) {
      V2TuneSlot &U = P.tune[o].slot[tune_class];      // (a size class of its own for every power of two: a short last batch does not unsettle the others')
      if (U.choice) rescue_waves = U.choice;
      else {
        const int k = U.launches - 1;            // sample index of this launch (the first launch of a size is not timed)
        if (k >= 0 && k < V2Tune::SAMPLES) {
          if (!U.created) {
            bool ok = true;
            for (int i = 0; i < V2Tune::SAMPLES && ok; i++)
              ok = hipEventCreate(&U.ev[i][0]) == hipSuccess && hipEventCreate(&U.ev[i][1]) == hipSuccess;
            U.created = ok;
            if (!ok) { (void)hipGetLastError(); U.choice = waves_first; }
          }
          if (U.created) { rescue_waves = (k & 1) ? waves_second : waves_first; tune_start = U.ev[k][0]; tune_stop = U.ev[k][1]; }
        } else if (k >= V2Tune::SAMPLES && U.created) {
          // (a big batch's samples may be waited for, once, where the caller has said so — dcrx_set_tune_wait —: a caller that queues
          // such launches ahead of the device — each takes milliseconds — would otherwise never find them complete; without that
          // permission the call keeps its contract of never waiting and the handle stays on the first setting until a query succeeds)
          if (big && k == V2Tune::SAMPLES && P.tune[o].may_wait) (void)hipEventSynchronize(U.ev[V2Tune::SAMPLES - 1][1]);      // (only where the caller allowed it: dcrx_set_tune_wait)
          bool ready = true;
          for (int i = 0; i < V2Tune::SAMPLES && ready; i++) ready = hipEventQuery(U.ev[i][1]) == hipSuccess;
          (void)hipGetLastError();
          if (ready) {
            float ms[2] = {0.f, 0.f};
            bool ok = true;
            for (int i = 0; i < V2Tune::SAMPLES && ok; i++) {
              float t = 0.f;
              ok = hipEventElapsedTime(&t, U.ev[i][0], U.ev[i][1]) == hipSuccess;
              ms[i & 1] += t;
            }
            (void)hipGetLastError();
            U.choice = (ok && ms[1] < 0.985f * ms[0]) ? waves_second : waves_first;
            if (ok) { U.us[0] = 1e3f * ms[0] / (V2Tune::SAMPLES / 2); U.us[1] = 1e3f * ms[1] / (V2Tune::SAMPLES / 2); }
            rescue_waves = U.choice;
            static const bool say = dcrx_debug_env("DCRX_DEBUG_TUNE") != nullptr;
            if (say) fprintf(stderr, "dcrx tune: finishing launches of %llu reads, frame %d: %.1f us on %u rescue waves, %.1f on %u -> %u (launch %d)\n",
                             (unsigned long long)B.n_reads, o, 1e3f * ms[0] / (V2Tune::SAMPLES / 2), waves_first, 1e3f * ms[1] / (V2Tune::SAMPLES / 2), waves_second, U.choice, U.launches);
          }
        }
        U.launches++;
      }
    }
    static const uint32_t tail_role_waves_env = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_TAIL_ROLE_WAVES"); const int v = e ? atoi(e) : 0; return v >= 256 ? (uint32_t)v : 0u; }();      // (A/B)
    const uint32_t tail_role_waves = tail_role_waves_env ? tail_role_waves_env : (separate ? 8192u : 4096u);
    const uint32_t tsplit = std::max<uint32_t>(1u, std::min<uint32_t>(64u, tail_role_waves / n_regions));
    const uint32_t rsplit_full = std::max<uint32_t>(1u, std::min<uint32_t>(64u, rescue_waves / n_regions));
    const uint32_t rsplit = fuse_e ? 1u : rsplit_full;      // (FUSE_E: list E is empty — its entries went through the scan's event ring —: one wave per region looks)
    static const uint32_t rescue_waves_c = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_RESCUE_WAVES_C"); const int v = e ? atoi(e) : 0; return v >= 256 ? (uint32_t)v : 0u; }();      // (A/B)
    const uint32_t csplit = rescue_waves_c ? std::max<uint32_t>(1u, std::min<uint32_t>(64u, rescue_waves_c / n_regions)) : rsplit_full;
    // (list C's jobs behind list E's on the same waves — one round of blocks instead of two — were measured: the step 3 % longer
    // on config 2, 8 % on config 5: list C's batches are the slow ones, two sweeps each, and want to start with the launch;
    // profiles/r05/finish_list_c_folded_ab.log.  List C's jobs IN FRONT of list E's on the same waves: 1-2 % longer still
    // than list E's blocks first and list C's in the second round, as shipped; finish_one_round_c_first_ab.log)
    const uint32_t fgrid = (n_regions * (rsplit + csplit) + DCRX_V2_FBLOCK / 64 - 1) / (DCRX_V2_FBLOCK / 64);
    const uint32_t egrid = (n_regions + DCRX_V2_FBLOCK / 64 - 1) / (DCRX_V2_FBLOCK / 64);      // the general form over a whole event list (A/B): a block takes four regions
    // blocks of a short list's pass that share a region: as a pass of its own (A/B forms) the list's latency is the launch's, and
    // four blocks per region halve it; as a role under the lean rescue one block per region does (its rounds run hidden)
    static const uint32_t bsplit_env = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_X_BSPLIT"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 16) ? (uint32_t)v : 0u; }();      // (A/B)
    // (FUSE_E: with list E gone the launch is as long as list X's chains of single reads: two blocks per region 46 us, one 57, four 65 —
    // profiles/r06/list_e_in_the_scan_ab.log)
    const uint32_t bsplit = bsplit_env ? bsplit_env : (separate ? std::max<uint32_t>(1u, std::min<uint32_t>(16u, 1024u / n_regions)) : (fuse_e ? 2u : 1u));
    const uint32_t sgrid = n_regions * bsplit;
    const uint32_t flds = v2_finish_lds_bytes(T, o);
    const uint32_t llds = v2_finish_block_lds<NW>(T, o, DCRX_V2_FBLOCK);      // the lean roles: + a strip per lane (+ the rescue's scratch counters)
    const uint32_t tlds = v2_finish_block_lds<NW>(T, o, DCRX_V2_TBLOCK);
    const uint32_t elds_ext = flds + (T.lds_image2_bytes - T.lds_image_bytes);
    const uint32_t ext = elds_ext <= 64u * 1024u ? 1u : 0u;      // the event kernel's LDS with the germline regions in it
    const uint32_t elds = ext ? elds_ext : flds;
    const uint32_t slow_width = 4u;        // lanes of a wave that take entries of a short list (64 / 16 / 4 / 2 / 1: 97 / 62 / 57 / 65 / 79 us)
    const uint32_t tgrid = (n_regions * tsplit + DCRX_V2_TBLOCK / 64 - 1) / (DCRX_V2_TBLOCK / 64);
    V2Roles R;
    R.xgrid = sgrid; R.rgrid = fgrid; R.tgrid = ring_batches ? 0u : (n_regions * tsplit + DCRX_V2_FBLOCK / 64 - 1) / (DCRX_V2_FBLOCK / 64);      // (fused: the tail list is empty)
    R.rsplit = rsplit | (csplit << 8); R.tsplit = tsplit; R.bsplit = bsplit; R.width = slow_width;
    V2FinishArgs A;
    A.T0 = T; A.B = B; A.cfg = cfg; A.records = rec; A.counters = d_counters; A.Q = Q; A.n_regions = n_regions; A.R = R;
    A.queue = queue; A.gqueue = gqueue; A.qcap = qcap; A.queue_count = queue_count; A.Tmem = P.dev_tables; A.S = S;
    if (!separate) {
      hipExtLaunchKernelGGL(S.dev ? (ring_batches ? kf0_sink : kf_sink) : (ring_batches ? kf0 : kf), dim3(R.xgrid + R.rgrid + R.tgrid), dim3(DCRX_V2_FBLOCK), llds, s, tune_start,
                            tune_stop ? tune_stop : e_ev_stop, 0, A);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
    } else {
      auto general = [&](hipStream_t st, const int which, const bool whole_list, hipEvent_t stop) -> hipError_t {
        if (whole_list)
          hipExtLaunchKernelGGL(ke, dim3(egrid), dim3(DCRX_V2_FBLOCK), elds, st, nullptr, stop, 0, T, B, cfg, rec, d_counters, Q, which, (uint32_t)(DCRX_V2_FBLOCK / 64), 64u, 1u, ext,
                                n_regions, queue, gqueue, qcap, queue_count);
        else
          hipExtLaunchKernelGGL(ke, dim3(sgrid), dim3(DCRX_V2_FBLOCK), elds, st, nullptr, stop, 0, T, B, cfg, rec, d_counters, Q, which, 1u, slow_width, bsplit, ext, n_regions, queue,
                                gqueue, qcap, queue_count);
        return hipGetLastError();
      };
      if (side) {
        e = hipStreamWaitEvent(P.v2_side, fork_ev, 0); if (e != hipSuccess) return e;
        hipExtLaunchKernelGGL(kt, dim3(tgrid), dim3(DCRX_V2_TBLOCK), tlds, P.v2_side, nullptr, P.v2_ev_join, 0, T, B, cfg, rec, d_counters, Q, n_regions, tsplit, queue,
                              gqueue, qcap, queue_count, P.dev_tables);
        e = hipGetLastError(); if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(P.v2_side2, fork_ev, 0); if (e != hipSuccess) return e;
        e = general(P.v2_side2, V2_L_X, false, P.v2_ev_join2); if (e != hipSuccess) return e;
      }
      // the scan kernel's event lists E and C: the lean rescue, or — A/B — the general form at once
      if (cfg.flags & DCRX_F_V2_NO_LEAN_RESCUE) {
        for (int which = V2_L_E; which <= V2_L_C; which++) { e = general(s, which, true, nullptr); if (e != hipSuccess) return e; }
      } else {
        hipLaunchKernelGGL(kr, dim3(fgrid), dim3(DCRX_V2_FBLOCK), llds, s, T, B, cfg, rec, d_counters, Q, n_regions, rsplit | (csplit << 8), queue, gqueue, qcap,
                           queue_count, P.dev_tables);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      if (side) {
        e = hipStreamWaitEvent(s, P.v2_ev_join, 0); if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(s, P.v2_ev_join2, 0); if (e != hipSuccess) return e;
      } else {
        hipLaunchKernelGGL(kt, dim3(tgrid), dim3(DCRX_V2_TBLOCK), tlds, s, T, B, cfg, rec, d_counters, Q, n_regions, tsplit, queue, gqueue, qcap, queue_count, P.dev_tables);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = general(s, V2_L_X, false, nullptr); if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(kl, dim3(1), dim3(DCRX_V2_FBLOCK), llds, s, A);      // what the lean kernels left
      e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    static const bool dbg = dcrx_debug_env("DCRX_DEBUG_V2_COUNTS") != nullptr;
    if (dbg && e == hipSuccess) {          // debugging aid: the lists' populations (synchronises)
      std::vector<uint32_t> h((size_t)V2_L_COUNTS * n_regions);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h.data(), Q.counts, h.size() * 4, hipMemcpyDeviceToHost);
      unsigned long long t[V2_L_COUNTS] = {0};
      for (uint32_t r = 0; r < n_regions; r++) for (int k = 0; k < V2_L_COUNTS; k++) t[k] += h[(size_t)V2_L_COUNTS * r + k];
      fprintf(stderr, "dcrx v2 lists: regions %u tail %llu E %llu C %llu X %llu; LDS of a finishing block %u bytes (side tables %u, buckets %u), of a scan block %u\n", n_regions,
              t[V2_L_TAIL], t[V2_L_E], t[V2_L_C], t[V2_L_X], llds, T.lds_image_bytes - T.dfa_bytes, T.v2[o].bk_bytes, scan_lds);
    }
  }
#ifdef DCRX_SCAN_STAMPS
  {      // instrumented build (tools/): when do a scan block's waves start, leave their loops and end?  (synchronises; launch DCRX_STAMPS_LAUNCH of the process, default 30)
    static int launches = 0;
    static const int want = [] { const char *e = dcrx_debug_env("DCRX_STAMPS_LAUNCH"); return e ? atoi(e) : 30; }();
    if (launches++ == want && e == hipSuccess) {
      const size_t nw = (size_t)n_regions * (DCRX_V2_BLOCK / 64);
      std::vector<uint32_t> h(4 * nw);
      (void)hipStreamSynchronize(s);
      uint32_t *dptr = nullptr;
      (void)hipMemcpyFromSymbol(&dptr, HIP_SYMBOL(g_scan_stamps), sizeof(dptr));
      (void)hipMemcpy(h.data(), dptr, h.size() * 4, hipMemcpyDeviceToHost);
      if (const char *dump = dcrx_debug_env("DCRX_STAMPS_DUMP")) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(h.data(), 4, h.size(), f); fclose(f); } }
      uint32_t t0 = 0xFFFFFFFFu;
      for (size_t i = 0; i < nw; i++) t0 = std::min(t0, h[4 * i]);
      auto pcs = [&](const char *what, std::vector<double> v) {
        std::sort(v.begin(), v.end());
        auto pc = [&](double q) { return v[(size_t)(q * (v.size() - 1))]; };
        fprintf(stderr, "dcrx scan stamps: %-44s min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f p99 %6.1f max %6.1f us\n", what, pc(0), pc(0.1), pc(0.5), pc(0.9), pc(0.99), pc(1.0));
      };
      std::vector<double> st, su, le_scan, le_tail, en, blk_scan_end, blk_end;
      for (uint32_t r = 0; r < n_regions; r++) {
        double last_scan = 0, last = 0;
        for (int wv = 0; wv < DCRX_V2_BLOCK / 64; wv++) {
          const uint32_t *c = &h[4 * ((size_t)r * (DCRX_V2_BLOCK / 64) + wv)];
          st.push_back((c[0] - t0) / 100.0); su.push_back((c[1] - t0) / 100.0); en.push_back((c[3] - t0) / 100.0);
          const bool tailw = ring_batches && wv >= (DCRX_V2_BLOCK / 64) - 3;      // (a guess at three tail waves: the block's own choice is not known here)
          (tailw ? le_tail : le_scan).push_back((c[2] - t0) / 100.0);
          if (!tailw) last_scan = std::max(last_scan, (c[2] - t0) / 100.0);
          last = std::max(last, (c[3] - t0) / 100.0);
        }
        blk_scan_end.push_back(last_scan); blk_end.push_back(last);
      }
      pcs("wave start", st); pcs("wave set-up done (tables staged)", su); pcs("scanning wave leaves its loop", le_scan);
      if (!le_tail.empty()) pcs("tail wave leaves its loop (last three waves)", le_tail);
      pcs("wave end", en); pcs("per block: last scanning wave leaves", blk_scan_end); pcs("per block: end", blk_end);
    }
  }
#endif
  return e;
}

// the tuple sink's place kernel behind the list kernel (the call's last launch: its stop event rides here)
hipError_t launch_v2_place(const LaunchPlan &P, const V2SinkLaunch &K, uint64_t n_reads, hipStream_t s, hipEvent_t ev_stop) {
  static bool seen[64];
  if (first_use_on_device(seen)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(v2_place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attributes_set_on_device(seen);
  }
  // LDS: the region's bitmap and its ranks, then room to stage its tuples (5 bytes each) for whole-line stores: as many as fit
  const uint32_t words = (K.S.per_block / 32u) * 2u;
  static const int hcap_forced = [] { const char *e = dcrx_debug_env("DCRX_DEBUG_PLACE_HCAP"); return e ? atoi(e) : -1; }();      // (tests: the path of regions whose tuples do not fit)
  const uint32_t hcap = hcap_forced >= 0 ? ((uint32_t)hcap_forced & ~3u) : (std::min<uint32_t>(K.S.per_block, (150u * 1024u - words * 4u) / 5u) & ~3u);
  const uint32_t lds = words * 4u + hcap * 5u;
  hipExtLaunchKernelGGL(v2_place_kernel, dim3(K.n_regions), dim3(V2_PLACE_BLOCK), lds, s, nullptr, ev_stop, 0, K.S, K.counts, K.n_regions, K.tcap, K.ecap,
                        K.ccap, K.fused, (uint64_t)n_reads, P.sink.n_slots, P.sink.msg, P.sink.bytes - 4u, P.sink.d_total, hcap);
  return hipGetLastError();
}
// items of a sink's slabs for batches of up to max_reads reads: per region the three lists' capacities and as many late slots
// as the region has reads (launch_v2: tcap, ecap, scap / 2 <= pb128 each)
uint64_t v2_sink_items(uint64_t max_reads, uint32_t n_cu) { return 4 * (max_reads + (uint64_t)n_cu * 1024) + 4096; }

hipError_t launch_v2_any(const LaunchPlan &P, const DevTables &T, const BatchDev &B, const CfgDev &cfg, dcrx_record_t *rec,
                         uint32_t *queue, uint32_t *gqueue, uint32_t qcap, uint32_t *queue_count, unsigned long long *d_counters,
                         hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t retry, V2SinkLaunch *sink) {
  const int o = cfg.orientation == DCRX_ORIENT_FORWARD ? 0 : 1;
  const bool uniform = B.lens == nullptr, nw10 = B.stride <= 40, narrow = T.v2[o].narrow != 0;
  int shape = (int)((cfg.flags >> 8) & 3u);
  if (shape == 0) shape = nw10 ? 2 : 3;      // two reads per lane (two independent chains per wave) where the registers allow: measured faster than one
#define DCRX_V2A(UN, NW_, RP, NA) launch_v2<UN, NW_, RP, NA>(P, T, B, cfg, rec, queue, gqueue, qcap, queue_count, d_counters, s, ev_start, ev_stop, retry, sink)
#define DCRX_V2X(UN, NW_, RP, NA, PF) launch_v2<UN, NW_, RP, NA, PF>(P, T, B, cfg, rec, queue, gqueue, qcap, queue_count, d_counters, s, ev_start, ev_stop, retry, sink)
#define DCRX_V2(UN, NW_, NA) (shape == 3 ? DCRX_V2A(UN, NW_, 1, NA) : (shape == 1 && UN && NW_ == 10 && NA) ? DCRX_V2X(true, 10, 4, true, false) : DCRX_V2A(UN, NW_, 2, NA))
#ifdef DCRX_FAST_BUILD      // (experiment builds, tools/build_variant.sh: the benchmark's launch shape only)
  if (nw10 && uniform && shape == 2) return narrow ? DCRX_V2A(true, 10, 2, true) : DCRX_V2A(true, 10, 2, false);
  return hipErrorNotSupported;
#else
  if (nw10) {
    if (uniform) return narrow ? DCRX_V2(true, 10, true) : DCRX_V2(true, 10, false);
    return narrow ? DCRX_V2(false, 10, true) : DCRX_V2(false, 10, false);
  }
  if (B.stride <= 4 * DCRX_NWMAX) {
    if (uniform) return narrow ? DCRX_V2(true, DCRX_NWMAX, true) : DCRX_V2(true, DCRX_NWMAX, false);
    return narrow ? DCRX_V2(false, DCRX_NWMAX, true) : DCRX_V2(false, DCRX_NWMAX, false);
  }
  // 321-511 nt: one read per lane, no item in flight beside the one in hand (the registers hold one read and its log)
  if (uniform) return narrow ? DCRX_V2X(true, DCRX_V2_NWLONG, 1, true, false) : DCRX_V2X(true, DCRX_V2_NWLONG, 1, false, false);
  return narrow ? DCRX_V2X(false, DCRX_V2_NWLONG, 1, true, false) : DCRX_V2X(false, DCRX_V2_NWLONG, 1, false, false);
#endif
#undef DCRX_V2
#undef DCRX_V2A
#undef DCRX_V2X
}

// 16-byte rows the two lists need for batches of up to max_reads reads of `stride` bytes on n_cu compute units:
// a tail entry per read (every read can be one) and half as many event entries
void v2_list_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu, uint64_t *tail_rows, uint64_t *event_rows) {
  const uint64_t rt = stride <= 40 ? V2Rows<10>::T : stride <= 4 * DCRX_NWMAX ? V2Rows<DCRX_NWMAX>::T : V2Rows<DCRX_V2_NWLONG>::T;
  const uint64_t re = stride <= 40 ? V2Rows<10>::E : stride <= 4 * DCRX_NWMAX ? V2Rows<DCRX_NWMAX>::E : V2Rows<DCRX_V2_NWLONG>::E;
  const uint64_t entries = max_reads + max_reads / 8 + (uint64_t)n_cu * 16 * 256;
  *tail_rows = entries * rt;
  *event_rows = (entries / 2) * re;
}
// ... and for lists C and X (half of a region of this allocation each): as many entries as the event list
uint64_t v2_slow_rows(uint64_t max_reads, uint32_t stride, uint32_t n_cu) {
  uint64_t tr, er;
  v2_list_rows(max_reads, stride, n_cu, &tr, &er);
  return er;
}

}  // namespace dcrx
