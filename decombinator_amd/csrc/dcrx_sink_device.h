// dcrx_sink_device.h — the tuple sink of a tables handle (include/dcrx.h, dcrx_set_tuple_sink): the kernels that write a
// batch's records also leave the narrow tuples of its decombined reads behind, so that the message a sharded run gathers
// (bitmap | low words | high bytes, in read order: dcrx_compact_hits_narrow_device's) is put together from ~40 MB of items
// instead of a second and third pass over the batch's 16-byte records.
//
//   items   per region of the scan (a scan block's range of reads) one slab of 8-byte items (the tuple's low word | the read's
//           index inside the region, 24 bits, and the tuple's bits 32-39), in four sections:
//             tail   slot = the tail entry's place in the block's ring (fused scan) or in the region's tail list
//             E, C   slot = the event entry's place in the region's list
//             late   what the general form, the left list's polling wave and the list kernel decombine: slots drawn with
//                    an atomic per read (a few thousand reads per 10 M; many only on inputs full of Ns)
//           A lean role writes the item of EVERY entry it takes — the tuple, or the empty mark when the read did not
//           decombine —: no atomic whose result a wave waits for, no compaction inside the lean loops.
//   hits    per region the number of decombined reads (atomics without a result): the place kernel's offsets
//   place   (v2_place_kernel, one block per region, behind the list kernel) builds the region's bitmap in LDS from its
//           items, ranks them, writes the tuples in read order and the bitmap words, and re-arms the counters.
#pragma once

#include <cstdint>

#include "dcrx_launch.h"

namespace dcrx {

constexpr uint32_t V2_SINK_EMPTY = 0xFFFFFFFFu;      // an item's second word when its read did not decombine (a region holds fewer than 2^24 - 1 reads)

#ifdef __HIPCC__
// the narrow tuple of include/dcrx.h from a record's fields; wpack: w_v | w_j << 5 | w_vdel << 10 | w_jdel << 15 | w_pos << 20
__device__ __forceinline__ uint64_t sink_tuple(const dcrx_record_t &r, const uint32_t wpack, const uint8_t *__restrict__ j_tag_len,
                                               const int32_t *__restrict__ j_jump) {
  const uint32_t w_v = wpack & 31u, w_j = (wpack >> 5) & 31u, w_vdel = (wpack >> 10) & 31u, w_jdel = (wpack >> 15) & 31u, w_pos = (wpack >> 20) & 31u;
  const int tagpos = (int)r.ins_start + (int)r.ins_len - (int)r.jdel + j_jump[r.j];
  const uint64_t short_end = ((int)r.j_end - tagpos) != (int)j_tag_len[r.j] ? 1u : 0u;      // decombine.py:450-454
  uint64_t t = r.v;
  uint32_t sh = w_v;
  t |= (uint64_t)r.j << sh; sh += w_j;
  t |= (uint64_t)r.vdel << sh; sh += w_vdel;
  t |= (uint64_t)r.jdel << sh; sh += w_jdel;
  t |= (uint64_t)r.v_start << sh; sh += w_pos;
  t |= (uint64_t)r.j_end << sh; sh += w_pos;
  t |= short_end << sh; sh += 1;
  t |= (uint64_t)(r.frame & 1u) << sh;
  return t;
}

// the same from what a lean role has in hand: no table is read (short_end: the J half1 rescue found the J gene and the split is
// not the tag's middle; a tag of the lean forms' classes has the class's length)
__device__ __forceinline__ uint64_t sink_tuple_lean(const dcrx_record_t &r, const uint32_t wpack, const bool short_end) {
  const uint32_t w_v = wpack & 31u, w_j = (wpack >> 5) & 31u, w_vdel = (wpack >> 10) & 31u, w_jdel = (wpack >> 15) & 31u, w_pos = (wpack >> 20) & 31u;
  // (the widths are scalars; where everything below j_end fits one word — 29 bits for human beta — one 64-bit shift is all)
  const uint32_t w_lo = w_v + w_j + w_vdel + w_jdel + w_pos;
  if (w_lo <= 32u) {
    const uint32_t lo = (uint32_t)r.v | ((uint32_t)r.j << w_v) | ((uint32_t)r.vdel << (w_v + w_j)) | ((uint32_t)r.jdel << (w_v + w_j + w_vdel)) |
                        (w_pos ? ((uint32_t)r.v_start << ((w_v + w_j + w_vdel + w_jdel) & 31u)) : 0u);
    const uint32_t hi = (uint32_t)r.j_end | ((short_end ? 1u : 0u) << w_pos) | ((uint32_t)(r.frame & 1u) << (w_pos + 1u));
    return (uint64_t)lo | ((uint64_t)hi << w_lo);
  }
  uint32_t sh = w_v;
  uint64_t t = r.v;
  t |= (uint64_t)r.j << sh; sh += w_j;
  t |= (uint64_t)r.vdel << sh; sh += w_vdel;
  t |= (uint64_t)r.jdel << sh; sh += w_jdel;
  t |= (uint64_t)r.v_start << sh; sh += w_pos;
  t |= (uint64_t)((uint32_t)r.j_end | ((short_end ? 1u : 0u) << w_pos) | ((uint32_t)(r.frame & 1u) << (w_pos + 1u))) << sh;
  return t;
}

// A lean role's item: slot `slot` of the section at `section_off` of `region` (every live lane writes; `hit` lanes the tuple).
// hits_lds: the block's own tally in LDS (the fused scan: block == region), else the region's count in memory.
__device__ __forceinline__ void sink_put(const V2SinkCall &S, const uint32_t region, const uint32_t section_off, const uint32_t slot,
                                         const bool live, const bool hit, const uint32_t r, const uint64_t tuple, const int lane,
                                         uint32_t *hits_lds) {
  const unsigned long long mh = __ballot(hit);
  if (mh && lane == (int)__builtin_ctzll(mh)) {
    const uint32_t k = (uint32_t)__popcll(mh);
    if (hits_lds) (void)__hip_atomic_fetch_add(hits_lds, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_add(S.hits + region, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (live) {
    const size_t at = (size_t)region * S.stride + section_off + slot;
    const uint32_t rel = r - region * S.per_block;
    S.items[at] = make_uint2((uint32_t)tuple, hit ? (rel | ((uint32_t)(tuple >> 32) << 24)) : V2_SINK_EMPTY);      // (one 8-byte store; the place kernel reads it back out of the L2)
  }
}

// A read some general form has just finished (its record is in memory behind this lane's own store): when it decombined,
// its tuple goes to the late section of its region.  Rare paths only: one atomic with a result per read.
__device__ __forceinline__ void sink_late(const V2SinkCall &S, const dcrx_record_t *__restrict__ records, const uint32_t r) {
  typedef uint32_t dcrx_v4u __attribute__((ext_vector_type(4)));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the record's store has been performed)
  const dcrx_v4u v = *reinterpret_cast<const volatile dcrx_v4u *>(records + r);      // (past this unit's vector cache: the store went through)
  if (((v.w >> 16) & 0xFFu) != (uint32_t)DCRX_S_OK) return;
  dcrx_record_t rec;
  rec.v = (uint16_t)v.x; rec.j = (uint16_t)(v.x >> 16); rec.v_start = (uint16_t)v.y; rec.j_end = (uint16_t)(v.y >> 16);
  rec.ins_start = (uint16_t)v.z; rec.ins_len = (uint16_t)(v.z >> 16); rec.vdel = (uint8_t)v.w; rec.jdel = (uint8_t)(v.w >> 8);
  rec.status = (uint8_t)(v.w >> 16); rec.frame = (uint8_t)(v.w >> 24);
  const V2SinkDev *dev = S.dev;
  const uint64_t t = sink_tuple(rec, S.wpack, dev->j_tag_len, dev->j_jump);
  const uint32_t region = r / S.per_block;
  (void)__hip_atomic_fetch_add(S.hits + region, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t k = atomicAdd(dev->late + region, 1u);
  if (k >= S.late_cap) return;      // (cannot be: the section holds every read of the region)
  const size_t at = (size_t)region * S.stride + S.late_off + k;
  S.items[at] = make_uint2((uint32_t)t, (r - region * S.per_block) | ((uint32_t)(t >> 32) << 24));
}
#endif

}  // namespace dcrx
