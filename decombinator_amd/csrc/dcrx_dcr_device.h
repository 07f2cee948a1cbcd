// dcrx_dcr_device.h — the per-read device code of the decombine hot path: the
// DFA scan, the half-tag rescue, the germline-end walks and the filters.
// Included by dcrx_kernels.hip (device build) and by tests/host_emul/ (a
// test-only host build of the very same functions, used to debug and to run
// sanitizers without a GPU; it is never linked into libdcrx.so).
#pragma once

#include <cstdint>

#include "../../include/dcrx.h"
#include "dcrx_device.h"
#include "dcrx_launch_types.h"

#ifndef DCRX_HOST_EMUL
#define DCRX_DEV __device__ __forceinline__
#define DCRX_DEVNI __device__ __forceinline__
DCRX_DEV uint32_t dcrx_funnel_r(uint32_t lo, uint32_t hi, uint32_t sh) { return __funnelshift_r(lo, hi, sh); }
DCRX_DEV int dcrx_sbfe(int v, uint32_t off, uint32_t w) { return __builtin_amdgcn_sbfe(v, off, w); }
DCRX_DEV uint32_t dcrx_ubfe(uint32_t v, uint32_t off, uint32_t w) { return __builtin_amdgcn_ubfe(v, off, w); }
DCRX_DEV void dcrx_atomic_inc(uint32_t *p) { atomicAdd(p, 1u); }
DCRX_DEV int dcrx_ctz64(uint64_t v) { return __ffsll((unsigned long long)v) - 1; }
DCRX_DEV int dcrx_clz64(uint64_t v) { return __clzll((long long)v); }
DCRX_DEV int dcrx_popc64(uint64_t v) { return __popcll((unsigned long long)v); }
DCRX_DEV int dcrx_popc32(uint32_t v) { return __popc(v); }
DCRX_DEV int dcrx_ctz32(uint32_t v) { return __ffs((int)v) - 1; }
DCRX_DEV int dcrx_clz32(uint32_t v) { return __clz((int)v); }
// (byte BYTE of w) & mask in one VALU op (SDWA source-byte select)
template <int BYTE>
DCRX_DEV uint32_t dcrx_byte_and(uint32_t w, uint32_t mask) {
  uint32_t r;
  if (BYTE == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(mask));
  else if (BYTE == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(mask));
  else if (BYTE == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(mask));
  else asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(mask));
  return r;
}
DCRX_DEV uint32_t dcrx_brev32(uint32_t v) { return __brev(v); }
// Records are written once and never read back by these kernels: streamed past the caches, so
// that the L2 keeps the packed reads the batched tail comes back to.
DCRX_DEV void dcrx_store_record(dcrx_record_t *dst, const dcrx_record_t &rec) {
  typedef uint32_t dcrx_v4u __attribute__((ext_vector_type(4)));
  // composed from the fields (the record's little-endian layout): reading the struct through a
  // vector pointer would keep it in scratch memory
  dcrx_v4u v;
  v.x = (uint32_t)rec.v | ((uint32_t)rec.j << 16);
  v.y = (uint32_t)rec.v_start | ((uint32_t)rec.j_end << 16);
  v.z = (uint32_t)rec.ins_start | ((uint32_t)rec.ins_len << 16);
  v.w = (uint32_t)rec.vdel | ((uint32_t)rec.jdel << 8) | ((uint32_t)rec.status << 16) | ((uint32_t)rec.frame << 24);
  __builtin_nontemporal_store(v, reinterpret_cast<dcrx_v4u *>(dst));
}
// ... and through the caches: for the kernels whose records fall between records other kernels write (a 64-byte
// line of records gets its pieces from up to three kernels: merged in L2 / the Infinity Cache, not in memory)
DCRX_DEV void dcrx_store_record_cached(dcrx_record_t *dst, const dcrx_record_t &rec) {
  typedef uint32_t dcrx_v4u __attribute__((ext_vector_type(4)));
  dcrx_v4u v;
  v.x = (uint32_t)rec.v | ((uint32_t)rec.j << 16);
  v.y = (uint32_t)rec.v_start | ((uint32_t)rec.j_end << 16);
  v.z = (uint32_t)rec.ins_start | ((uint32_t)rec.ins_len << 16);
  v.w = (uint32_t)rec.vdel | ((uint32_t)rec.jdel << 8) | ((uint32_t)rec.status << 16) | ((uint32_t)rec.frame << 24);
  *reinterpret_cast<dcrx_v4u *>(dst) = v;
}
// pointer into LDS with its address space spelled out (ds_read/ds_write instead of flat_*)
typedef __attribute__((address_space(3))) uint32_t dcrx_lds_u32;
#define DCRX_TO_LDS(p) ((dcrx_lds_u32 *)(p))
DCRX_DEV uint32_t dcrx_lds_address(const uint8_t *p) {
  return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const uint8_t *)(p));
}
using ::min;
using ::max;
#else
#include <algorithm>
#define DCRX_DEV inline
#define DCRX_DEVNI inline
inline uint32_t dcrx_funnel_r(uint32_t lo, uint32_t hi, uint32_t sh) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(v >> (sh & 31));
}
inline int dcrx_sbfe(int v, uint32_t off, uint32_t w) { return (int)((uint32_t)v << (32 - off - w)) >> (32 - w); }
inline uint32_t dcrx_ubfe(uint32_t v, uint32_t off, uint32_t w) { return (v >> off) & ((1u << w) - 1u); }
inline void dcrx_atomic_inc(uint32_t *p) { ++*p; }
inline int dcrx_ctz64(uint64_t v) { return __builtin_ctzll(v); }
inline int dcrx_clz64(uint64_t v) { return __builtin_clzll(v); }
inline int dcrx_popc64(uint64_t v) { return __builtin_popcountll(v); }
inline int dcrx_popc32(uint32_t v) { return __builtin_popcount(v); }
inline int dcrx_ctz32(uint32_t v) { return __builtin_ctz(v); }
inline int dcrx_clz32(uint32_t v) { return __builtin_clz(v); }
template <int BYTE>
inline uint32_t dcrx_byte_and(uint32_t w, uint32_t mask) { return (w >> (8 * BYTE)) & mask; }
inline uint32_t dcrx_brev32(uint32_t v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}
inline void dcrx_store_record(dcrx_record_t *dst, const dcrx_record_t &rec) { *dst = rec; }
inline void dcrx_store_record_cached(dcrx_record_t *dst, const dcrx_record_t &rec) { *dst = rec; }
typedef uint32_t dcrx_lds_u32;
#define DCRX_TO_LDS(p) (p)
inline uint32_t dcrx_lds_address(const uint8_t *) { return 0; }
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
#define __align__(n) alignas(n)
using std::min;
using std::max;
#endif

namespace dcrx {

// The side-table pointers of `T`, re-pointed at an LDS copy: `lds_side` holds the bytes
// image[side_off .. lds_image_bytes) (side_off = 0: the whole image, DFA first; side_off =
// dfa_bytes: only the side tables, as the two-bases-per-step kernel stages them).
// `ext`: the extended image (up to lds_image2_bytes: the packed germline regions) was staged as well.
DCRX_DEV DevTables tables_in_lds(const DevTables &T, const uint8_t *lds_side, uint32_t side_off, bool ext = false) {
  DevTables L = T;
#define DCRX_MV(p) L.p = reinterpret_cast<decltype(L.p)>(lds_side + ((reinterpret_cast<const uint8_t *>(T.p) - T.image) - side_off))
  DCRX_MV(st_full); DCRX_MV(st_out); DCRX_MV(outs); DCRX_MV(kw_base); DCRX_MV(kw_first); DCRX_MV(kw_begin); DCRX_MV(kw_tags);
  for (int g = 0; g < 2; g++) {
    DCRX_MV(g[g].tag_len); DCRX_MV(g[g].jump); DCRX_MV(g[g].tag_pk_fwd); DCRX_MV(g[g].tag_pk_rc);
    DCRX_MV(g[g].w64_fwd); DCRX_MV(g[g].w64_rc); DCRX_MV(g[g].w64_ok); DCRX_MV(g[g].reg_len);
    if (ext) { DCRX_MV(g[g].reg_pk_off); DCRX_MV(g[g].reg_clean); DCRX_MV(g[g].reg_pk); DCRX_MV(g[g].reg_pk_rc); }
  }
#undef DCRX_MV
  return L;
}

// Copies `n16` 16-byte units from global memory into LDS; the first `n_rows16` of them are
// transition rows whose row fields get `row_base` added (rows become absolute LDS addresses).
template <int BLOCK>
DCRX_DEV void stage_lds(const uint8_t *src_bytes, uint32_t *dst_words, uint32_t n16, uint32_t n_rows16,
                        uint32_t row_base, int tid) {
  const uint4 *src = reinterpret_cast<const uint4 *>(src_bytes);
  uint4 *dst = reinterpret_cast<uint4 *>(dst_words);
  for (uint32_t i = tid; i < n16; i += BLOCK) {
    uint4 v = src[i];
    if (i < n_rows16) { v.x += row_base; v.y += row_base; v.z += row_base; v.w += row_base; }
    dst[i] = v;
  }
}

// ------------------------------------------------------------------------------
// accumulator layout for a full-tag class: count | state<<9 | end_pos<<23.
// Summed over hits; when count == 1 the upper fields are that hit's state and
// end position, otherwise they are not read.
// ------------------------------------------------------------------------------
constexpr uint32_t ACC_CNT_MASK = 0x1FFu;
constexpr int ACC_STATE_SHIFT = 9;
constexpr int ACC_POS_SHIFT = 23;

struct Counters {
  uint32_t *lds;  // [DCRX_N_COUNTERS] block-local
  DCRX_DEV void add(int idx) const { dcrx_atomic_inc(&lds[idx]); }
};

// ------------------------------------------------------------------------------
// One read seen in one frame.  Frame position i is forward position n-1-i with
// complemented bases when REV.
// ------------------------------------------------------------------------------
struct ReadView {
  const uint8_t *comp;    // 256-entry complement table (Biopython's, decombine.py:184)
  const uint32_t *words;  // this read's packed words (global memory)
  int n;
  int e0, e1;             // this read's slice of the exception list (e0 == e1: none)
  const uint16_t *exc_pos;
  const uint8_t *exc_chr;
};

template <bool REV>
struct Frame {
  static constexpr bool kRev = REV;
  static constexpr bool kWindowedWalks = false;   // the walks go window by window past the first 23 steps (the v2 event kernel's frame does; here the code is not worth its registers)
  const ReadView &r;
  DCRX_DEVNI explicit Frame(const ReadView &rv) : r(rv) {}
  // an exception byte 'N' (as the frame shows it) at frame positions [lo, hi)
  DCRX_DEVNI bool has_N(int lo, int hi) const {
    bool hasN = false;
    for (int x = r.e0; x < r.e1; x++) {
      const int i = REV ? r.n - 1 - (int)r.exc_pos[x] : (int)r.exc_pos[x];
      const uint8_t b = REV ? r.comp[r.exc_chr[x]] : r.exc_chr[x];
      if (i >= lo && i < hi && b == (uint8_t)'N') hasN = true;
    }
    return hasN;
  }
  DCRX_DEV int n() const { return r.n; }
  DCRX_DEV int fpos(int i) const { return REV ? r.n - 1 - i : i; }
  DCRX_DEV int code(int i) const {
    int m = fpos(i);
    uint32_t w = r.words[m >> 4];
    int c = (int)((w >> ((m & 15) * 2)) & 3u);
    return REV ? (c ^ 3) : c;
  }
  DCRX_DEVNI int exc_index(int i) const {
    int m = fpos(i);
    for (int x = r.e0; x < r.e1; x++)
      if ((int)r.exc_pos[x] == m) return x;
    return -1;
  }
  DCRX_DEV bool has_exc() const { return r.e1 > r.e0; }
  // no exception byte at frame positions [a, b)
  DCRX_DEV bool clean(int a, int b) const {
    for (int x = r.e0; x < r.e1; x++) {
      const int i = REV ? r.n - 1 - (int)r.exc_pos[x] : (int)r.exc_pos[x];
      if (i >= a && i < b) return false;
    }
    return true;
  }
  // the character str(read)[i] the reference would see
  DCRX_DEVNI uint8_t chr(int i) const {
    if (has_exc()) {
      int x = exc_index(i);
      if (x >= 0) { uint8_t b = r.exc_chr[x]; return REV ? r.comp[b] : b; }
    }
    return (uint8_t)("ACGT"[code(i)]);
  }
  // the exception bytes among the stored bases [b, b + 32): bit 2s for base b + s (the slots of load64(b))
  DCRX_DEV uint64_t exc_slots(int b) const {
    uint64_t m = 0;
    for (int x = r.e0; x < r.e1; x++) {
      const int d = (int)r.exc_pos[x] - b;
      if (d >= 0 && d < 32) m |= 1ull << (2 * d);
    }
    return m;
  }
  // 2*len bits of the packed forward read covering frame positions [a, a+len);
  // caller guarantees 0 <= a, a+len <= n, len <= 16, no exception inside.
  DCRX_DEV uint32_t window(int a, int len) const {
    int lo = REV ? r.n - a - len : a;
    int bit = lo * 2;
    uint32_t w0 = r.words[bit >> 5];
    int sh = bit & 31;
    uint32_t w1 = (sh + 2 * len > 32) ? r.words[(bit >> 5) + 1] : 0u;
    uint32_t v = dcrx_funnel_r(w0, w1, sh);
    return v & ((len >= 16) ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u));
  }
  // 32 bases of the packed FORWARD read starting at forward position b (slot y = base b+y);
  // caller guarantees 0 <= b and b+32 <= n.
  DCRX_DEV uint64_t load64(int b) const {
    const int bit = b * 2, i = bit >> 5, sh = bit & 31;
    const uint32_t w0 = r.words[i], w1 = r.words[i + 1];
    const uint32_t w2 = sh ? r.words[i + 2] : 0u;
    return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
  }
};

// ---- bit-parallel 10-mer window search over 32-base words -------------------------
// One bit per base slot (at bit 2*slot): set where two 2-bit-packed words differ.
DCRX_DEV uint64_t mismatch_slots(uint64_t a, uint64_t b) {
  const uint64_t x = a ^ b;
  return (x | (x >> 1)) & 0x5555555555555555ull;
}
// bit 2t = OR of slots t..t+9 (meaningful for t <= 22)
DCRX_DEV uint64_t or10_up(uint64_t y) {
  const uint64_t a2 = y | (y >> 2), a4 = a2 | (a2 >> 4), a8 = a4 | (a4 >> 8);
  return a8 | (a2 >> 16);
}
// bit 2t = OR of slots t-9..t (meaningful for t >= 9)
DCRX_DEV uint64_t or10_down(uint64_t y) {
  const uint64_t a2 = y | (y << 2), a4 = a2 | (a2 << 4), a8 = a4 | (a4 << 8);
  return a8 | (a2 << 16);
}
// smallest k in [k0, 22] whose window is mismatch-free, or -1.
// UP: window k = slots k..k+9 (z = or10_up);  DOWN: window k = slots 22-k..31-k (z = or10_down)
DCRX_DEV int first_clean_up(uint64_t z, int k0) {
  if (k0 > 22) return -1;
  const uint64_t allowed = (0x5555555555555555ull >> (2 * k0) << (2 * k0)) & ((1ull << 46) - 1ull);
  const uint64_t cand = ~z & allowed;
  return cand ? (dcrx_ctz64(cand) >> 1) : -1;
}
DCRX_DEV int first_clean_down(uint64_t z, int k0) {
  if (k0 > 22) return -1;
  // slots 9 .. 31-k0
  const uint64_t upto = (k0 == 0) ? ~0ull : ((1ull << (2 * (32 - k0))) - 1ull);
  const uint64_t allowed = 0x5555555555555555ull & upto & ~((1ull << 18) - 1ull);
  const uint64_t cand = ~z & allowed;
  return cand ? 31 - ((63 - dcrx_clz64(cand)) >> 1) : -1;
}

// the same with an upper bound: smallest k in [k0, k1] (k1 <= 22)
DCRX_DEV int first_clean_up_range(uint64_t z, int k0, int k1) {
  if (k0 > k1 || k0 > 22) return -1;
  const uint64_t allowed = (0x5555555555555555ull >> (2 * k0) << (2 * k0)) & ((1ull << (2 * (k1 + 1))) - 1ull);
  const uint64_t cand = ~z & allowed;
  return cand ? (dcrx_ctz64(cand) >> 1) : -1;
}
DCRX_DEV int first_clean_down_range(uint64_t z, int k0, int k1) {
  if (k0 > k1 || k0 > 22) return -1;
  // window k = slots 22-k .. 31-k: its top slot 31-k runs from 31-k0 down to 31-k1 (>= 9)
  const uint64_t upto = (k0 == 0) ? ~0ull : ((1ull << (2 * (32 - k0))) - 1ull);
  const uint64_t allowed = 0x5555555555555555ull & upto & ~((1ull << (2 * (31 - k1))) - 1ull);
  const uint64_t cand = ~z & allowed;
  return cand ? 31 - ((63 - dcrx_clz64(cand)) >> 1) : -1;
}

// Walks a read's exception list in frame order while a scan advances: hit(i) is true
// exactly when frame position i holds an exception byte.  One list load per exception,
// none per symbol.
template <bool REV>
struct ExcCursor {
  const ReadView &r;
  int x, nextpos;
  DCRX_DEV explicit ExcCursor(const ReadView &rv) : r(rv) {
    x = REV ? rv.e1 - 1 : rv.e0;
    load();
  }
  DCRX_DEV void load() {
    const bool any = REV ? (x >= r.e0) : (x < r.e1);
    nextpos = any ? (REV ? r.n - 1 - (int)r.exc_pos[x] : (int)r.exc_pos[x]) : 0x7FFFFFFF;
  }
  DCRX_DEV void advance() { x += REV ? -1 : 1; load(); }
  DCRX_DEV bool hit(int i) {
    if (i != nextpos) return false;
    advance();
    return true;
  }
};

// Python s[a:b] bounds on a sequence of length n
DCRX_DEV void pyslice(int n, int a, int b, int &lo, int &hi) {
  if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
  if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
  if (b < a) b = a;
  lo = a; hi = b;
}

DCRX_DEV uint32_t packed_window(const uint32_t *pk, int base_pos, int len) {
  int bit = base_pos * 2;
  uint32_t w0 = pk[bit >> 5];
  uint32_t w1 = pk[(bit >> 5) + 1];  // regions carry a spare word
  uint32_t v = dcrx_funnel_r(w0, w1, bit & 31);
  return v & ((len >= 16) ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u));
}

// 32 bases of a packed region from base b (b + 32 <= region length)
DCRX_DEV uint64_t packed_window64(const uint32_t *pk, int b) {
  const int i = b >> 4, sh = (b & 15) * 2;
  const uint32_t w0 = pk[i], w1 = pk[i + 1];
  const uint32_t w2 = sh ? pk[i + 2] : 0u;
  return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
}

// G[ga:gb] == read[ra:rb] with Python slice semantics (the comparisons at
// decombine.py:769-772 and :802-805).
template <class FR>
DCRX_DEVNI bool slice_eq(const GeneDevPtrs &G, int g, int ga, int gb, const FR &F, int ra, int rb) {
  constexpr bool REV = FR::kRev;
  const int Lg = (int)G.reg_len[g];
  const int n = F.n();
  // fast path: both slices are whole 10-mers inside their sequences, nothing but ACGT involved
  if (ga >= 0 && gb <= Lg && gb - ga == rb - ra && ra >= 0 && rb <= n && gb - ga <= 16 && gb > ga &&
      G.reg_clean[g]) {
    // a read window that holds a byte outside ACGT cannot equal a stretch of a pure-ACGT germline
    if (!F.clean(ra, rb)) return false;
    int len = gb - ga;
    uint32_t rw = F.window(ra, len);
    uint32_t gw = REV ? packed_window(G.reg_pk_rc + G.reg_pk_off[g], Lg - ga - len, len)
                      : packed_window(G.reg_pk + G.reg_pk_off[g], ga, len);
    return rw == gw;
  }
  int glo, ghi, rlo, rhi;
  pyslice(Lg, ga, gb, glo, ghi);
  pyslice(n, ra, rb, rlo, rhi);
  if (ghi - glo != rhi - rlo) return false;
  const uint8_t *gs = G.reg_bytes + G.reg_off[g];
  for (int k = 0; k < ghi - glo; k++)
    if (gs[glo + k] != F.chr(rlo + k)) return false;
  return true;
}

#if defined(DCRX_HOST_EMUL) && defined(DCRX_R2_REASONS)
extern unsigned long long g_walk_steps[2];
#define DCRX_WALK_STEP(g) (g_walk_steps[g]++)
#else
#define DCRX_WALK_STEP(g) ((void)0)
#endif
// get_v_deletions — decombine.py:749-785
template <class FR>
DCRX_DEVNI bool get_v_deletions(const GeneDevPtrs &G, const FR &F, int v_match, int temp_end_v,
                                int &end_v, int &deletions_v, const Counters &C) {
  constexpr bool REV = FR::kRev;
  const int n = F.n();
  int f = temp_end_v;                                       // :753
  const int Lg = (int)G.reg_len[v_match];
  int pos = Lg - 10;                                        // :754-756
  if (f >= n) { C.add(DCRX_C_V_DEL_FAILED_TAG_AT_END); return false; }  // :760-762
  f += 1;                                                   // :764
  int num_del0 = 0;
  // Bit-parallel form of the loop below for the common geometry: the first 23 iterations
  // only touch read[f-32:f] and the last 32 germline bases, all whole 10-mers of pure ACGT.
  if (f >= 32 && f < n && G.w64_ok[v_match] && F.clean(f - 32, f)) {
    const uint64_t rw = REV ? F.load64(n - f) : F.load64(f - 32);
    const uint64_t y = mismatch_slots(rw, REV ? G.w64_rc[v_match] : G.w64_fwd[v_match]);
    const int k = REV ? first_clean_up(or10_up(y), 0) : first_clean_down(or10_down(y), 0);
    if (k >= 0) { deletions_v = k; end_v = temp_end_v - k; return true; }
    num_del0 = 23;                                          // the 23 steps that window stands for have failed
  }
  // Further windows of 32 read bases along the same diagonal (read position x faces region position
  // x + Lg - f): 23 steps each while a window fits between the read's start and the current position;
  // then one window aligned with the read's start for the steps that are left (only its slots below the
  // current position count).  An exception byte in a window is a mismatch (the region is pure ACGT).  What
  // remains for the loop below are the steps whose slices leave the read or the region (Python's slice
  // rules decide those) — a walk that runs far costs a few loads, not one round of loads per step.
  if (FR::kWindowedWalks && f < n && Lg >= 32 && n >= 32 && G.reg_clean[v_match]) {
    const uint32_t *gp = (REV ? G.reg_pk_rc : G.reg_pk) + G.reg_pk_off[v_match];
    int d0 = num_del0;
    for (;;) {
      const int ft = f - d0;                               // the 10-mer of step d0 ends here
      if (ft < 10 || Lg - d0 - 10 < 0) break;
      const int we = ft >= 32 ? ft : 32;                   // window: frame positions [we - 32, we)
      const int k0 = we - ft;
      // region bases facing the window: [we - 32 + Lg - f, we + Lg - f), clamped into the region and shifted
      uint64_t gw;
      if (!REV) {
        const int base = we - 32 + Lg - f;
        if (base < 0) break;                               // the region's start: the loop below
        const int b = base < Lg - 32 ? base : Lg - 32;
        gw = packed_window64(gp, b) >> (2 * (base - b));
      } else {
        const int base = f - we;                           // in the reverse-complemented region
        if (base + 32 > Lg) break;
        const int b = base > 0 ? base : 0;
        gw = packed_window64(gp, b) << (2 * (b - base));
      }
      const int sb = REV ? n - we : we - 32;               // the window in the stored read
      const uint64_t y2 = mismatch_slots(F.load64(sb), gw) | (F.has_exc() ? F.exc_slots(sb) : 0ull);
      const int k2 = REV ? first_clean_up(or10_up(y2), k0) : first_clean_down(or10_down(y2), k0);
      if (k2 >= 0) { deletions_v = d0 + (k2 - k0); end_v = temp_end_v - deletions_v; return true; }
      d0 += 23 - k0;
      if (k0 > 0) break;                                   // that was the window at the read's start
    }
    num_del0 = d0;
  }
  int num_del = num_del0;                                   // :765 (the steps the windows above have settled are not repeated)
  pos -= num_del0; f -= num_del0;
  while (0 <= f && f < n) {                                 // :767
    DCRX_WALK_STEP(0);
    if (slice_eq(G, v_match, pos, pos + 10, F, f - 10, f)) {  // :769-772
      deletions_v = num_del;                                // :774
      end_v = temp_end_v - num_del;                         // :775
      return true;
    }
    pos -= 1; num_del += 1; f -= 1;                         // :777-779
  }
  C.add(DCRX_C_V_DEL_FAILED);                               // :784
  return false;
}

// get_j_deletions — decombine.py:788-817
template <class FR>
DCRX_DEVNI bool get_j_deletions(const GeneDevPtrs &G, const FR &F, int j_match, int temp_start_j,
                                int end_of_v, int &start_j, int &deletions_j, const Counters &C) {
  constexpr bool REV = FR::kRev;
  const int n = F.n();
  int f = temp_start_j;                                     // :792
  int pos = 0;                                              // :793
  int p0 = 0;                                               // steps settled by the windows below
  // Bit-parallel form of the loop below while it stays inside read[ts:ts+32] and the first
  // 32 germline bases: skipped steps (f < end_of_v, :798-800) only raise the first position tried.
  if (f >= 0 && f + 32 <= n && G.w64_ok[j_match] && F.clean(f, f + 32)) {
    const int k0 = end_of_v > f ? end_of_v - f : 0;
    const uint64_t rw = REV ? F.load64(n - f - 32) : F.load64(f);
    const uint64_t y = mismatch_slots(rw, REV ? G.w64_rc[j_match] : G.w64_fwd[j_match]);
    const int k = REV ? first_clean_down(or10_down(y), k0) : first_clean_up(or10_up(y), k0);
    if (k >= 0) { deletions_j = k; start_j = f + k; return true; }
    p0 = 23;                                                // the 23 steps that window stands for have failed (or were skipped)
  }
  // Further windows, as in get_v_deletions: read position x faces region position x - temp_start_j.  Only steps
  // whose 10-mers lie inside the read and the region are settled here (a window that would pass the read's or
  // the region's end is aligned with that end, and only its steps from the current one on count).
  const int Lg = (int)G.reg_len[j_match];
  if (FR::kWindowedWalks && f >= 0 && Lg >= 32 && n >= 32 && G.reg_clean[j_match]) {
    const uint32_t *gp = (REV ? G.reg_pk_rc : G.reg_pk) + G.reg_pk_off[j_match];
    const int last = (n - 10 - f < Lg - 10) ? n - 10 - f : Lg - 10;        // the last step with both 10-mers whole
    while (p0 <= last) {
      int ws = f + p0;                                     // window: frame positions [ws, ws + 32)
      if (ws + 32 > n) ws = n - 32;
      if (ws - f + 32 > Lg) ws = f + Lg - 32;              // ... and region positions [ws - f, ws - f + 32)
      if (ws < 0 || ws + 32 > n) break;
      const int k0a = f + p0 - ws;                         // the window's step of p0
      const int k0b = end_of_v > ws ? end_of_v - ws : 0;   // steps before the end of V are passed over (:798-800)
      const int k0 = k0a > k0b ? k0a : k0b;
      const int k1 = last - (ws - f) < 22 ? last - (ws - f) : 22;
      const uint64_t gw = packed_window64(gp, REV ? Lg - 32 - (ws - f) : ws - f);
      const int sb = REV ? n - ws - 32 : ws;
      const uint64_t y2 = mismatch_slots(F.load64(sb), gw) | (F.has_exc() ? F.exc_slots(sb) : 0ull);
      const int k2 = REV ? first_clean_down_range(or10_down(y2), k0, k1) : first_clean_up_range(or10_up(y2), k0, k1);
      if (k2 >= 0) { deletions_j = (ws - f) + k2; start_j = ws + k2; return true; }
      p0 = (ws - f) + k1 + 1;
    }
  }
  pos = p0; f += p0;
  while (0 <= f + 2 && f + 2 < n) {                         // :795
    DCRX_WALK_STEP(1);
    if (pos >= Lg && f >= 0) break;                         // region[pos:pos+10] is empty from here on and read[f:f+10] never is (f + 2 < n): no step can succeed
    if (f < end_of_v) { pos += 1; f += 1; }                 // :798-800
    else if (slice_eq(G, j_match, pos, pos + 10, F, f, f + 10)) {  // :802-805
      deletions_j = pos; start_j = f;                       // :807-808
      return true;
    } else { pos += 1; f += 1; }                            // :810-811
  }
  C.add(DCRX_C_J_DEL_FAILED);                               // :816
  return false;
}

// Levenshtein.hamming(tag k, read[lo:hi]) <= 1 (decombine.py:308-317 and siblings);
// a length difference counts like rapidfuzz's padding.
template <class FR>
DCRX_DEVNI bool hamming_le1(const GeneDevPtrs &G, int k, const FR &F, int lo, int hi) {
  constexpr bool REV = FR::kRev;
  const int Lt = (int)G.tag_len[k];
  const int n = F.n();
  // packed form for a frame that can say where its exception bytes lie: the slice is the whole tag-long window; a 32-base
  // load placed inside the read covers it, and an exception byte is a mismatch (tags are pure ACGT) — no per-character
  // loop at the read's ends or next to an N
  if (FR::kWindowedWalks && hi - lo == Lt && n >= 32 && Lt <= 32) {
    const int b = REV ? n - lo - Lt : lo;
    const int bs = b < n - 32 ? b : n - 32;
    const int sh = 2 * (b - bs);
    const uint64_t mask = (Lt >= 32) ? ~0ull : ((1ull << (2 * Lt)) - 1ull);
    const uint64_t xm = F.has_exc() ? (F.exc_slots(bs) >> sh) : 0ull;
    const uint64_t y = (mismatch_slots(F.load64(bs) >> sh, REV ? G.tag_pk_rc[k] : G.tag_pk_fwd[k]) | xm) & mask;
    return dcrx_popc64(y) <= 1;
  }
  // packed form: the slice is the whole tag-long window, pure ACGT, and a 32-base load covers it
  if (hi - lo == Lt) {
    const int b = REV ? n - lo - Lt : lo;           // forward position of the window's first stored base
    if (b >= 0 && b + 32 <= n && F.clean(lo, hi)) {
      const uint64_t mask = (Lt >= 32) ? ~0ull : ((1ull << (2 * Lt)) - 1ull);
      const uint64_t y = mismatch_slots(F.load64(b), REV ? G.tag_pk_rc[k] : G.tag_pk_fwd[k]) & mask;
      return dcrx_popc64(y) <= 1;
    }
  }
  const uint8_t *t = G.tag_ascii + k * 32;
  int m = min(Lt, hi - lo);
  int d = max(Lt, hi - lo) - m;
  for (int x = 0; x < m && d <= 1; x++) d += (t[x] != F.chr(lo + x));
  return d <= 1;
}

struct XDat { int match, pos, dels, tagpos; };  // (v_match,end_v,v_dels,v_seq_start) / (j_match,start_j,j_dels,j_seq_end)

// One transition.  TABLE_LDS: `byte_addr` is the absolute LDS address of the entry (the staged
// table's rows carry their own LDS address, so the look-up needs no base add); otherwise a byte
// offset into the table in global memory.
template <bool TABLE_LDS>
DCRX_DEV uint32_t trans_at(const uint32_t *lds_trans, const DevTables &T, uint32_t byte_addr) {
#ifndef DCRX_HOST_EMUL
  if (TABLE_LDS) return *reinterpret_cast<const dcrx_lds_u32 *>(static_cast<uintptr_t>(byte_addr));
#endif
  return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T.trans) + byte_addr);
}

// ------------------------------------------------------------------------------
// Half-tag rescue: the `else` branches of vanalysis (:292-394) and janalysis
// (:420-531).  Re-scans the frame and, at every state where a keyword of class
// CLS ends, walks that keyword's candidate tags in ascending index order — the
// same order as iterating findall()'s list and the `indices` comprehension.
// GENE 0 = V, 1 = J; HALF 1 or 2 are run-time values so that the lanes of a wave can
// rescue different classes side by side.  Returns true with `out` filled on success;
// otherwise the caller bumps the "found half not other half" counter.
// ------------------------------------------------------------------------------
// Candidates of one half-tag hit (keyword `gk` = global keyword id of class (GENE, HALF), `hlen`
// long, starting at frame position p): its tags in ascending index — the `indices`
// comprehension of decombine.py:298-300 / :342-346 / :425-427 / :476-480.
template <class FR>
DCRX_DEVNI bool rescue_candidates(const DevTables &T, const FR &F, const int GENE, const int HALF, const uint32_t gk,
                                  const int hlen, const int p, const int end_of_v, XDat &out, const Counters &C) {
  const GeneDevPtrs &G = T.g[GENE];
  const int n = F.n();
  const int split = G.split;
  const int k0 = (int)T.kw_first[gk];                  // half_seqs.index(...)
  const int L0 = (int)G.tag_len[k0];
  for (uint32_t x = T.kw_begin[gk]; x < T.kw_begin[gk + 1]; x++) {
    const int k = (int)T.kw_tags[x];                   // indices, ascending
    const int Lk = (int)G.tag_len[k];
    const int q = (HALF == 1) ? p : p - split;         // window start
    int lo, hi;
    pyslice(n, q, q + L0, lo, hi);                     // :302-307 / :348-357 / :429-434 / :482-491
    if (Lk != hi - lo) continue;
    pyslice(n, q, (HALF == 1) ? p + Lk : p + Lk - split, lo, hi);  // :311-314 / :361-366
    if (!hamming_le1(G, k, F, lo, hi)) continue;
    if (GENE == 0) {
      C.add(HALF == 1 ? DCRX_C_VERR2 : DCRX_C_VERR1);  // :318 / :370
      const int te = (HALF == 1) ? p + G.jump[k] - 1 : p + G.jump[k] - split - 1;  // :320-322 / :372-377
      int end_v, dels;
      if (get_v_deletions(G, F, k, te, end_v, dels, C)) {
        out.match = k; out.pos = end_v; out.dels = dels; out.tagpos = q;  // :327-333 / :382-388
        return true;
      }
    } else {
      C.add(HALF == 1 ? DCRX_C_JERR2 : DCRX_C_JERR1);  // :445 / :504
      const int ts = (HALF == 1) ? p - G.jump[k] : p - G.jump[k] - split;  // :447-449 / :506-510
      const int jend = (HALF == 1) ? p + hlen + split : p + hlen;          // :450-454 / :511
      int start_j, dels;
      if (get_j_deletions(G, F, k, ts, end_of_v, start_j, dels, C)) {
        out.match = k; out.pos = start_j; out.dels = dels; out.tagpos = jend;  // :463-468 / :520-525
        return true;
      }
    }
  }
  return false;
}

// Candidates of one half-tag hit: every keyword of class (GENE, HALF) that ends at
// state `st` reached after frame position `i`, each with its tags in ascending index.
template <bool REV>
DCRX_DEVNI bool rescue_at(const DevTables &T, const Frame<REV> &F, const int GENE, const int HALF, const uint32_t st,
                          const int i, const int end_of_v, XDat &out, const Counters &C) {
  const int CLS = (GENE == 0) ? (HALF == 1 ? K_VH1 : K_VH2) : (HALF == 1 ? K_JH1 : K_JH2);
  for (uint32_t o = T.st_out[st - T.first_out];; o++) {
    const uint32_t ent = T.outs[o];
    if ((int)(ent & 7u) == CLS) {
      const int hlen = (int)((ent >> 3) & 63u);
      const uint32_t gk = T.kw_base[CLS] + ((ent >> 9) & 0xFFFFu);
      if (rescue_candidates(T, F, GENE, HALF, gk, hlen, i + 1 - hlen, end_of_v, out, C)) return true;   // p = hold_x[i][1]
    }
    if (ent >> 31) break;
  }
  return false;
}

// Rescue by re-scanning the frame (any read: exception bytes reset the machine).  Only taken
// when a hit list overflowed; walks the packed words like the scans do (one load per 16 bases).
template <bool REV, bool TABLE_LDS>
DCRX_DEVNI bool rescue(const DevTables &T, const uint32_t *lds_trans, const Frame<REV> &F, const int GENE,
                       const int HALF, int end_of_v, XDat &out, const Counters &C) {
  const int BIT = (GENE == 0) ? (HALF == 1 ? TE_VH1_BIT : TE_VH2_BIT) : (HALF == 1 ? TE_JH1_BIT : TE_JH2_BIT);
  const int n = F.n();
  if (n <= 0) return false;
  uint32_t e = T.row0;
  ExcCursor<REV> xc(F.r);
  const int top = (n - 1) >> 4;
  int i = 0;
  for (int wi = 0; wi <= top; wi++) {
    const int kk = REV ? top - wi : wi;                       // word index in scan order
    const int cnt = (kk == top) ? ((n - 1) & 15) + 1 : 16;    // bases it holds
    uint32_t wv = F.r.words[kk];
    if (cnt == 16 && xc.nextpos >= i + 16) {
      // a whole word without an unknown byte: sixteen look-ups with nothing but the OR of the entries; only a word inside which
      // a keyword of the class ends is walked base by base (round 6: the long form's reads meet this loop whenever a class has
      // more hits than its list holds — short half tags in 600 nt and more)
      const uint32_t wf = REV ? ~wv : wv;
      uint32_t ef = e, accw = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        ef = trans_at<TABLE_LDS>(lds_trans, T, (ef & TE_ROW_MASK) + (dcrx_ubfe(wf, REV ? 30 - 2 * j : 2 * j, 2) << 2));
        accw |= ef;
      }
      if (!((accw >> BIT) & 1u)) { e = ef; i += 16; continue; }
    }
    if (REV) wv = ~wv << (2 * (16 - cnt));                    // complement; first base of the frame on top
#pragma unroll 1
    for (int k = 0; k < cnt; k++, i++) {
      const uint32_t code = REV ? (wv >> 30) : (wv & 3u);
      wv = REV ? (wv << 2) : (wv >> 2);
      if (xc.hit(i)) { e = T.row0; continue; }                // unknown byte: machine back to the root
      e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + (code << 2));
      if (!((e >> BIT) & 1u)) continue;
      if (rescue_at<REV>(T, F, GENE, HALF, ((e & TE_ROW_MASK) - T.row0) >> 4, i, end_of_v, out, C)) return true;
    }
  }
  return false;
}

// Half-tag hits of one read, kept in LDS by the collecting scan of the queue kernel:
// per class (V half1, V half2, J half1, J half2) up to HH_K entries `state<<9 | end_pos<<23 | 1`
// in scan order, and the four hit counts (8 bits each, saturating at 255).
constexpr int HH_K = 4;
constexpr int HH_STRIDE = 4 * HH_K;  // dwords per lane
struct HalfHits {
  dcrx_lds_u32 *slot;
  uint32_t cnts;
  uint32_t keep = 0xFu;   // classes that are collected at all
  uint32_t compact = 0;   // 1: two lists only (V classes share list 0, J classes list 1; `keep` holds one class of each)
  uint32_t cap = 4;       // entries per list (HH_K; the long form's two compact lists hold 2 * HH_K each)
  uint32_t wide = 0;      // 1: entries `1 | state (relative to the first row) << 1 | end_pos << 15` — positions up to 65 535 (the long form)
  DCRX_DEV int list_of(int cls4) const { return compact ? (cls4 >> 1) : cls4; }
  DCRX_DEV int count(int cls4) const { return (int)((cnts >> (8 * cls4)) & 0xFFu); }
};

// Rescue from the collected hit list (same order as the re-scan would meet them).
template <bool REV>
DCRX_DEVNI bool rescue_list(const DevTables &T, const Frame<REV> &F, const HalfHits &hh, const int GENE,
                            const int HALF, int end_of_v, XDat &out, const Counters &C) {
  const int cls4 = GENE * 2 + (HALF - 1);
  const int cnt = hh.count(cls4);
  for (int h = 0; h < cnt; h++) {
    const uint32_t t = hh.slot[hh.list_of(cls4) * (int)hh.cap + h];
    const uint32_t st = hh.wide ? ((t >> 1) & 0x3FFFu) : ((t >> 9) & 0x3FFFu) - (T.row0 >> 4);
    if (rescue_at<REV>(T, F, GENE, HALF, st, hh.wide ? (int)(t >> 15) : (int)(t >> 23), end_of_v, out, C)) return true;
  }
  return false;
}

// ------------------------------------------------------------------------------
// The scan.  STEP consumes one symbol: one LDS look-up, flags OR-ed into `acc`,
// full-tag hits added into vacc / jacc.
// ------------------------------------------------------------------------------
struct ScanAcc { uint32_t acc, vacc, jacc; };   // raw accumulators of a scan

// What dcr_frame needs from a scan: the OR of all entry flags, and per full-tag class the hit
// count (0, 1, or >= 2) and, for a single hit, the state it ends in and its end position.
struct ScanOut {
  uint32_t acc;
  uint32_t vcount, jcount;
  uint32_t vstate, jstate;
  int vend, jend;
};

// One-base-per-step scans: the accumulators already hold state and position of a single hit.
DCRX_DEV ScanOut finish4(const DevTables &T, const ScanAcc &a) {
  ScanOut so;
  so.acc = a.acc;
  so.vcount = a.vacc & ACC_CNT_MASK;
  if ((a.acc >> TE_VMULTI_BIT) & 1u) so.vcount = 2;          // two tags ended at one position
  so.jcount = a.jacc & ACC_CNT_MASK;
  if ((a.acc >> TE_JMULTI_BIT) & 1u) so.jcount = 2;
  so.vstate = ((a.vacc >> ACC_STATE_SHIFT) & 0x3FFFu) - (T.row0 >> 4);
  so.jstate = ((a.jacc >> ACC_STATE_SHIFT) & 0x3FFFu) - (T.row0 >> 4);
  so.vend = (int)(a.vacc >> ACC_POS_SHIFT);
  so.jend = (int)(a.jacc >> ACC_POS_SHIFT);
  return so;
}

#define DCRX_STEP(CODE)                                                                         \
  do {                                                                                          \
    e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) | ((uint32_t)(CODE) << 2));         \
    acc |= e;                                                                                   \
    const uint32_t t_ = ((e & TE_ROW_MASK) << 5) | it;                                          \
    vacc += (uint32_t)dcrx_sbfe((int)e, TE_VFULL_BIT, 1) & t_;                                  \
    jacc += (uint32_t)dcrx_sbfe((int)e, TE_JFULL_BIT, 1) & t_;                                  \
    it += (1u << ACC_POS_SHIFT);                                                                \
  } while (0)

// Fast scan: no exceptions in this read.  w[] holds the read's first NW words (NW = 10 for
// strides up to 40 bytes, i.e. 150-nt reads; DCRX_NWMAX otherwise).
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV ScanAcc scan_fast(const DevTables &T, const uint32_t *lds_trans,
                                             const uint32_t (&w)[NW], const uint32_t *words, int n) {
  uint32_t e = T.row0, acc = 0, vacc = 0, jacc = 0, it = 1u;
  if (n > 0) {
    const int top = (n - 1) >> 4;           // index of the last (possibly partial) word
    const int cnt = ((n - 1) & 15) + 1;     // bases in it
    if (REV) {
      // the frame starts at the read's last base: partial word first, from its top base down
      uint32_t wp = ~words[top] << (2 * (16 - cnt));
      for (int k = 0; k < cnt; k++) { DCRX_STEP(wp >> 30); wp <<= 2; }
#pragma unroll
      for (int kk = NW - 1; kk >= 0; kk--) {
        if (kk < top) {
          const uint32_t wv = ~w[kk];
#pragma unroll
          for (int j = 15; j >= 0; j--) DCRX_STEP(dcrx_ubfe(wv, 2 * j, 2));
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NW; kk++) {
        if (kk < top) {
          const uint32_t wv = w[kk];
#pragma unroll
          for (int j = 0; j < 16; j++) DCRX_STEP(dcrx_ubfe(wv, 2 * j, 2));
        }
      }
      uint32_t wp = words[top];
      for (int k = 0; k < cnt; k++) { DCRX_STEP(wp & 3u); wp >>= 2; }
    }
  }
  return ScanAcc{acc, vacc, jacc};
}

// ------------------------------------------------------------------------------
// Two bases per step (fast kernel): one LDS look-up consumes a pair of bases.  The
// accumulators sum, per full-tag class, `1 | state BEFORE the pair << 9 | pair index << 23`
// over the pairs inside which a tag ends; finish16 turns a single hit into its exact end
// position and end state.
// ------------------------------------------------------------------------------
constexpr uint32_t TE16_ROW_MASK = 0x3FFFFu;

template <bool TABLE_LDS>
DCRX_DEV uint32_t trans16_at(const DevTables &T, uint32_t byte_addr) {
#ifndef DCRX_HOST_EMUL
  if (TABLE_LDS) return *reinterpret_cast<const dcrx_lds_u32 *>(static_cast<uintptr_t>(byte_addr));
#endif
  return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T.trans16) + byte_addr);
}

// Accumulator term of a pair = its look-up address | `it`: bits 2..5 the pair's bases (first * 4 +
// second), bits 6..17 the row of the state BEFORE the pair, bits 18..25 the pair index, bits
// 26..31 a count of one.  With exactly one hit the sum IS that hit.  A count that wraps (>= 64
// hits) is caught by the OR-ed VFULL/JFULL flag: flag seen but count 0 means "many".
constexpr int ACC16_PAIR_SHIFT = 18;
constexpr int ACC16_CNT_SHIFT = 26;
#define DCRX_ACC16_COUNT(x) ((x) >> dcrx::ACC16_CNT_SHIFT)

#define DCRX_STEP16(PAIR4) /* PAIR4 = (first base * 4 + second base) << 2 */                   \
  do {                                                                                          \
    const uint32_t addr_ = (e & TE16_ROW_MASK) | (uint32_t)(PAIR4);                             \
    e = trans16_at<TABLE_LDS>(T, addr_);                                                        \
    acc |= e;                                                                                   \
    const uint32_t t_ = addr_ | it;                                                             \
    vacc += (uint32_t)dcrx_sbfe((int)e, TE_VFULL_BIT, 1) & t_;                                  \
    jacc += (uint32_t)dcrx_sbfe((int)e, TE_JFULL_BIT, 1) & t_;                                  \
    it += (1u << ACC16_PAIR_SHIFT);                                                             \
  } while (0)

struct ScanAcc16 { uint32_t acc, vacc, jacc, e_last; };

// Pairs are (frame position 2k, 2k+1); an odd read length leaves one last base, which the
// caller finishes with a one-base step (finish16).
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV ScanAcc16 scan_fast16(const DevTables &T, const uint32_t (&w)[NW], const uint32_t *words, int n) {
  uint32_t e = T.row16_0, acc = 0, vacc = 0, jacc = 0, it = 1u << ACC16_CNT_SHIFT;
  const int npairs = n >> 1;
  if (npairs > 0) {
    // The frame's bases in scan order form a bit stream; it is consumed pair by pair.  REV: the
    // stream starts at the top base of the last word.  Whole words hold 8 pairs only when the
    // stream is word-aligned, i.e. when the partial top word holds an even number of bases.
    const int top = (n - 1) >> 4;
    const int cnt = ((n - 1) & 15) + 1;       // bases in the top word
    if (REV) {
      // top (partial) word first: cnt bases, from its top base down; complemented
      uint32_t wp = ~words[top] << (2 * (16 - cnt));
      const int pairs_top = cnt >> 1;
      for (int k = 0; k < pairs_top; k++) { DCRX_STEP16((wp >> 28) << 2); wp <<= 4; }
      if (cnt & 1) {
        // odd: the pair straddles words; generic path (one base from wp, one from the next word), keeps
        // straddling for every following word
        uint32_t carry = wp >> 30;            // last base of the top word (already complemented)
        for (int kk = top - 1; kk >= 0; kk--) {
          const uint32_t wv = ~words[kk];
          DCRX_STEP16(((carry << 2) | (wv >> 30)) << 2);
#pragma unroll
          for (int j = 14; j >= 2; j -= 2) DCRX_STEP16(dcrx_ubfe(wv, 2 * j - 2, 4) << 2);
          carry = wv & 3u;
        }
      } else {
#pragma unroll
        for (int kk = NW - 1; kk >= 0; kk--) {
          if (kk < top) {
            // nibble j, times four, is bits 2..5 of byte j/2 of the word shifted by two either way
            const uint32_t wv = ~w[kk], wlo = wv << 2, whi = wv >> 2;
            DCRX_STEP16(dcrx_byte_and<3>(whi, 0x3Cu)); DCRX_STEP16(dcrx_byte_and<3>(wlo, 0x3Cu));
            DCRX_STEP16(dcrx_byte_and<2>(whi, 0x3Cu)); DCRX_STEP16(dcrx_byte_and<2>(wlo, 0x3Cu));
            DCRX_STEP16(dcrx_byte_and<1>(whi, 0x3Cu)); DCRX_STEP16(dcrx_byte_and<1>(wlo, 0x3Cu));
            DCRX_STEP16(dcrx_byte_and<0>(whi, 0x3Cu)); DCRX_STEP16(dcrx_byte_and<0>(wlo, 0x3Cu));
          }
        }
      }
    } else {
      // forward: pairs never straddle words (a word holds 16 bases from an even frame position)
#pragma unroll
      for (int kk = 0; kk < NW; kk++) {
        if (kk < top) {
          const uint32_t wv = w[kk];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const uint32_t nib = dcrx_ubfe(wv, 4 * j, 4);           // second base in the high 2 bits
            DCRX_STEP16((((nib & 3u) << 2) | (nib >> 2)) << 2);
          }
        }
      }
      uint32_t wp = words[top];
      for (int k = 0; k < (cnt >> 1); k++) { DCRX_STEP16((((wp & 3u) << 2) | ((wp >> 2) & 3u)) << 2); wp >>= 4; }
    }
  }
  return ScanAcc16{acc, vacc, jacc, e};
}

// Resolves the accumulators of a two-bases-per-step scan.  A single hit is located by redoing
// its pair: the row of the state before the pair and the pair index are in the accumulator,
// the pair's bases come from the read, and the LDS entry says whether the tag ends at the
// second base (then its row is the end state) or at the first (one step in the one-base table
// in global memory gives the end state).  An odd read length leaves a last base, stepped here
// with the one-base table.
// `off`: frame position of the first base of pair 0 (1 when the caller consumed one base
// before the pairs, see scan_collect16); `leftover`: a last single base follows the pairs.
template <bool REV, bool TABLE_LDS>
DCRX_DEV ScanOut finish16(const DevTables &T, const Frame<REV> &F, const ScanAcc16 &a, const int off, const bool leftover) {
  ScanOut so;
  so.acc = DCRX_ACC16_FLAGS(a.acc) & ~((1u << TE16_V2_BIT) | (1u << TE16_J2_BIT));   // one-base flag layout
  so.vcount = DCRX_ACC16_COUNT(a.vacc);
  if (((a.acc >> TE_VMULTI_BIT) & 1u) || (so.vcount == 0 && ((a.acc >> TE_VFULL_BIT) & 1u))) so.vcount = 2;
  so.jcount = DCRX_ACC16_COUNT(a.jacc);
  if (((a.acc >> TE_JMULTI_BIT) & 1u) || (so.jcount == 0 && ((a.acc >> TE_JFULL_BIT) & 1u))) so.jcount = 2;
  so.vstate = so.jstate = 0; so.vend = so.jend = 0;
  const int n = F.n();
  // one-base step from state index st (new numbering) with base c, in the global one-base table
  auto step4 = [&](uint32_t st, int c) { return T.trans[st * 4 + (uint32_t)c]; };
  auto locate = [&](uint32_t accv, int full_bit, int second_bit, uint32_t &state, int &end) {
    const uint32_t rowp = accv & TE16_ROW_MASK & ~0x3Fu;             // row (address) of the state before the pair
    const int k = (int)((accv >> ACC16_PAIR_SHIFT) & 0xFFu);
    const int c1 = (int)((accv >> 4) & 3u);                          // the pair's bases ride in the accumulator
    const uint32_t e16 = trans16_at<TABLE_LDS>(T, accv & TE16_ROW_MASK);
    if ((e16 >> second_bit) & 1u) {
      state = ((e16 & TE16_ROW_MASK) - T.row16_0) >> 6; end = off + 2 * k + 1;
    } else {
      const uint32_t e1 = step4((rowp - T.row16_0) >> 6, c1);
      state = (e1 & TE_ROW_MASK) >> 4; end = off + 2 * k;
    }
    (void)full_bit;
  };
  bool vnew = false, jnew = false;
  if (leftover) {
    // the last base: one-base step from the state after the last pair
    const uint32_t slast = ((a.e_last & TE16_ROW_MASK) - T.row16_0) >> 6;
    const uint32_t e = step4(slast, F.code(n - 1));
    so.acc |= e & ~TE_ROW_MASK;
    if ((e >> TE_VMULTI_BIT) & 1u) so.vcount = 2;
    if ((e >> TE_JMULTI_BIT) & 1u) so.jcount = 2;
    if ((e >> TE_VFULL_BIT) & 1u) {
      if (so.vcount == 0) { so.vcount = 1; so.vstate = (e & TE_ROW_MASK) >> 4; so.vend = n - 1; vnew = true; }
      else if (so.vcount == 1) so.vcount = 2;
    }
    if ((e >> TE_JFULL_BIT) & 1u) {
      if (so.jcount == 0) { so.jcount = 1; so.jstate = (e & TE_ROW_MASK) >> 4; so.jend = n - 1; jnew = true; }
      else if (so.jcount == 1) so.jcount = 2;
    }
  }
  if (so.vcount == 1 && !vnew) locate(a.vacc, TE_VFULL_BIT, TE16_V2_BIT, so.vstate, so.vend);
  if (so.jcount == 1 && !jnew) locate(a.jacc, TE_JFULL_BIT, TE16_J2_BIT, so.jstate, so.jend);
  return so;
}

template <bool REV, bool TABLE_LDS>
DCRX_DEV ScanOut finish16(const DevTables &T, const Frame<REV> &F, const ScanAcc16 &a) {
  return finish16<REV, TABLE_LDS>(T, F, a, 0, (F.n() & 1) != 0);
}

// Collecting scan (queue kernel): scan_fast that also appends every half-tag hit to the
// read's LDS lists.
DCRX_DEV void collect_hits(HalfHits &hh, uint32_t hb, uint32_t t) {
  hb &= hh.keep;
  while (hb) {
    const int c = dcrx_ctz32(hb);
    hb &= hb - 1u;
    const uint32_t idx = (hh.cnts >> (8 * c)) & 0xFFu;
    if (idx < hh.cap) hh.slot[hh.list_of(c) * (int)hh.cap + (int)idx] = t;
    if (idx < 255u) hh.cnts += 1u << (8 * c);
  }
}

#define DCRX_STEP_C(CODE)                                                                       \
  do {                                                                                          \
    if ((int)(it >> ACC_POS_SHIFT) == xc.nextpos) {                                      \
      e = T.row0; xc.advance();   /* a byte outside ACGT: the machine goes back to the root */ \
    } else {                                                                                    \
      e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + ((uint32_t)(CODE) << 2));       \
      acc |= e;                                                                                 \
      const uint32_t t_ = ((e & TE_ROW_MASK) << 5) | it;                                        \
      vacc += (uint32_t)dcrx_sbfe((int)e, TE_VFULL_BIT, 1) & t_;                                \
      jacc += (uint32_t)dcrx_sbfe((int)e, TE_JFULL_BIT, 1) & t_;                                \
      const uint32_t hb_ = (e >> TE_VH1_BIT) & 0xFu;                                            \
      if (hb_) collect_hits(hh, hb_, t_);                                                       \
    }                                                                                           \
    it += (1u << ACC_POS_SHIFT);                                                                \
  } while (0)

// Exception bytes are walked with an ExcCursor while scanning (a clean read never matches it).
template <bool REV, bool TABLE_LDS>
DCRX_DEV ScanOut scan_collect(const DevTables &T, const uint32_t *lds_trans, const ReadView &rv, HalfHits &hh) {
  const uint32_t *words = rv.words;
  const int n = rv.n;
  uint32_t e = T.row0, acc = 0, vacc = 0, jacc = 0, it = 1u;
  ExcCursor<REV> xc(rv);
  hh.cnts = 0;
  if (n > 0) {
    const int top = (n - 1) >> 4;
    const int cnt = ((n - 1) & 15) + 1;
    if (REV) {
      uint32_t wp = ~words[top] << (2 * (16 - cnt));
      for (int k = 0; k < cnt; k++) { DCRX_STEP_C(wp >> 30); wp <<= 2; }
      for (int kk = top - 1; kk >= 0; kk--) {
        const uint32_t wv = ~words[kk];
#pragma unroll
        for (int j = 15; j >= 0; j--) DCRX_STEP_C(dcrx_ubfe(wv, 2 * j, 2));
      }
    } else {
      for (int kk = 0; kk < top; kk++) {
        const uint32_t wv = words[kk];
#pragma unroll
        for (int j = 0; j < 16; j++) DCRX_STEP_C(dcrx_ubfe(wv, 2 * j, 2));
      }
      uint32_t wp = words[top];
      for (int k = 0; k < cnt; k++) { DCRX_STEP_C(wp & 3u); wp >>= 2; }
    }
  }
  return finish4(T, ScanAcc{acc, vacc, jacc});
}

// The four filters of dcr() (decombine.py:553-569) and the record of a decombined read (:572-581).
template <class FR>
DCRX_DEVNI int dcr_filters(const DevTables &T, const FR &F, const XDat &vdat, const XDat &jdat, const dcrx::CfgDev &cfg,
                           const Counters &C, dcrx_record_t &rec) {
  const int n = F.n();
  const GeneDevPtrs &GV = T.g[0];
  const GeneDevPtrs &GJ = T.g[1];
  if (F.has_exc() && !cfg.allow_ns) {
    int lo, hi;
    pyslice(n, vdat.tagpos, jdat.tagpos, lo, hi);            // "N" in read[vdat[3]:jdat[3]]
    if (F.has_N(lo, hi)) { C.add(DCRX_C_DCRFILTER_INTERTAGN); return DCRX_S_F_INTERTAG_N; }
  }
  if ((vdat.tagpos - jdat.tagpos) >= cfg.lenthreshold) {     // :557-560
    C.add(DCRX_C_DCRFILTER_TOOLONG_INTERTAG); return DCRX_S_F_TOOLONG;
  }
  if (vdat.dels > (GV.jump[vdat.match] - (int)GV.tag_len[vdat.match]) || jdat.dels > GJ.jump[jdat.match]) {  // :561-565
    C.add(DCRX_C_DCRFILTER_IMPOSS_DELETION); return DCRX_S_F_IMPOSS_DEL;
  }
  if ((vdat.tagpos + (int)GV.tag_len[vdat.match]) > (jdat.tagpos + (int)GJ.tag_len[jdat.match])) {  // :566-569
    C.add(DCRX_C_DCRFILTER_TAG_OVERLAP); return DCRX_S_F_OVERLAP;
  }
  int lo, hi;
  pyslice(n, vdat.pos + 1, jdat.pos, lo, hi);                // read[vdat[1]+1 : jdat[1]] :577
  rec.v = (uint16_t)vdat.match; rec.j = (uint16_t)jdat.match;
  rec.v_start = (uint16_t)vdat.tagpos; rec.j_end = (uint16_t)jdat.tagpos;
  rec.ins_start = (uint16_t)lo; rec.ins_len = (uint16_t)(hi - lo);
  rec.vdel = (uint8_t)vdat.dels; rec.jdel = (uint8_t)jdat.dels;
  return DCRX_S_OK;
}

// ------------------------------------------------------------------------------
// dcr() for one frame — decombine.py:534-585 with vanalysis :273-394 and
// janalysis :397-531 folded around the single scan.  Returns the status and
// fills `rec` on DCRX_S_OK.
// ------------------------------------------------------------------------------
// DEFER: return DCRX_S_DEFER instead of entering a half-tag rescue (the fast kernel hands
// such reads to the queue kernel); a deferred read has touched no counter.
constexpr int DCRX_S_DEFER = 255;

template <bool REV, bool TABLE_LDS, bool DEFER>
DCRX_DEVNI int dcr_frame(const DevTables &T, const uint32_t *lds_trans, const ReadView &rv, const ScanOut &so,
                         const dcrx::CfgDev &cfg, const Counters &C, dcrx_record_t &rec,
                         const HalfHits *hh = nullptr) {
  const Frame<REV> F(rv);
  const int n = rv.n;
  const GeneDevPtrs &GV = T.g[0];
  const GeneDevPtrs &GJ = T.g[1];
  XDat vdat, jdat;

  // ---- vanalysis ---------------------------------------------------------------
  {
    const uint32_t vcount = so.vcount;
    if (vcount > 1) { C.add(DCRX_C_MULTIPLE_V_MATCHES); return DCRX_S_V_MULTI; }  // :278-280
    if (vcount == 1) {
      const uint32_t st = so.vstate;
      const int iend = so.vend;
      const int v = (int)(T.st_full[st - T.first_out] & 0xFFFFu);          // v_seqs.index(tag) :282
      const int p = iend + 1 - (int)GV.tag_len[v];           // hold_v[0][1]
      const int te = p + GV.jump[v] - 1;                     // :283-285
      int end_v, dels;
      if (!get_v_deletions(GV, F, v, te, end_v, dels, C))                      // :288-290
        return (te >= n) ? DCRX_S_V_WALK_FAIL_AT_END : DCRX_S_V_WALK_FAIL;         // :760-762 / :783-785
      vdat = XDat{v, end_v, dels, p};
    } else if ((so.acc >> TE_VH1_BIT) & 3u) {                // a V half1 (:294-335) or half2 (:339-390) keyword occurs
      if (DEFER) return DCRX_S_DEFER;
      const int half = ((so.acc >> TE_VH1_BIT) & 1u) ? 1 : 2;  // half2 is tried only when no half1 hit exists
      if (!((hh && hh->count(half - 1) <= (int)hh->cap) ? rescue_list<REV>(T, F, *hh, 0, half, 0, vdat, C)
               : rescue<REV, TABLE_LDS>(T, lds_trans, F, 0, half, 0, vdat, C))) {
        C.add(half == 1 ? DCRX_C_FOUNDV1NOTV2 : DCRX_C_FOUNDV2NOTV1);       // :334 / :389
        return half == 1 ? DCRX_S_V_HALF1_EXHAUSTED : DCRX_S_V_HALF2_EXHAUSTED;
      }
    } else {
      C.add(DCRX_C_NO_VTAGS_FOUND); return DCRX_S_V_NONE;    // :393-394
    }
  }
  const int end_of_v = vdat.pos + 1;                         // :547

  // ---- janalysis ---------------------------------------------------------------
  int jstatus = DCRX_S_OK;
  {
    const uint32_t jcount = so.jcount;
    if (jcount > 1) { C.add(DCRX_C_MULTIPLE_J_MATCHES); jstatus = DCRX_S_J_MULTI; }  // :402-404
    else if (jcount == 1) {
      const uint32_t st = so.jstate;
      const int iend = so.jend;
      const int j = (int)(T.st_full[st - T.first_out] >> 16);              // j_seqs.index(tag) :406
      const int Lj = (int)GJ.tag_len[j];
      const int p = iend + 1 - Lj;
      const int ts = p - GJ.jump[j];                         // :407-409
      int start_j, dels;
      if (get_j_deletions(GJ, F, j, ts, end_of_v, start_j, dels, C)) jdat = XDat{j, start_j, dels, p + Lj};  // :411-418
      else jstatus = DCRX_S_J_WALK_FAIL;
    } else if ((so.acc >> TE_JH1_BIT) & 3u) {                // a J half1 (:422-470) or half2 (:473-527) keyword occurs
      if (DEFER) return DCRX_S_DEFER;                        // nothing has been counted for this read yet
      const int half = ((so.acc >> TE_JH1_BIT) & 1u) ? 1 : 2;
      if (!((hh && hh->count(2 + half - 1) <= (int)hh->cap) ? rescue_list<REV>(T, F, *hh, 1, half, end_of_v, jdat, C)
               : rescue<REV, TABLE_LDS>(T, lds_trans, F, 1, half, end_of_v, jdat, C))) {
        C.add(half == 1 ? DCRX_C_FOUNDJ1NOTJ2 : DCRX_C_FOUNDV2NOTV1);       // :469 / :526 (the reference bumps the V key)
        jstatus = half == 1 ? DCRX_S_J_HALF1_EXHAUSTED : DCRX_S_J_HALF2_EXHAUSTED;
      }
    } else {
      C.add(DCRX_C_NO_J_ASSIGNED); jstatus = DCRX_S_J_NONE;  // :530-531
    }
  }
  if (jstatus != DCRX_S_OK) { C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); return jstatus; }  // :583-585

  return dcr_filters(T, F, vdat, jdat, cfg, C, rec);
}

// ------------------------------------------------------------------------------
// One read through the orientation dispatch of the reference's read loop
// (decombine.py:991, :998-1013) and out as a 16-byte record, in two forms.
//
// Fast-kernel form: clean reads (no exception bytes), one frame, no half-tag rescue.  Returns FAST_DONE, or — having touched neither counters nor
// the record — FAST_TO_RESCUE (a half-tag rescue is needed: rescue kernel) or
// FAST_TO_GENERAL (exception bytes, orientation `both`: general kernel).
// ------------------------------------------------------------------------------
enum { FAST_DONE = 0, FAST_TO_RESCUE = 1, FAST_TO_GENERAL = 2 };

template <bool TABLE_LDS, bool UNIFORM_LEN, int NW, int ARITY>
DCRX_DEV int decombine_fast_one(const DevTables &T, const uint32_t *lds_trans, const BatchDev &B,
                                 const CfgDev &cfg, uint64_t r, uint32_t nw, const Counters &C,
                                 dcrx_record_t *records) {
  if (cfg.orientation == DCRX_ORIENT_BOTH || (cfg.flags & DCRX_F_FORCE_SLOW_READER)) return FAST_TO_GENERAL;
  if (B.n_exc && ((B.exc_flag[r >> 5] >> (r & 31)) & 1u)) return FAST_TO_GENERAL;
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = rv.e1 = 0;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  uint32_t w[NW];
  {
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(rv.words);
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
      uint2 t = make_uint2(0u, 0u);
      if ((uint32_t)(2 * k) < nw) t = wp2[k];
      w[2 * k] = t.x; w[2 * k + 1] = t.y;
    }
  }
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  int status, frame;
  ScanOut so;
  const bool fwd = cfg.orientation == DCRX_ORIENT_FORWARD;           // decombine.py:1002-1004, else :999-1001
  if (ARITY == 16) {
    const ScanAcc16 a = fwd ? scan_fast16<false, TABLE_LDS, NW>(T, w, rv.words, rv.n)
                            : scan_fast16<true, TABLE_LDS, NW>(T, w, rv.words, rv.n);
    if (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) {  // profiling aid: price the scan alone
      rec.v = (uint16_t)a.acc; rec.j = (uint16_t)(a.acc >> 16); rec.v_start = (uint16_t)a.vacc; rec.j_end = (uint16_t)a.jacc;
      rec.status = 254; rec.frame = 0;
      dcrx_store_record(records + r, rec);
      return FAST_DONE;
    }
    so = fwd ? finish16<false, TABLE_LDS>(T, Frame<false>(rv), a) : finish16<true, TABLE_LDS>(T, Frame<true>(rv), a);
  } else {
    const ScanAcc a = fwd ? scan_fast<false, TABLE_LDS, NW>(T, lds_trans, w, rv.words, rv.n)
                          : scan_fast<true, TABLE_LDS, NW>(T, lds_trans, w, rv.words, rv.n);
    if (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) {
      rec.v = (uint16_t)a.acc; rec.j = (uint16_t)(a.acc >> 16); rec.v_start = (uint16_t)a.vacc; rec.j_end = (uint16_t)a.jacc;
      rec.status = 254; rec.frame = 0;
      dcrx_store_record(records + r, rec);
      return FAST_DONE;
    }
    so = finish4(T, a);
  }
  if (fwd) { status = dcr_frame<false, TABLE_LDS, true>(T, lds_trans, rv, so, cfg, C, rec); frame = 1; }
  else { status = dcr_frame<true, TABLE_LDS, true>(T, lds_trans, rv, so, cfg, C, rec); frame = 0; }
  if (status == DCRX_S_DEFER) return FAST_TO_RESCUE;
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
  return FAST_DONE;
}

// ------------------------------------------------------------------------------
// Pair-scan fast kernel, split in two so that the tail can run on full waves:
//   fast16_scan_one  loads the read into registers and scans it (every lane of a tile)
//   fast16_tail_one  locates the hits, walks, filters, writes the record — for the reads
//                    that have exactly one V tag (the others exit dcr_frame at once)
// Between the two the kernel ballot-compacts the V-hit reads of successive tiles into a
// per-wave LDS buffer of packed accumulators (TailEntry) and runs the tail 64 at a time.
// ------------------------------------------------------------------------------
struct TailEntry { uint32_t r, a, b; };   // read index | vacc(26)+flags(6) | jacc(26)+flags(4)

DCRX_DEV TailEntry tail_pack(uint32_t r, const ScanAcc16 &s) {      // only reads with exactly one V tag are packed
  TailEntry t;
  t.r = r;
  t.a = (s.vacc & 0x3FFFFFFu) | (((DCRX_ACC16_FLAGS(s.acc) >> TE_VFULL_BIT) & 0x3Fu) << 26);   // VFULL JFULL VH1 VH2 JH1 JH2 (either base)
  uint32_t jc = DCRX_ACC16_COUNT(s.jacc);                           // 0, 1, or "several" (a wrapped count shows as JFULL with 0)
  if (jc == 0 && ((s.acc >> TE_JFULL_BIT) & 1u)) jc = 2;
  jc = jc < 2u ? jc : 2u;
  t.b = (s.jacc & 0x3FFFFFFu) | (((s.acc >> TE_VMULTI_BIT) & 0xFu) << 26) | (jc << 30);       // VMULTI JMULTI V2 J2, J count
  return t;
}
DCRX_DEV ScanAcc16 tail_unpack(const TailEntry &t, uint32_t row16_0) {
  ScanAcc16 s;
  s.vacc = (t.a & 0x3FFFFFFu) | (1u << ACC16_CNT_SHIFT);
  s.jacc = (t.b & 0x3FFFFFFu) | ((t.b >> 30) << ACC16_CNT_SHIFT);
  s.acc = ((t.a >> 26) << TE_VFULL_BIT) | (((t.b >> 26) & 0xFu) << TE_VMULTI_BIT);
  s.e_last = row16_0;   // only even read lengths are batched: no last single base
  return s;
}

template <bool TABLE_LDS, bool UNIFORM_LEN, int NW>
DCRX_DEV ScanAcc16 fast16_scan_one(const DevTables &T, const BatchDev &B, const CfgDev &cfg, uint64_t r, uint32_t nw) {
  const uint32_t *words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  const int n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  uint32_t w[NW];
  {
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(words);
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
      uint2 t = make_uint2(0u, 0u);
      if ((uint32_t)(2 * k) < nw) t = wp2[k];
      w[2 * k] = t.x; w[2 * k + 1] = t.y;
    }
  }
  return (cfg.orientation == DCRX_ORIENT_FORWARD) ? scan_fast16<false, TABLE_LDS, NW>(T, w, words, n)
                                                  : scan_fast16<true, TABLE_LDS, NW>(T, w, words, n);
}

template <bool TABLE_LDS, bool UNIFORM_LEN>
DCRX_DEV int fast16_tail_one(const DevTables &T, const uint32_t *lds_trans, const BatchDev &B, const CfgDev &cfg,
                             uint64_t r, const ScanAcc16 &a, const Counters &C, dcrx_record_t *records) {
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = rv.e1 = 0;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  if (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) {  // profiling aid: price the scan alone
    rec.v = (uint16_t)a.acc; rec.j = (uint16_t)(a.acc >> 16); rec.v_start = (uint16_t)a.vacc; rec.j_end = (uint16_t)a.jacc;
    rec.status = 254; rec.frame = 0;
    dcrx_store_record(records + r, rec);
    return FAST_DONE;
  }
  int status, frame;
  if (cfg.orientation == DCRX_ORIENT_FORWARD) {                       // decombine.py:1002-1004
    const ScanOut so = finish16<false, TABLE_LDS>(T, Frame<false>(rv), a);
    status = dcr_frame<false, TABLE_LDS, true>(T, lds_trans, rv, so, cfg, C, rec); frame = 1;
  } else {                                                            // :999-1001
    const ScanOut so = finish16<true, TABLE_LDS>(T, Frame<true>(rv), a);
    status = dcr_frame<true, TABLE_LDS, true>(T, lds_trans, rv, so, cfg, C, rec); frame = 0;
  }
  if (status == DCRX_S_DEFER) return FAST_TO_RESCUE;
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
  return FAST_DONE;
}

// ------------------------------------------------------------------------------
// Rescue form (rescue kernel): clean reads the fast kernel deferred because a V or J tag
// needs the half-tag rescue.  The half-tag hit lists that scan_collect gathers base by
// base are produced here with the pair table:
//   1. scan_collect16: the pair scan of the fast kernel, which also marks, one bit per pair,
//      the pairs where some half-tag keyword ends (entry bits TE_VH1_BIT..TE_JH2_BIT);
//   2. resolve_half_hits: for each marked pair, in scan order, both of its positions are
//      resolved by re-running the automaton from the root over the W bases that end there
//      (W = the longest half tag + 1, rounded up to even; at most DCRX_MINI_W = 16).  The state so reached carries the same half-tag outputs as the state of the
//      full scan: every half-tag keyword that ends at a position is a suffix of that window
//      (W > max_half_len), hence a suffix of the longest window suffix that is a
//      trie prefix, and any deeper state of the full scan is longer than every half tag.
// The lists come out exactly as scan_collect writes them (same entries, same order).
// ------------------------------------------------------------------------------
constexpr int DCRX_MINI_W = 16;    // bases per window: one packed word

template <int NW>
struct PairMarks {
  uint32_t m[(NW + 3) / 4];   // byte (kk & 3) of m[kk >> 2]: bit j = pair in nibble j of read word kk (whole words below the top one)
  uint32_t top;               // the same for the partial top word
};

#define DCRX_STEP16C(PAIR4, MARKS, SHIFT)                                                       \
  do {                                                                                          \
    DCRX_STEP16(PAIR4);                                                                         \
    uint32_t h_ = e & hmask;                                                                    \
    h_ = h_ < 1u ? h_ : 1u;                                                                     \
    (MARKS) |= h_ << (SHIFT);                                                                   \
  } while (0)

// `off` (out): 1 when the frame's first base was consumed alone before the pairs (REV frame of an
// odd-length read: that keeps the pairs aligned with the nibbles of the packed words), else 0.
// An odd-length FWD read leaves its last base to finish16, like scan_fast16.
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV ScanAcc16 scan_collect16(const DevTables &T, const uint32_t (&w)[NW], const uint32_t *words, int n,
                                  uint32_t hmask, PairMarks<NW> &pm, int &off) {   // hmask: the half-tag class bits (both bases) that count
  uint32_t e = T.row16_0, acc = 0, vacc = 0, jacc = 0, it = 1u << ACC16_CNT_SHIFT;
#pragma unroll
  for (int x = 0; x < (NW + 3) / 4; x++) pm.m[x] = 0;
  pm.top = 0;
  off = 0;
  if (n > 0) {
    const int top = (n - 1) >> 4;
    const int cnt = ((n - 1) & 15) + 1;       // bases in the top word
    if (REV) {
      uint32_t wp = ~words[top] << (2 * (16 - cnt));
      int c = cnt;
      if (cnt & 1) {
        // frame position 0 alone, with the one-base table (global memory; keywords are >= 2 long, so nothing ends here)
        const uint32_t e4 = T.trans[wp >> 30];
        e = T.row16_0 + ((e4 & TE_ROW_MASK) >> 4) * 64u;
        wp <<= 2; c = cnt - 1; off = 1;
      }
      for (int k = 0; k < (c >> 1); k++) { DCRX_STEP16C((wp >> 28) << 2, pm.top, (c >> 1) - 1 - k); wp <<= 4; }
#pragma unroll
      for (int kk = NW - 1; kk >= 0; kk--) {
        if (kk < top) {
          const uint32_t wv = ~w[kk];
#pragma unroll
          for (int j = 7; j >= 0; j--) DCRX_STEP16C(dcrx_ubfe(wv, 4 * j, 4) << 2, pm.m[kk >> 2], 8 * (kk & 3) + j);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < NW; kk++) {
        if (kk < top) {
          const uint32_t wv = w[kk];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const uint32_t nib = dcrx_ubfe(wv, 4 * j, 4);
            DCRX_STEP16C((((nib & 3u) << 2) | (nib >> 2)) << 2, pm.m[kk >> 2], 8 * (kk & 3) + j);
          }
        }
      }
      uint32_t wp = words[top];
      for (int k = 0; k < (cnt >> 1); k++) {
        DCRX_STEP16C((((wp & 3u) << 2) | ((wp >> 2) & 3u)) << 2, pm.top, k);
        wp >>= 4;
      }
    }
  }
  return ScanAcc16{acc, vacc, jacc, e};
}

// The automaton's state after frame position `end`, and the half-tag classes (bit c = class
// K_VH1 + c) that hit there, from a scan of the last DCRX_MINI_W bases only.
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV void mini_scan(const DevTables &T, const ReadView &rv, const uint32_t (&w)[NW], int end, uint32_t &state,
                        uint32_t &classes, uint32_t &classes_before) {   // classes_before: those that end at end - 1 (0 when end == 0)
  const int n = rv.n;
  // even, <= DCRX_MINI_W (pair_rescue), and one base longer than every half tag: the classes reported for
  // end - 1 come from the window without its last base
  const int W = (int)((T.max_half_len + 2u) & ~1u);
  const int len = end + 1 < W ? end + 1 : W;
  // the window's bases, oldest first, 2 bits each (frame codes)
  const int p0 = REV ? n - 1 - end : end - len + 1;        // lowest read position of the window
  const int wi = p0 >> 4, sh = 2 * (p0 & 15);
  // the read's words sit in registers: a select chain instead of a (dependent, global) load
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int x = 0; x < NW; x++) { lo = (wi == x) ? w[x] : lo; hi = (wi + 1 == x) ? w[x] : hi; }
  uint32_t fv = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
  if (REV) {
    uint32_t x = dcrx_brev32(~fv);                                  // groups reversed, bits inside a group swapped
    x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);        // bits inside each group back in order
    fv = x >> (2 * (DCRX_MINI_W - len));
  }
  uint32_t st = 0;   // root
  int k = 0;
  uint32_t cls = 0;
  if (len & 1) {     // an odd window starts at frame position 0: its first base goes through the one-base table
    const uint32_t e4 = T.trans[fv & 3u];
    st = (e4 & TE_ROW_MASK) >> 4;
    cls = (e4 >> TE_VH1_BIT) & 0xFu;
    fv >>= 2; k = 1;
  }
  uint32_t e = T.row16_0 + st * 64u;
  for (; k < len; k += 2) {
    const uint32_t nib = fv & 15u;
    e = trans16_at<TABLE_LDS>(T, (e & TE16_ROW_MASK) | ((((nib & 3u) << 2) | (nib >> 2)) << 2));
    cls = e >> TE16_H2_SHIFT;
    fv >>= 4;
  }
  classes_before = len > 1 ? (e >> TE_VH1_BIT) & 0xFu : 0u;
  if (len > 1) st = ((e & TE16_ROW_MASK) - T.row16_0) >> 6;
  state = st; classes = cls;
}

template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV void resolve_pair(const DevTables &T, const ReadView &rv, const uint32_t (&w)[NW], int kk, int j, HalfHits &hh) {
  const int pa = 16 * kk + 2 * j;                              // read positions pa, pa + 1
  const int i1 = REV ? rv.n - 2 - pa : pa;                     // frame positions i1, i1 + 1
  // one window scan ending at the second base tells which (kept) classes end at either base; a
  // hit at the first base needs that position's own state: a second window scan, ending there
  uint32_t st2, c2, c1, st1, x0, x1;
  mini_scan<REV, TABLE_LDS, NW>(T, rv, w, i1 + 1, st2, c2, c1);
  c1 &= hh.keep; c2 &= hh.keep;
  if (c1) {
    mini_scan<REV, TABLE_LDS, NW>(T, rv, w, i1, st1, x0, x1);
    collect_hits(hh, c1, ((T.row0 + st1 * 16u) << 5) | ((uint32_t)i1 << ACC_POS_SHIFT) | 1u);
  }
  if (c2) collect_hits(hh, c2, ((T.row0 + st2 * 16u) << 5) | ((uint32_t)(i1 + 1) << ACC_POS_SHIFT) | 1u);
}

// Marked pairs in scan order.  `off`/REV as in scan_collect16.
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV void resolve_half_hits(const DevTables &T, const ReadView &rv, const uint32_t (&w)[NW], const PairMarks<NW> &pm,
                                HalfHits &hh) {
  const int n = rv.n;
  if (n <= 0) return;
  const int top = (n - 1) >> 4;
  auto word_marks = [&](int kk) {
    uint32_t v = 0;
#pragma unroll
    for (int x = 0; x < (NW + 3) / 4; x++) v = (kk >> 2) == x ? pm.m[x] : v;
    return (v >> (8 * (kk & 3))) & 0xFFu;
  };
  // One flat loop over the marked pairs in scan order (REV: top word first, nibbles downwards):
  // the wave's trip count is then the largest number of marked pairs of a lane, not the sum over
  // the words of the per-word maxima.
  auto marks_of = [&](int kk) { return kk == top ? pm.top : word_marks(kk); };
  int kk = REV ? top : 0;
  uint32_t b = marks_of(kk);
  for (;;) {
    while (!b) {
      kk += REV ? -1 : 1;
      if (kk < 0 || kk > top) return;
      b = marks_of(kk);
    }
    const int jj = REV ? 31 - dcrx_clz32(b) : dcrx_ctz32(b);
    b &= ~(1u << jj);
    resolve_pair<REV, TABLE_LDS, NW>(T, rv, w, kk, jj, hh);
  }
}

template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV int rescue16_frame(const DevTables &T, const ReadView &rv, const uint32_t (&w)[NW], const CfgDev &cfg,
                            const uint32_t hints, const Counters &C, dcrx_record_t &rec, HalfHits &hh) {
  // hints (from the fast kernel): bit 1 = the V tag needs the rescue, bit 0 = the J tag may need it
  const uint32_t genes = ((hints & 2u) ? 3u : 0u) | ((hints & 1u) ? 12u : 0u);          // as half-tag classes
  PairMarks<NW> pm;
  int off;
  const ScanAcc16 a = scan_collect16<REV, TABLE_LDS, NW>(T, w, rv.words, rv.n, (genes << TE_VH1_BIT) | (genes << TE16_H2_SHIFT),
                                                         pm, off);
  const Frame<REV> F(rv);
  const bool leftover = !REV && (rv.n & 1);
  const ScanOut so = finish16<REV, TABLE_LDS>(T, F, a, off, leftover);
  hh.cnts = 0;
  // dcr_frame tries half 1 of a gene when any half-1 keyword occurs, else half 2 (:294/:339,
  // :422/:473): only those two lists are kept, in two LDS lists per lane
  hh.keep = ((((so.acc >> TE_VH1_BIT) & 1u) ? 1u : 2u) | (((so.acc >> TE_JH1_BIT) & 1u) ? 4u : 8u)) & genes;
  hh.compact = 1;
  if (cfg.flags & DCRX_F_PROFILE_LIST_SCAN_ONLY) {   // profiling aid: price the marking scan alone
    rec.v = (uint16_t)so.acc; rec.j = (uint16_t)(so.vstate + so.jstate); rec.v_start = (uint16_t)(so.vend + so.jend);
    rec.j_end = (uint16_t)(pm.top + pm.m[0] + pm.m[1]); return 254;
  }
  resolve_half_hits<REV, TABLE_LDS, NW>(T, rv, w, pm, hh);
  if (cfg.flags & DCRX_F_PROFILE_RESCUE_HITS_ONLY) {   // profiling aid: ... and the hit resolution
    rec.v = (uint16_t)so.acc; rec.j = (uint16_t)(so.vstate + so.jstate); rec.v_start = (uint16_t)(so.vend + so.jend);
    rec.j_end = (uint16_t)hh.cnts; rec.ins_start = (uint16_t)hh.slot[0]; return 254;
  }
  if (leftover) {           // the last single base of an odd-length forward read
    uint32_t st, cls, before;
    mini_scan<REV, TABLE_LDS, NW>(T, rv, w, rv.n - 1, st, cls, before);
    if (cls) collect_hits(hh, cls, ((T.row0 + st * 16u) << 5) | ((uint32_t)(rv.n - 1) << ACC_POS_SHIFT) | 1u);
  }
  // TABLE_LDS false for dcr_frame: its re-scanning fallback (a class with more than HH_K hits)
  // reads the one-base table, which this kernel keeps in global memory
  return dcr_frame<REV, false, false>(T, nullptr, rv, so, cfg, C, rec, &hh);
}

// One clean read of the rescue queue; `slot`: this lane's 2 * HH_K dwords of LDS.
template <bool TABLE_LDS, bool UNIFORM_LEN, int NW>
DCRX_DEV void decombine_rescue16_one(const DevTables &T, const BatchDev &B, const CfgDev &cfg, uint64_t r, uint32_t hints,
                                     uint32_t nw, const Counters &C, dcrx_record_t *records, uint32_t *slot) {
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = rv.e1 = 0;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  uint32_t w[NW];
  {
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(rv.words);
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
      uint2 t = make_uint2(0u, 0u);
      if ((uint32_t)(2 * k) < nw) t = wp2[k];
      w[2 * k] = t.x; w[2 * k + 1] = t.y;
    }
  }
  HalfHits hh;
  hh.slot = DCRX_TO_LDS(slot);
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  int status, frame;
  if (cfg.orientation == DCRX_ORIENT_FORWARD) { status = rescue16_frame<false, TABLE_LDS, NW>(T, rv, w, cfg, hints, C, rec, hh); frame = 1; }
  else { status = rescue16_frame<true, TABLE_LDS, NW>(T, rv, w, cfg, hints, C, rec, hh); frame = 0; }
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
}

// ------------------------------------------------------------------------------
// General form inside the rescue kernel: a read with exception bytes (non-ACGT), one frame.
// Same results as scan_collect (one base per step, one-base table), but the state advances two
// bases per step through the pair table in LDS; the one-base table — in global memory in this
// kernel — is only consulted where it matters: a pair whose entry shows that some keyword ends
// inside it is replayed base by base (full accumulator / hit-list bookkeeping of DCRX_STEP_C),
// and a lone base before an exception byte or at the end of the read takes a single step.
// ------------------------------------------------------------------------------
template <bool REV, bool TABLE_LDS, int NW>
DCRX_DEV ScanOut scan_collect16_exc(const DevTables &T, const ReadView &rv, const uint32_t (&w)[NW], HalfHits &hh) {
  const int n = rv.n;
  uint32_t acc = 0, vacc = 0, jacc = 0;
  hh.cnts = 0;
  ExcCursor<REV> xc(rv);
  auto code = [&](int i) {           // frame code of position i, from the words in registers
    const int m = REV ? n - 1 - i : i;
    uint32_t word = 0;
#pragma unroll
    for (int x = 0; x < NW; x++) word = ((m >> 4) == x) ? w[x] : word;
    const uint32_t c = (word >> (2 * (m & 15))) & 3u;
    return REV ? (c ^ 3u) : c;
  };
  // one step of the one-base table from state `st` with base c at frame position i: DCRX_STEP_C's bookkeeping
  auto step1 = [&](uint32_t st, uint32_t c, int i) {
    const uint32_t e = T.trans[st * 4u + c];
    acc |= e;
    const uint32_t t_ = ((T.row0 + (e & TE_ROW_MASK)) << 5) | ((uint32_t)i << ACC_POS_SHIFT) | 1u;
    vacc += (uint32_t)dcrx_sbfe((int)e, TE_VFULL_BIT, 1) & t_;
    jacc += (uint32_t)dcrx_sbfe((int)e, TE_JFULL_BIT, 1) & t_;
    const uint32_t hb = (e >> TE_VH1_BIT) & 0xFu;
    if (hb) collect_hits(hh, hb, t_);
    return (e & TE_ROW_MASK) >> 4;    // the new state
  };
  constexpr uint32_t INTEREST = (1u << TE_VFULL_BIT) | (1u << TE_JFULL_BIT) | (0xFu << TE_VH1_BIT) | (0xFu << TE16_H2_SHIFT) |
                                (1u << TE_VMULTI_BIT) | (1u << TE_JMULTI_BIT);
  uint32_t st = 0;                    // root
  int i = 0;
  while (i < n) {
    if (i == xc.nextpos) { st = 0; xc.advance(); i++; continue; }   // a byte outside ACGT: back to the root
    if (i + 1 < n && i + 1 != xc.nextpos) {
      const uint32_t c1 = code(i), c2 = code(i + 1);
      const uint32_t e16 = trans16_at<TABLE_LDS>(T, (T.row16_0 + st * 64u) | ((c1 * 4u + c2) << 2));
      if (e16 & INTEREST) { st = step1(st, c1, i); st = step1(st, c2, i + 1); }
      else st = ((e16 & TE16_ROW_MASK) - T.row16_0) >> 6;
      i += 2;
    } else {
      st = step1(st, code(i), i);
      i += 1;
    }
  }
  return finish4(T, ScanAcc{acc, vacc, jacc});
}

// `e0`: index of the read's first entry in the exception list (from the prologue kernel).
// `slot`: DCRX_GENERAL_SLOT dwords of LDS for this lane: hit lists, exception copy, the read's words.
constexpr int DCRX_GENERAL_WORDS_AT = 24;                       // after HH_STRIDE + the exception copy
constexpr int DCRX_GENERAL_SLOT = DCRX_GENERAL_WORDS_AT + DCRX_NWMAX;
template <bool TABLE_LDS, bool UNIFORM_LEN, int NW>
DCRX_DEV void decombine_general16_one(const DevTables &T, const BatchDev &B, const CfgDev &cfg, uint64_t r, uint32_t e0,
                                      uint32_t nw, const Counters &C, dcrx_record_t *records, uint32_t *slot) {
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  uint64_t hi = e0;
  while (hi < B.n_exc && B.exc_read[hi] == (uint32_t)r) hi++;
  rv.e0 = (int)e0; rv.e1 = (int)hi;
  if (rv.e1 - rv.e0 <= DCRX_EXC_LDS) {
    uint16_t *xp = reinterpret_cast<uint16_t *>(slot + HH_STRIDE);
    uint8_t *xb = reinterpret_cast<uint8_t *>(slot + HH_STRIDE + DCRX_EXC_LDS / 2);
    const int cnt = rv.e1 - rv.e0;
    for (int x = 0; x < cnt; x++) { xp[x] = B.exc_pos[rv.e0 + x]; xb[x] = B.exc_chr[rv.e0 + x]; }
    rv.exc_pos = xp; rv.exc_chr = xb; rv.e0 = 0; rv.e1 = cnt;  // indices re-based onto the LDS copy
  }
  uint32_t w[NW];
  {
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(rv.words);
#pragma unroll
    for (int k = 0; k < NW / 2; k++) {
      uint2 t = make_uint2(0u, 0u);
      if ((uint32_t)(2 * k) < nw) t = wp2[k];
      w[2 * k] = t.x; w[2 * k + 1] = t.y;
    }
  }
  // the walks and comparisons of a read with exception bytes go character by character: they read
  // the packed words from this lane's LDS copy, not from global memory
  uint32_t *lw = slot + DCRX_GENERAL_WORDS_AT;
#pragma unroll
  for (int x = 0; x < NW; x++) lw[x] = w[x];
  rv.words = lw;
  HalfHits hh;
  hh.slot = DCRX_TO_LDS(slot);
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  int status, frame;
  if (cfg.orientation == DCRX_ORIENT_FORWARD) {
    const ScanOut so = scan_collect16_exc<false, TABLE_LDS, NW>(T, rv, w, hh);
    status = dcr_frame<false, false, false>(T, nullptr, rv, so, cfg, C, rec, &hh); frame = 1;
  } else {
    const ScanOut so = scan_collect16_exc<true, TABLE_LDS, NW>(T, rv, w, hh);
    status = dcr_frame<true, false, false>(T, nullptr, rv, so, cfg, C, rec, &hh); frame = 0;
  }
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
}

// ------------------------------------------------------------------------------
// List form (list kernel): any read — exception bytes, orientation `both` with its
// second, forward attempt (decombine.py:1005-1010), half-tag rescue.  One collecting
// scan per frame, then dcr_frame with the rescue fed from the LDS hit lists (a class
// with more than HH_K hits falls back to the re-scanning rescue inside dcr_frame).
// `slot`: this lane's LDS area (hit lists, then a copy of the read's first
// DCRX_EXC_LDS exception entries).
// ------------------------------------------------------------------------------
template <bool TABLE_LDS, bool UNIFORM_LEN>
DCRX_DEV void decombine_list_one(const DevTables &T, const uint32_t *lds_trans, const BatchDev &B,
                                 const CfgDev &cfg, uint64_t r, const Counters &C, dcrx_record_t *records,
                                 uint32_t *slot, const bool from_general) {
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = rv.e1 = 0;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  // A read of the general list may have exception bytes (its slice of the list is looked up: empty for a clean read that is
  // there because every read is — orientation `both`, the forced slow reader); a read of the rescue queue has none.  (Not the
  // workspace bitmap: behind the v2 kernels it is all zero again by the time a handed-over read arrives here.)
  if (from_general && B.n_exc) {
    // binary search of this read's slice in the sorted exception list
    uint64_t lo = 0, hi = B.n_exc;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (B.exc_read[mid] < (uint32_t)r) lo = mid + 1; else hi = mid; }
    rv.e0 = (int)lo;
    while (lo < B.n_exc && B.exc_read[lo] == (uint32_t)r) lo++;
    rv.e1 = (int)lo;
    if (rv.e1 - rv.e0 <= DCRX_EXC_LDS) {
      uint16_t *xp = reinterpret_cast<uint16_t *>(slot + HH_STRIDE);
      uint8_t *xb = reinterpret_cast<uint8_t *>(slot + HH_STRIDE + DCRX_EXC_LDS / 2);
      const int cnt = rv.e1 - rv.e0;
      for (int x = 0; x < cnt; x++) { xp[x] = B.exc_pos[rv.e0 + x]; xb[x] = B.exc_chr[rv.e0 + x]; }
      rv.exc_pos = xp; rv.exc_chr = xb; rv.e0 = 0; rv.e1 = cnt;  // indices re-based onto the LDS copy
    }
  }
  HalfHits hh;
  hh.slot = DCRX_TO_LDS(slot);
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  // Orientation dispatch of decombine.py:999-1010 as a loop, so that each frame's code exists
  // once: reverse (unless `forward`), then forward for `forward`, or for `both` after a miss.
  int status = DCRX_S_V_NONE, frame = 0;
  for (int attempt = (cfg.orientation == DCRX_ORIENT_FORWARD) ? 1 : 0; attempt < 2; attempt++) {
    if (attempt == 0) {
      const ScanOut so = scan_collect<true, TABLE_LDS>(T, lds_trans, rv, hh);
      if (cfg.flags & DCRX_F_PROFILE_LIST_SCAN_ONLY) {   // profiling aid: price the collecting scan alone
        rec.v = (uint16_t)so.acc; rec.j = (uint16_t)(so.vstate + so.jstate); rec.v_start = (uint16_t)(so.vend + so.jend);
        rec.j_end = (uint16_t)hh.cnts; status = 254; break;
      }
      status = dcr_frame<true, TABLE_LDS, false>(T, lds_trans, rv, so, cfg, C, rec, &hh); frame = 0;
      if (status == DCRX_S_OK || cfg.orientation != DCRX_ORIENT_BOTH) break;
    } else {
      const ScanOut so = scan_collect<false, TABLE_LDS>(T, lds_trans, rv, hh);
      status = dcr_frame<false, TABLE_LDS, false>(T, lds_trans, rv, so, cfg, C, rec, &hh); frame = 1;
    }
  }
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
}

// ------------------------------------------------------------------------------
// Reads beyond the register shapes and the packed hit lists (512 .. 65 535 nt: merged pairs, long amplicons — the
// reference has no length limit, decombine.py:228-265, :534-585): one read per lane in the plainest form there is.
// Per frame a scan over the packed words in memory with the one-base table — in LDS where the tables' image fits beside the
// block's counters and the lanes' slots (every BASELINE tag set's does: 28 KB of rows for config 2, 59 KB for the extended alpha
// set), else in global memory —: the OR of the entry flags, per full-tag class the count and the first hit's state and end, the
// half-tag hits in lists per class, all positions in plain integers (the accumulators and hit lists of the forms above pack a
// position into nine bits); then dcr_frame with the rescue from the lists.
// ------------------------------------------------------------------------------
// The long form's scan, in two passes over the words of the frame (round 6; scan_plain before: one pass, every base with its
// tests — 12 of the form's 62 vector instructions per base and lane — and dcr_frame's rescue by re-scanning the whole read, once
// per half-tag class, for every wave that held one such read: nearly all).
//   pass 1: word by word, sixteen unrolled look-ups with nothing but the OR of the entries (4 instructions per base); a word whose
//           OR shows a flag — some keyword ends inside it: two to four words of a rearranged read, none of the others — is noted
//           with the state in front of it (7 to 15 notes per lane in LDS).  The partial word and words that hold an exception
//           byte go base by base.
//   pass 2: the noted words again, base by base with scan_plain's bookkeeping and every half-tag hit into the read's hit lists
//           (HalfHits, wide entries, one list of 2 * HH_K per gene: note_hit): dcr_frame rescues from the lists and re-scans only a
//           class with more hits than its list holds.
//           Where a lane's notes could run out, every lane of the wave takes pass 2 over the notes it has and goes on with pass 1.
constexpr int LONG_FW_MIN = 7, LONG_FW_MAX = 15;      // notes per lane: as many as the block's LDS holds beside the tables (the launch's choice, an odd slot)
constexpr int DCRX_LONG_SLOT_MIN = (HH_STRIDE + LONG_FW_MIN) | 1, DCRX_LONG_SLOT_MAX = (HH_STRIDE + LONG_FW_MAX) | 1;      // dwords of LDS per lane (odd: the lanes' slots on different banks)
constexpr uint32_t TE_FLAGS_MASK = 0xFFu << TE_VFULL_BIT;

template <bool REV, bool TABLE_LDS>
struct LongScan {
  const DevTables &T;
  const uint32_t *lds_trans;
  const ReadView &rv;
  HalfHits &hh;
  ScanOut so;
  ExcCursor<REV> xc;
  int top, cnt_top;
  DCRX_DEV LongScan(const DevTables &T_, const uint32_t *lt, const ReadView &rv_, HalfHits &hh_) : T(T_), lds_trans(lt), rv(rv_), hh(hh_), xc(rv_) {
    so.acc = 0; so.vcount = so.jcount = 0; so.vstate = so.jstate = 0; so.vend = so.jend = 0;
    top = (rv.n - 1) >> 4; cnt_top = ((rv.n - 1) & 15) + 1;
  }
  DCRX_DEV int word_of(int wi) const { return REV ? top - wi : wi; }                      // word index of the wi-th word in scan order
  DCRX_DEV int first_pos(int wi) const { return REV ? (wi == 0 ? 0 : cnt_top + 16 * (wi - 1)) : 16 * wi; }
  // pass 2 of one word: from entry `e` in front of it; returns the entry behind it
  // what a flagged entry (reached behind frame position i) leaves in the scan's results
  DCRX_DEV void note_hit(const uint32_t e, const int i) {
    const uint32_t fl = e & ~TE_ROW_MASK;
    so.acc |= fl;
    const uint32_t st = ((e & TE_ROW_MASK) - T.row0) >> 4;
    if ((fl >> TE_VFULL_BIT) & 1u) { if (so.vcount == 0) { so.vstate = st; so.vend = i; } so.vcount++; }
    if ((fl >> TE_JFULL_BIT) & 1u) { if (so.jcount == 0) { so.jstate = st; so.jend = i; } so.jcount++; }
    if ((fl >> TE_VMULTI_BIT) & 1u) so.vcount = so.vcount < 2 ? 2 : so.vcount;      // two tags ended at one position
    if ((fl >> TE_JMULTI_BIT) & 1u) so.jcount = so.jcount < 2 ? 2 : so.jcount;
    // The hit lists hold what dcr_frame will ask for, whatever is still to come: per gene ONE list — no hit once a full tag
    // was seen, the half-1 hits from the first of them on, the half-2 hits as long as no half-1 keyword has occurred (the
    // reference consults the half-2 list only then, :294/:339, :422/:473) —, two lists of 2 * HH_K entries in the room of four
    // of HH_K.  A half-1 hit drops its gene's half-2 class: its own first entry then overwrites what the list held.
    // (Short half tags meet a long read by chance: with four entries a list nearly every wave held a lane that overflowed and
    // scanned its whole read again.)
    if ((fl >> TE_VFULL_BIT) & 1u) hh.keep &= ~3u;
    if ((fl >> TE_JFULL_BIT) & 1u) hh.keep &= ~12u;
    const uint32_t hb = (e >> TE_VH1_BIT) & 0xFu;
    if (hb & 1u) hh.keep &= ~2u;
    if (hb & 4u) hh.keep &= ~8u;
    if (hb) collect_hits(hh, hb, 1u | (st << 1) | ((uint32_t)i << 15));
  }
  // pass 2 of one word: from entry `e` in front of it; returns the entry behind it.  wv: the word, rv.words[word_of(wi)]
  DCRX_DEVNI uint32_t replay(const int wi, uint32_t e, uint32_t wv) {
    const int kk = word_of(wi);
    const int cnt = (kk == top) ? cnt_top : 16;
    int i = first_pos(wi);
    while (xc.nextpos < i) xc.advance();
    if (cnt == 16 && xc.nextpos >= i + 16) {
      // A whole word without an unknown byte, in two halves: eight entries first, as pass 1 met them, with a bit per flagged one, then
      // the flagged ones in turn.  (Base by base with the bookkeeping behind a test per base, the lanes of a wave — each at another
      // base of its word — took the bookkeeping one after the other: sixteen turns per word where a lane needs one or two.  Eight
      // entries and a select chain; sixteen and a select tree spilled 30 registers of a kernel at its 128 and were slower than the
      // loop: 600 nt 0.745 / 0.796 / 0.561 ms per 2 M reads for the loop / the tree of 16 / this, profiles/r06/long_form_two_pass_scan.log.)
      constexpr int GE = 8;
      const uint32_t wf = REV ? ~wv : wv;
#pragma unroll
      for (int hf = 0; hf < 16 / GE; hf++) {
        uint32_t ee[GE], m = 0;
#pragma unroll
        for (int j = 0; j < GE; j++) {
          const int jj = GE * hf + j;
          e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + (dcrx_ubfe(wf, REV ? 30 - 2 * jj : 2 * jj, 2) << 2));
          ee[j] = e;
          m |= ((e & ~TE_ROW_MASK) ? 1u : 0u) << j;
        }
        while (m) {
          const int j = dcrx_ctz32(m);
          m &= m - 1u;
          uint32_t sel = ee[0];
#pragma unroll
          for (int q = 1; q < GE; q++) sel = (j == q) ? ee[q] : sel;
          note_hit(sel, i + GE * hf + j);
        }
      }
      return e;
    }
    if (REV) wv = ~wv << (2 * (16 - cnt));                    // complement; first base of the frame on top
#pragma unroll 1
    for (int k = 0; k < cnt; k++, i++) {
      const uint32_t code = REV ? (wv >> 30) : (wv & 3u);
      wv = REV ? (wv << 2) : (wv >> 2);
      if (xc.hit(i)) { e = T.row0; continue; }                // unknown byte: machine back to the root
      e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + (code << 2));
      if (e & ~TE_ROW_MASK) note_hit(e, i);
    }
    return e;
  }
};

template <bool REV, bool TABLE_LDS = false>
DCRX_DEVNI ScanOut scan_long(const DevTables &T, const uint32_t *lds_trans, const ReadView &rv, HalfHits &hh, dcrx_lds_u32 *fw, const int fwk,
                             const bool pass1_only = false) {
  hh.cnts = 0; hh.wide = 1u;
  LongScan<REV, TABLE_LDS> L(T, lds_trans, rv, hh);
  const int n = rv.n;
  if (n <= 0) return L.so;
  const int top = L.top, cnt_top = L.cnt_top;
  // ---- pass 1 ----
  uint32_t e = T.row0;
  int nfw = 0;
  // pass 2 over the notes in hand (the next note's word is asked for before this one's bases are walked: a load per note,
  // each a trip to memory of its own, was a third of the pass)
  auto take_back = [&]() {
    if (nfw <= 0) return;
    uint32_t t = fw[0];
    uint32_t wv = rv.words[L.word_of((int)(t >> 14))];
    for (int k = 0; k < nfw; k++) {
      uint32_t tn = t, wn = wv;
      if (k + 1 < nfw) { tn = fw[k + 1]; wn = rv.words[L.word_of((int)(tn >> 14))]; }
      (void)L.replay((int)(t >> 14), T.row0 + ((t & 0x3FFFu) << 4), wv);
      t = tn; wv = wn;
    }
    nfw = 0;
  };
  hh.keep = 0xFu; hh.compact = 1u; hh.cap = 2u * (uint32_t)HH_K;      // (LongScan::note_hit narrows `keep` as the flags come in)
  {
    ExcCursor<REV> x1(rv);
    // The words come in chunks of sixteen (64 bytes, eight 8-byte loads at consecutive addresses: a read starts on an 8-byte boundary
    // and its stride is a multiple of 8) and are scanned out of registers, four at a time.
    constexpr int CW = 16;      // words fetched per trip and lane (8 measured level)
    const int ptop = top >> 1, ctop = top / CW;
    const uint2 *wp2 = reinterpret_cast<const uint2 *>(rv.words);
#pragma unroll 1
    for (int ci = 0; ci <= ctop; ci++) {
      const int cp = REV ? ctop - ci : ci;
      uint32_t cw[CW];
#pragma unroll
      for (int k = 0; k < CW / 2; k++) {
        uint2 t; t.x = 0u; t.y = 0u;
        if ((CW / 2) * cp + k <= ptop) t = wp2[(CW / 2) * cp + k];
        cw[2 * k] = t.x; cw[2 * k + 1] = t.y;
      }
#pragma unroll 1
      for (int g = 0; g < CW / 4; g++) {
        const int gg = REV ? CW / 4 - 1 - g : g;
        if (CW * cp + 4 * gg > top) continue;                    // (the whole group lies beyond the read)
        uint32_t w4[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
          w4[h] = cw[h];
#pragma unroll
          for (int q = 1; q < CW / 4; q++) w4[h] = gg == q ? cw[4 * q + h] : w4[h];
        }
        // The notes are taken back whenever a lane of the wave could run out of them inside the next four words — by every lane of the
        // wave at once: a lane that did so on its own would have the other sixty-three wait for it, each in its turn.
#ifndef DCRX_HOST_EMUL
        if (__ballot(nfw > fwk - 4))
#else
        if (nfw > fwk - 4)
#endif
        {
          if (pass1_only) nfw = 0;      // (profiling: the first pass alone)
          take_back();
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
          const int hw = REV ? 3 - h : h;
          const int kk = CW * cp + 4 * gg + hw;
          if (kk > top) continue;                                  // (beyond the read's last word)
          const int wi = REV ? top - kk : kk;
          const uint32_t word = w4[hw];
          const int cnt = (kk == top) ? cnt_top : 16;
          const int i0 = L.first_pos(wi);
          const uint32_t e0 = e;
          uint32_t accw = 0;
          if (cnt == 16 && x1.nextpos >= i0 + 16) {
            const uint32_t wv = REV ? ~word : word;
#pragma unroll
            for (int j = 0; j < 16; j++) {
              const uint32_t code = dcrx_ubfe(wv, REV ? 30 - 2 * j : 2 * j, 2);
              e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + (code << 2));
              accw |= e;
            }
          } else {
            uint32_t wv = word;
            if (REV) wv = ~wv << (2 * (16 - cnt));
            int i = i0;
#pragma unroll 1
            for (int k = 0; k < cnt; k++, i++) {
              const uint32_t code = REV ? (wv >> 30) : (wv & 3u);
              wv = REV ? (wv << 2) : (wv >> 2);
              if (i == x1.nextpos) { e = T.row0; x1.advance(); continue; }
              e = trans_at<TABLE_LDS>(lds_trans, T, (e & TE_ROW_MASK) + (code << 2));
              accw |= e;
            }
          }
          if (accw & TE_FLAGS_MASK) fw[nfw++] = ((uint32_t)wi << 14) | (((e0 & TE_ROW_MASK) - T.row0) >> 4);
        }
      }
    }
  }
  // ---- pass 2: the notes that are left ----
  if (pass1_only) { L.so.acc = e; return L.so; }
  take_back();
  return L.so;
}

// `slot`: slot_dwords (DCRX_LONG_SLOT_MIN .. _MAX) dwords of LDS for this lane (the hit lists, then the notes of pass 1)
template <bool UNIFORM_LEN, bool TABLE_LDS = false>
DCRX_DEV void decombine_long_one(const DevTables &T, const uint32_t *lds_trans, const BatchDev &B, const CfgDev &cfg, const uint64_t r,
                                 const Counters &C, dcrx_record_t *records, uint32_t *slot, const int slot_dwords) {
  const int fwk = slot_dwords - HH_STRIDE;
  ReadView rv;
  rv.comp = T.comp;
  rv.words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = rv.e1 = 0;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  if (B.n_exc) {      // this read's slice of the sorted exception list
    uint64_t lo = 0, hi = B.n_exc;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (B.exc_read[mid] < (uint32_t)r) lo = mid + 1; else hi = mid; }
    rv.e0 = (int)lo;
    while (lo < B.n_exc && B.exc_read[lo] == (uint32_t)r) lo++;
    rv.e1 = (int)lo;
  }
  HalfHits hh;
  hh.slot = DCRX_TO_LDS(slot);
  dcrx_lds_u32 *fw = DCRX_TO_LDS(slot) + HH_STRIDE;
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  // (the orientation dispatch of decombine.py:999-1010, as decombine_list_one has it)
  int status = DCRX_S_V_NONE, frame = 0;
  for (int attempt = (cfg.orientation == DCRX_ORIENT_FORWARD) ? 1 : 0; attempt < 2; attempt++) {
    const bool p1 = (cfg.flags & DCRX_F_PROFILE_LIST_SCAN_ONLY) != 0u, p12 = (cfg.flags & DCRX_F_PROFILE_SCAN_ONLY) != 0u;      // profiling aids: price the scan's passes alone (records are NOT results)
    if (attempt == 0) {
      const ScanOut so = scan_long<true, TABLE_LDS>(T, lds_trans, rv, hh, fw, fwk, p1);
      if (p1 || p12) { rec.v = (uint16_t)so.acc; rec.j = (uint16_t)(so.vstate + so.jstate); rec.v_start = (uint16_t)(so.vend + so.jend); rec.j_end = (uint16_t)hh.cnts; status = 254; break; }
      status = dcr_frame<true, TABLE_LDS, false>(T, lds_trans, rv, so, cfg, C, rec, &hh); frame = 0;
      if (status == DCRX_S_OK || cfg.orientation != DCRX_ORIENT_BOTH) break;
    } else {
      const ScanOut so = scan_long<false, TABLE_LDS>(T, lds_trans, rv, hh, fw, fwk, p1);
      if (p1 || p12) { rec.v = (uint16_t)so.acc; rec.j = (uint16_t)(so.vstate + so.jstate); rec.v_start = (uint16_t)(so.vend + so.jend); rec.j_end = (uint16_t)hh.cnts; status = 254; break; }
      status = dcr_frame<false, TABLE_LDS, false>(T, lds_trans, rv, so, cfg, C, rec, &hh); frame = 1;
    }
  }
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record(records + r, rec);
}

}  // namespace dcrx
