// dcrx_v2_device.h — per-read code of the v2 decombine kernel (dcrx_kernels.hip,
// decombine2_kernel) and of its test-only host build (tests/host_emul).
//
// What changes against dcrx_dcr_device.h's scan:
//   * the stored read is always scanned FORWARDS, from its first base and the root state; the
//     reverse frame (decombine.py:1000, revcomp :182-184) is served by an automaton over the
//     reverse-complemented keywords, so no frame needs a backwards walk and read lengths never
//     shape the scan;
//   * two bases per 16-bit entry (32 bytes per state: automata of up to 4095 states fit the LDS);
//     an entry carries the next state and four flags: a V tag / J tag / V half tag / J half tag
//     ends inside the pair;
//   * the scan keeps nothing but a log of those flags, four bits per pair (one v_alignbit per
//     step).  Flags of pairs that lie inside the read are exact (the scan starts at the root at
//     base 0: no context precedes the read); what the automaton saw is then resolved by comparing
//     the read's window with the packed keywords of the class (bucket tables, V2Ori) — the
//     automaton is the filter, the comparison decides which keyword ends where;
//   * half-tag hits (the findall() lists at decombine.py:294, :339, :422, :473) come out of the
//     same log, so a read that needs the half-tag rescue is never scanned twice.
#pragma once

#include "dcrx_dcr_device.h"

namespace dcrx {

#ifndef DCRX_HOST_EMUL
DCRX_DEV uint32_t dcrx_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
typedef __attribute__((address_space(3))) uint16_t dcrx_lds_u16;
struct V2Tab {};                      // the staged pair table starts at LDS address 0 (the kernel checks)
DCRX_DEV uint32_t v2_entry(const V2Tab &, uint32_t byte_off) {
  return *reinterpret_cast<const dcrx_lds_u16 *>(static_cast<uintptr_t>(byte_off));
}
#else
inline uint32_t dcrx_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
struct V2Tab { const uint16_t *base; };
inline uint32_t v2_entry(const V2Tab &tab, uint32_t byte_off) { return tab.base[byte_off >> 1]; }
#endif

// Typed access to the staged tables and to the reads: on the device an LDS address (32 bits) read
// with ds_read, and a global-memory pointer read with global_load — never a flat access, which
// would wait for both memory counters; in the host build plain pointers.
#ifndef DCRX_HOST_EMUL
typedef uint32_t dcrx_ldsaddr;
template <typename T>
DCRX_DEV T dcrx_lds_at(dcrx_ldsaddr a, uint32_t i) {
  return reinterpret_cast<const __attribute__((address_space(3))) T *>(static_cast<uintptr_t>(a))[i];
}
DCRX_DEV dcrx_ldsaddr dcrx_ldsaddr_of(const void *p) { return dcrx_lds_address(reinterpret_cast<const uint8_t *>(p)); }
typedef const __attribute__((address_space(1))) uint32_t *dcrx_gwords;
DCRX_DEV dcrx_gwords dcrx_gwords_of(const uint8_t *p) { return reinterpret_cast<dcrx_gwords>(reinterpret_cast<uintptr_t>(p)); }
#else
typedef uintptr_t dcrx_ldsaddr;
template <typename T>
inline T dcrx_lds_at(dcrx_ldsaddr a, uint32_t i) { return reinterpret_cast<const T *>(a)[i]; }
inline dcrx_ldsaddr dcrx_ldsaddr_of(const void *p) { return reinterpret_cast<uintptr_t>(p); }
typedef const uint32_t *dcrx_gwords;
inline dcrx_gwords dcrx_gwords_of(const uint8_t *p) { return reinterpret_cast<const uint32_t *>(p); }
#endif

// One step: two bases.  NIB2 = raw nibble of the packed read (first base | second base << 2) times two.
// NARROW (<= 2047 states): entry = state << 5 | flags, the entry's state bits ARE the row's byte offset.
#define DCRX_V2_STEP(NARROW, E, LOG, NIB2)                                                       \
  do {                                                                                            \
    const uint32_t off_ = (NARROW) ? (((E) & 0xFFE0u) | (uint32_t)(NIB2))                         \
                                   : ((((E) & 0xFFF0u) << 1) | (uint32_t)(NIB2));                 \
    (E) = v2_entry(tab, off_);                                                                    \
    (LOG) = dcrx_alignbit((E), (LOG), 4);                                                         \
  } while (0)

// The scan of RPL reads side by side (independent chains in one instruction stream).
// w[q][kk]: word kk of read q.  lg[q][kk]: nibble j = flags of the pair at bases 16kk+2j, 16kk+2j+1.
// npairs: pairs to scan (wave-uniform): (n + 1) / 2 for a batch of one length, 8 * words otherwise.
// NPAIRS_CT > 0: the number of pairs is known where the kernel is compiled (the 150-nt batches: 75) — no test per word, no
// counted loop for a partial word, and the chains of the two reads never drain at a word's end: one straight run of look-ups.
template <int NW, int RPL, bool NARROW, int NPAIRS_CT = 0>
DCRX_DEV void scan2(const V2Tab &tab, const uint32_t (&w)[RPL][NW], uint32_t (&lg)[RPL][NW], const int npairs_rt) {
  const int npairs = NPAIRS_CT > 0 ? NPAIRS_CT : npairs_rt;
  uint32_t e[RPL];
#pragma unroll
  for (int q = 0; q < RPL; q++) e[q] = 0;      // root
#pragma unroll
  for (int kk = 0; kk < NW; kk++) {
    const int cnt = npairs - 8 * kk;             // pairs of this word that are scanned
    if (cnt >= 8) {
      uint32_t lo[RPL], hi[RPL], l[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) { lo[q] = w[q][kk] << 1; hi[q] = w[q][kk] >> 3; l[q] = 0; }
      // nibble j, times two: bits 1..4 of byte j/2 of the word shifted left by one (even j) or right by three (odd j)
#define DCRX_V2_BYTE(BY)                                                                          \
  _Pragma("unroll") for (int q = 0; q < RPL; q++) DCRX_V2_STEP(NARROW, e[q], l[q], dcrx_byte_and<BY>(lo[q], 0x1Eu));  \
  _Pragma("unroll") for (int q = 0; q < RPL; q++) DCRX_V2_STEP(NARROW, e[q], l[q], dcrx_byte_and<BY>(hi[q], 0x1Eu));
      DCRX_V2_BYTE(0) DCRX_V2_BYTE(1) DCRX_V2_BYTE(2) DCRX_V2_BYTE(3)
#undef DCRX_V2_BYTE
#pragma unroll
      for (int q = 0; q < RPL; q++) lg[q][kk] = l[q];
    } else if (cnt > 0) {
      uint32_t wv[RPL], l[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) { wv[q] = w[q][kk]; l[q] = 0; }
      if constexpr (NPAIRS_CT > 0) {      // (a count the compiler knows: straight-line)
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (j < cnt) {
#pragma unroll
            for (int q = 0; q < RPL; q++) { DCRX_V2_STEP(NARROW, e[q], l[q], (wv[q] & 15u) << 1); wv[q] >>= 4; }
          }
      } else {
        for (int j = 0; j < cnt; j++) {
#pragma unroll
          for (int q = 0; q < RPL; q++) { DCRX_V2_STEP(NARROW, e[q], l[q], (wv[q] & 15u) << 1); wv[q] >>= 4; }
        }
      }
#pragma unroll
      for (int q = 0; q < RPL; q++) lg[q][kk] = l[q] >> (4 * (8 - cnt));
    } else {
#pragma unroll
      for (int q = 0; q < RPL; q++) lg[q][kk] = 0;
    }
  }
}

// What the dispatch after the scan needs from a read's log.
struct Digest2 {
  uint32_t any;            // OR of all flags (V2_F_*)
  uint32_t vf_n, vf_pair;  // pairs with a V tag ending inside, and the first such pair (8 * word + nibble)
  uint32_t jf_n, jf_pair;
};

// Clears the flags of pairs that start at or beyond base n (batches of mixed lengths scan whole words).
template <int NW>
DCRX_DEV void mask_log2(uint32_t (&lg)[NW], const int n) {
  const int pairs = (n + 1) >> 1;
#pragma unroll
  for (int kk = 0; kk < NW; kk++) {
    const int c = pairs - 8 * kk;
    const uint32_t m = c >= 8 ? 0xFFFFFFFFu : (c <= 0 ? 0u : ((1u << (4 * c)) - 1u));
    lg[kk] &= m;
  }
}

template <int NW>
DCRX_DEV Digest2 digest2(const uint32_t (&lg)[NW]) {
  Digest2 d;
  uint32_t o = 0, vn = 0, jn = 0, vp = 0xFFFFFFFFu, jp = 0xFFFFFFFFu;
#pragma unroll
  for (int kk = 0; kk < NW; kk++) {
    const uint32_t l = lg[kk];
    o |= l;
    const uint32_t tv = l & 0x11111111u, tj = l & 0x22222222u;
    vn += (uint32_t)dcrx_popc64(tv);
    jn += (uint32_t)dcrx_popc64(tj);
    // lowest set bit (all ones when none) | word index: the minimum over the words is the first pair
    const uint32_t qv = tv ? (uint32_t)dcrx_ctz32(tv) : 0xFFFFFFFFu, qj = tj ? (uint32_t)dcrx_ctz32(tj) : 0xFFFFFFFFu;
    vp = min(vp, qv | ((uint32_t)kk << 5));
    jp = min(jp, qj | ((uint32_t)kk << 5));
  }
  o |= o >> 16; o |= o >> 8; o |= o >> 4;
  d.any = o & 0xFu;
  d.vf_n = vn; d.vf_pair = vp >> 2;
  d.jf_n = jn; d.jf_pair = jp >> 2;
  return d;
}

// The digest as the scan kernel takes it (same results where they are read: `any`, the two counts, and a gene's pair when
// exactly one pair holds its tag — tail2_pack, rescue2_digest_pack and classify2 ask for nothing else; with several such pairs
// the pair fields hold no meaning).  Per word: the OR, and per gene a masked population count weighted with 1 + (word << 12),
// so that one sum carries the count (low 12 bits) and — one pair — the word it lies in; the pair inside the word comes out of
// the OR of all words.  7 vector instructions per word where digest2 takes 17: - 3.7 % of config 2's step.
template <int NW>
DCRX_DEV Digest2 digest2_lean(const uint32_t (&lg)[NW]) {
  uint32_t o = 0, av = 0, aj = 0;
#pragma unroll
  for (int kk = 0; kk < NW; kk++) {
    const uint32_t l = lg[kk];
    const uint32_t k = 1u + ((uint32_t)kk << 12);
    o |= l;
    av += (uint32_t)dcrx_popc32(l & 0x11111111u) * k;
    aj += (uint32_t)dcrx_popc32(l & 0x22222222u) * k;
  }
  Digest2 d;
  const uint32_t ov = o & 0x11111111u, oj = o & 0x22222222u;
  o |= o >> 16; o |= o >> 8; o |= o >> 4;
  d.any = o & 0xFu;
  d.vf_n = av & 0xFFFu; d.jf_n = aj & 0xFFFu;
  d.vf_pair = ((av >> 12) << 3) | (ov ? ((uint32_t)dcrx_ctz32(ov) >> 2) : 0u);
  d.jf_pair = ((aj >> 12) << 3) | (oj ? ((uint32_t)dcrx_ctz32(oj) >> 2) : 0u);
  return d;
}

// The ten-word shape's digest on MERGED words (round 6): the two full-tag flags of two words share one word (bit-field insert:
// a shift and one v_bfi), then each gene's flags of four words share one — bit i of a nibble: words 0, 2, 1, 3 of the four —,
// so that counts and positions are taken on three words per gene instead of ten: ~65 vector instructions where digest2_lean takes
// ~100.  Same contract: `any`, the two counts, and a gene's pair when exactly one pair holds its tag.
DCRX_DEV uint32_t dcrx_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
DCRX_DEV Digest2 digest2_packed10(const uint32_t (&lg)[10]) {
  uint32_t o = lg[0];
#pragma unroll
  for (int kk = 1; kk < 10; kk++) o |= lg[kk];
  uint32_t M[5];      // nibble bits: V tag of word 2 j, J tag of word 2 j, V tag of word 2 j + 1, J tag of word 2 j + 1
#pragma unroll
  for (int j = 0; j < 5; j++) M[j] = dcrx_bfi(0x33333333u, lg[2 * j], lg[2 * j + 1] << 2);
  const uint32_t e = 0x55555555u;
  const uint32_t q0 = dcrx_bfi(e, M[0], M[1] << 1), q1 = dcrx_bfi(e, M[2], M[3] << 1), q2 = M[4] & e;                // V tags: words 0-3, 4-7, 8-9
  const uint32_t r0 = dcrx_bfi(e, M[0] >> 1, M[1]), r1 = dcrx_bfi(e, M[2] >> 1, M[3]), r2 = (M[4] >> 1) & e;          // J tags, the same places
  const uint32_t av = (uint32_t)dcrx_popc32(q0) + (uint32_t)dcrx_popc32(q1) * 0x1001u + (uint32_t)dcrx_popc32(q2) * 0x2001u;
  const uint32_t aj = (uint32_t)dcrx_popc32(r0) + (uint32_t)dcrx_popc32(r1) * 0x1001u + (uint32_t)dcrx_popc32(r2) * 0x2001u;
  auto place = [](const uint32_t a, const uint32_t any3) -> uint32_t {      // the pair of a gene's one flag: 8 * word + nibble
    const uint32_t tb = any3 ? (uint32_t)dcrx_ctz32(any3) : 0u;
    const uint32_t i = tb & 3u, word = 4u * (a >> 12) + (((i & 1u) << 1) | (i >> 1));
    return (word << 3) | (tb >> 2);
  };
  Digest2 d;
  o |= o >> 16; o |= o >> 8; o |= o >> 4;
  d.any = o & 0xFu;
  d.vf_n = av & 0xFFFu; d.jf_n = aj & 0xFFFu;
  d.vf_pair = place(av, q0 | q1 | q2);
  d.jf_pair = place(aj, r0 | r1 | r2);
  return d;
}
// what the scan kernel (and its host emulation) takes
template <int NW>
DCRX_DEV Digest2 digest2_scan(const uint32_t (&lg)[NW]) {
  if constexpr (NW == 10) return digest2_packed10(lg);
  else return digest2_lean<NW>(lg);
}

// flags of pair index `pair` (any lane-varying index: a select chain over the words)
template <int NW>
DCRX_DEV uint32_t log_nibble(const uint32_t (&lg)[NW], const int pair) {
  uint32_t l = 0;
#pragma unroll
  for (int kk = 0; kk < NW; kk++) l = ((pair >> 3) == kk) ? lg[kk] : l;
  return (l >> (4 * (pair & 7))) & 0xFu;
}

// ---- what a read needs after the scan -----------------------------------------------------------
//   V2_VNONE   no V tag and no V half tag: NoVDetected (decombine.py:393), the record is final
//   V2_VMULTI  two or more V tags: MultipleVtagMatches (:278-280), final
//   V2_TAIL    exactly one pair holds a V tag and the J side needs no half-tag rescue: the tail
//              resolves both tags from their pairs (entry: TailEntry2)
//   V2_EVENTS  anything else (half-tag rescue of V or of J, a flag on the half pair at the end of an
//              odd-length read): resolved from the read's event list (entry: events, up to V2_MAX_EVENTS)
enum { V2_VNONE = 0, V2_VMULTI = 1, V2_TAIL = 2, V2_EVENTS = 3 };
constexpr int V2_MAX_EVENTS = 8;      // 12 bits each: pair (8) | flags (4) << 8; 96 bits in three dwords

// `bnd`: flags of the pair that holds the last base of an odd-length read beside a base that is not
// part of the read (0 for even lengths): those flags may be set by a keyword that runs over the end.
DCRX_DEV int classify2(const Digest2 &d, const uint32_t bnd) {
  if (!(d.any & (V2_F_VF | V2_F_VH))) return V2_VNONE;
  if (bnd & (V2_F_VF | V2_F_JF)) return V2_EVENTS;
  if (d.vf_n >= 2) return V2_VMULTI;
  if (d.vf_n == 1) return (d.jf_n >= 1 || !(d.any & V2_F_JH)) ? V2_TAIL : V2_EVENTS;
  return V2_EVENTS;                      // V half tags only
}

// tail entry, second dword: V pair | J pair << 8 | J class << 16 (0 none, 1 one pair, 2 several pairs)
DCRX_DEV uint32_t tail2_pack(const Digest2 &d) {
  const uint32_t jc = d.jf_n < 2u ? d.jf_n : 2u;
  return (d.vf_pair & 0xFFu) | ((jc ? (d.jf_pair & 0xFFu) : 0u) << 8) | (jc << 16);
}

// An event list: up to V2_MAX_EVENTS entries of 12 bits (pair | flags << 8; flags 0 = empty), entry i
// at bits 12 i of the 96-bit string lo (64) : hi (32).
struct Events2 {
  uint64_t lo; uint32_t hi;
  DCRX_DEV void put(int i, uint32_t x) {
    const int bp = 12 * i;
    if (bp < 64) lo |= (uint64_t)x << bp;
    if (bp < 64 && bp + 12 > 64) hi |= x >> (64 - bp);
    if (bp >= 64) hi |= x << (bp - 64);
  }
  DCRX_DEV uint32_t get(int i) const {
    const int bp = 12 * i;
    uint64_t v = bp < 64 ? (lo >> bp) : 0ull;
    if (bp < 64 && bp + 12 > 64) v |= (uint64_t)hi << (64 - bp);
    if (bp >= 64) v = hi >> (bp - 64);
    return (uint32_t)v & 0xFFFu;
  }
  DCRX_DEV int count() const {
    int n = 0;
#pragma unroll
    for (int i = 0; i < V2_MAX_EVENTS; i++) n += (get(i) & 0xF00u) ? 1 : 0;
    return n;
  }
};

// The event list of a read: every pair with a flag that can matter, ascending.  V half-tag flags
// are dropped when a V tag was seen (the half tags are only consulted without one, :292), J likewise;
// `bnd` keeps them (the full-tag flag may then not stand).  Returns false when the read has more
// than V2_MAX_EVENTS such pairs.
template <int NW>
DCRX_DEV bool events2(const uint32_t (&lg)[NW], const Digest2 &d, const uint32_t bnd, uint32_t (&ev)[3]) {
  uint32_t keep = 0xFu;
  if (!bnd) {
    if (d.vf_n) keep &= ~V2_F_VH;
    if (d.jf_n) keep &= ~V2_F_JH;
  }
  Events2 E{0, 0};
  int ne = 0;
  const uint32_t keep8 = keep * 0x11111111u;
#pragma unroll
  for (int kk = 0; kk < NW; kk++) {
    uint32_t m = lg[kk] & keep8;
    while (m) {
      const int j = dcrx_ctz32(m) >> 2;
      const uint32_t fl = (m >> (4 * j)) & 0xFu;
      m &= ~(0xFu << (4 * j));
      if (ne < V2_MAX_EVENTS) E.put(ne, (uint32_t)(kk * 8 + j) | (fl << 8));
      ne++;
    }
  }
  ev[0] = (uint32_t)E.lo; ev[1] = (uint32_t)(E.lo >> 32); ev[2] = E.hi;
  return ne <= V2_MAX_EVENTS;
}

// events of a tail entry: its V pair and (one J pair) its J pair
// flags that travel in the top bits of an entry's read index (batches of fewer than 2^30 reads)
constexpr uint32_t V2_R_JMULTI = 0x80000000u;    // several pairs hold a J tag (they are not listed)
constexpr uint32_t V2_R_EXC = 0x40000000u;       // the read has exception bytes
constexpr uint32_t V2_R_MASK = 0x3FFFFFFFu;
DCRX_DEV void tail2_events(const uint32_t t, uint32_t (&ev)[3], bool &jmulti) {
  const uint32_t vp = t & 0xFFu, jp = (t >> 8) & 0xFFu, jc = (t >> 16) & 3u;
  jmulti = jc == 2u;
  const uint32_t ve = vp | (V2_F_VF << 8), je = jp | (V2_F_JF << 8);
  Events2 E{0, 0};
  if (jc != 1u) E.put(0, ve);
  else if (jp == vp) E.put(0, vp | ((V2_F_VF | V2_F_JF) << 8));
  else if (jp > vp) { E.put(0, ve); E.put(1, je); }
  else { E.put(0, je); E.put(1, ve); }
  ev[0] = (uint32_t)E.lo; ev[1] = (uint32_t)(E.lo >> 32); ev[2] = E.hi;
}

// ------------------------------------------------------------------------------
// A read whose packed words sit in registers, seen in one frame: the interface of Frame<REV>
// (dcrx_dcr_device.h) without a memory access — a lane-varying word index is a select chain over
// the NW registers.  The finishing code of the v2 kernel works on this: one round of loads per
// read, then no load at all.
// ------------------------------------------------------------------------------
constexpr int V2_MAX_EXC = 4;        // single exception bytes of a read the register frame holds beside one run of Ns (a read with more takes the list kernel)
// A read's exception bytes as the register frame holds them: its longest run of consecutive 'N' bytes (an N tail, an N head, a
// stretch of Ns, a whole read of Ns: what a sequencer writes where it cannot call bases) as a range of stored positions, and up
// to V2_MAX_EXC further bytes one by one.  False when the read has more than that.
struct ExcLayout { int run_lo, run_hi; uint64_t xpos; uint32_t xchr; int nx; };
DCRX_DEVNI bool exc_layout(const uint16_t *exc_pos, const uint8_t *exc_chr, const int e0, const int e1, ExcLayout &L) {
  L.run_lo = L.run_hi = 0; L.xpos = 0; L.xchr = 0; L.nx = 0;
  if (e1 - e0 > V2_MAX_EXC) {      // the longest run of Ns (entries are sorted by position)
    int best_a = e0, best_n = 0;
    for (int a = e0; a < e1;) {
      int b = a;
      if (exc_chr[a] == (uint8_t)'N') { b = a + 1; while (b < e1 && exc_chr[b] == (uint8_t)'N' && (int)exc_pos[b] == (int)exc_pos[b - 1] + 1) b++; }
      if (b - a > best_n) { best_n = b - a; best_a = a; }
      a = b > a ? b : a + 1;
    }
    if (e1 - e0 - best_n > V2_MAX_EXC) return false;
    L.run_lo = (int)exc_pos[best_a]; L.run_hi = L.run_lo + best_n;
    for (int k = e0; k < e1; k++) {
      if (k >= best_a && k < best_a + best_n) continue;
      L.xpos |= (uint64_t)exc_pos[k] << (16 * L.nx); L.xchr |= (uint32_t)exc_chr[k] << (8 * L.nx); L.nx++;
    }
    return true;
  }
  for (int k = e0; k < e1; k++) { L.xpos |= (uint64_t)exc_pos[k] << (16 * L.nx); L.xchr |= (uint32_t)exc_chr[k] << (8 * L.nx); L.nx++; }
  return true;
}
template <bool REV_, int NW>
struct FrameReg {
  static constexpr bool kRev = REV_;
  static constexpr bool kWindowedWalks = true;
  const uint32_t (&w)[NW];
  const ReadView &r;       // length and complement table (r.words and the exception list are not read after construction)
  // the read's exception bytes, loaded once: a run of Ns [run_lo, run_hi) in stored positions, and stored position (16 bits each)
  // and byte (8 bits each) of up to V2_MAX_EXC single ones
  uint64_t xpos; uint32_t xchr; int nx; int run_lo, run_hi;
  DCRX_DEVNI FrameReg(const uint32_t (&words)[NW], const ReadView &rv) : w(words), r(rv), xpos(0), xchr(0), nx(0), run_lo(0), run_hi(0) {
    ExcLayout L;
    (void)exc_layout(rv.exc_pos, rv.exc_chr, rv.e0, rv.e1, L);      // (the caller has checked that the read fits)
    xpos = L.xpos; xchr = L.xchr; nx = L.nx; run_lo = L.run_lo; run_hi = L.run_hi;
  }
  DCRX_DEV int xp(int k) const { return (int)((xpos >> (16 * k)) & 0xFFFFu); }
  DCRX_DEV uint8_t xc(int k) const { return (uint8_t)((xchr >> (8 * k)) & 0xFFu); }
  DCRX_DEV int n() const { return r.n; }
  DCRX_DEV int fpos(int i) const { return REV_ ? r.n - 1 - i : i; }
  DCRX_DEV uint32_t word(int idx) const {        // 0 beyond the registers
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) v = (idx == k) ? w[k] : v;
    return v;
  }
  DCRX_DEV int code(int i) const {
    const int m = fpos(i);
    const int c = (int)((word(m >> 4) >> ((m & 15) * 2)) & 3u);
    return REV_ ? (c ^ 3) : c;
  }
  DCRX_DEV int exc_index(int i) const {          // index into the held single exceptions, -2 inside the run of Ns, or -1
    const int m = fpos(i);
    int at = (m >= run_lo && m < run_hi) ? -2 : -1;
    for (int k = 0; k < nx; k++)
      if (xp(k) == m) at = k;
    return at;
  }
  DCRX_DEV bool has_exc() const { return nx > 0 || run_hi > run_lo; }
  // the run in frame positions: [flo, fhi)
  DCRX_DEV int run_flo() const { return REV_ ? r.n - run_hi : run_lo; }
  DCRX_DEV int run_fhi() const { return REV_ ? r.n - run_lo : run_hi; }
  DCRX_DEV bool clean(int a, int b) const {
    bool ok = !(run_hi > run_lo && run_flo() < b && run_fhi() > a);
    for (int k = 0; k < nx; k++) {
      const int i = REV_ ? r.n - 1 - xp(k) : xp(k);
      if (i >= a && i < b) ok = false;
    }
    return ok;
  }
  DCRX_DEV uint64_t exc_slots(int b) const {      // exception bytes among the stored bases [b, b + 32): bit 2s for base b + s
    uint64_t m = 0;
    for (int k = 0; k < nx; k++) {
      const int d = xp(k) - b;
      if (d >= 0 && d < 32) m |= 1ull << (2 * d);
    }
    const int lo = max(run_lo - b, 0), hi = min(run_hi - b, 32);      // the run's share of the window
    if (hi > lo) {
      const uint64_t upto_hi = hi >= 32 ? ~0ull : ((1ull << (2 * hi)) - 1ull);
      m |= 0x5555555555555555ull & upto_hi & ~((1ull << (2 * lo)) - 1ull);
    }
    return m;
  }
  DCRX_DEVNI bool has_N(int lo, int hi) const {
    bool hasN = run_hi > run_lo && run_flo() < hi && run_fhi() > lo;      // (N is its own complement: the run reads N in either frame)
    for (int k = 0; k < nx; k++) {
      const int i = REV_ ? r.n - 1 - xp(k) : xp(k);
      const uint8_t b = REV_ ? r.comp[xc(k)] : xc(k);
      if (i >= lo && i < hi && b == (uint8_t)'N') hasN = true;
    }
    return hasN;
  }
  DCRX_DEVNI uint8_t chr(int i) const {
    if (has_exc()) {
      const int k = exc_index(i);
      if (k == -2) return (uint8_t)'N';
      if (k >= 0) { const uint8_t b = xc(k); return REV_ ? r.comp[b] : b; }
    }
    return (uint8_t)("ACGT"[code(i)]);
  }
  // 32 bases of the stored read from base s >= 0 (one select chain per word, the three share their compares)
  DCRX_DEV uint64_t stored64(int s) const {
    const int i = s >> 4, sh = (s & 15) * 2;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      const bool at = i == k;
      w0 = at ? w[k] : w0;
      if (k + 1 < NW) w1 = at ? w[k + 1] : w1;
      if (k + 2 < NW) w2 = at ? w[k + 2] : w2;
    }
    return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
  }
  DCRX_DEV uint32_t window(int a, int len) const {
    const int lo = REV_ ? r.n - a - len : a;
    const uint32_t v = (uint32_t)stored64(lo);
    return v & ((len >= 16) ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u));
  }
  DCRX_DEV uint64_t load64(int b) const { return stored64(b); }
};

// the same window for a read in memory
template <bool REV>
DCRX_DEV uint64_t frame_stored64(const Frame<REV> &F, const int nwords, const int s);
template <bool REV, int NW>
DCRX_DEV uint64_t frame_stored64(const FrameReg<REV, NW> &F, const int, const int s) { return F.stored64(s); }

// ---- resolving events: window of the stored read against the packed keywords of a class ----------
// 32 bases of the packed read from base s >= 0 (bases beyond the allocated words read as A)
DCRX_DEV uint64_t win64(const uint32_t *words, const int nwords, const int s) {
  const int i = s >> 4, sh = (s & 15) * 2;
  const uint32_t w0 = i < nwords ? words[i] : 0u;
  const uint32_t w1 = i + 1 < nwords ? words[i + 1] : 0u;
  const uint32_t w2 = (sh && i + 2 < nwords) ? words[i + 2] : 0u;
  return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
}

template <bool REV>
DCRX_DEV uint64_t frame_stored64(const Frame<REV> &F, const int nwords, const int s) { return win64(F.r.words, nwords, s); }

// class-local index of the keyword whose packed form (as the stored read shows it) is `val`, or -1
// (the per-class members are picked with constant indices: a lane-varying index into a struct that
// lives in registers would send the struct to scratch memory)
#define DCRX_V2_PICK(ARR, CLS) ((CLS) == 0 ? (ARR)[0] : (CLS) == 1 ? (ARR)[1] : (CLS) == 2 ? (ARR)[2] : (CLS) == 3 ? (ARR)[3] : (CLS) == 4 ? (ARR)[4] : (ARR)[5])
DCRX_DEVNI int v2_lookup(const V2Ori &V, const int cls, const uint64_t val) {
  const uint16_t *kw = reinterpret_cast<const uint16_t *>(V.bk + DCRX_V2_PICK(V.bk_kw_off, cls));
  const uint64_t *pk = reinterpret_cast<const uint64_t *>(V.bk + DCRX_V2_PICK(V.bk_pk_off, cls));
  uint32_t s1, s2;
  v2_slots(val, DCRX_V2_PICK(V.bk_mask, cls), s1, s2);
  const uint64_t p1 = pk[s1], p2 = pk[s2];
  const uint32_t k1 = kw[s1], k2 = kw[s2];
  if (p1 == val && k1 != V2_PH_EMPTY) return (int)k1;
  if (p2 == val && k2 != V2_PH_EMPTY) return (int)k2;
  return -1;
}

// ------------------------------------------------------------------------------
// dcr() for one frame from a read's events — decombine.py:534-585 with vanalysis :273-394 and
// janalysis :397-531 — in two stages, so that a wave runs each piece of work once:
//   1. one sweep over the events in the frame's findall order (ascending end position); per
//      event one 32-base window of the stored read, out of which every keyword class its flags
//      name is tested at both bases of the pair (window slice -> bucket -> compare).  The hits
//      go to short per-class lists (the findall() lists, in order);
//   2. vanalysis from the V lists, then janalysis from the J lists, exactly as the reference
//      walks them.
// Returns DCRX_S_DEFER, having counted nothing, when a half-tag list does not fit its registers
// (the caller hands the read to the three-launch form).
// jmulti: several pairs hold a J tag (tail entries; their pairs are not listed).
// ------------------------------------------------------------------------------
constexpr int V2_MAX_HITS = 8;
struct Hits2 {            // hits of one class in findall order: 32 bits each, keyword (class-local) << 16 | stored end base
  uint64_t a, b, c, d;    // slots 0-1, 2-3, 4-5, 6-7
  int n;
  DCRX_DEV void add(int kw, int f) {
    const uint64_t h = ((uint64_t)(uint32_t)kw << 16) | (uint64_t)(uint32_t)f;
    const uint64_t x = h << (32 * (n & 1));
    if ((n >> 1) == 0) a |= x; else if ((n >> 1) == 1) b |= x; else if ((n >> 1) == 2) c |= x; else if ((n >> 1) == 3) d |= x;
    n++;
  }
  DCRX_DEV uint32_t at(int i) const {
    const uint64_t w = (i >> 1) == 0 ? a : ((i >> 1) == 1 ? b : ((i >> 1) == 2 ? c : d));
    return (uint32_t)(w >> (32 * (i & 1)));
  }
};

template <class FR>
DCRX_DEVNI int dcr_frame3(const DevTables &T, const V2Ori &V, const FR &F, const int nwords, const uint32_t (&ev_in)[3],
                          const bool jmulti, const CfgDev &cfg, const Counters &C, dcrx_record_t &rec) {
  const Events2 E{(uint64_t)ev_in[0] | ((uint64_t)ev_in[1] << 32), ev_in[2]};   // scalars: an array picked with a lane-varying index would live in scratch memory
  constexpr bool REV = FR::kRev;
  const int n = F.n();
  const GeneDevPtrs &GV = T.g[0];
  const GeneDevPtrs &GJ = T.g[1];
  const int ne = E.count();
  const int Lvf = (int)T.kw_len[K_VFULL], Ljf = (int)T.kw_len[K_JFULL], Lv1 = (int)T.kw_len[K_VH1], Lv2 = (int)T.kw_len[K_VH2],
            Lj1 = (int)T.kw_len[K_JH1], Lj2 = (int)T.kw_len[K_JH2];
  Hits2 hvf{0, 0, 0, 0, 0}, hjf{0, 0, 0, 0, 0}, hv1{0, 0, 0, 0, 0}, hv2{0, 0, 0, 0, 0}, hj1{0, 0, 0, 0, 0}, hj2{0, 0, 0, 0, 0};
  // ---- stage 1: the sweep, V flags first, then J flags.  Per flagged pair one window of the stored
  // read; per base of the pair two look-ups at most: the full tag or (no full-tag flag) the first
  // half tag — one piece of code, the class picked per lane — and the second half tag. ----
  auto sweep = [&](const uint32_t f_full, const uint32_t f_half, const int c_full, const int c_h1, const int c_h2, const int Lf,
                   const int L1, const int L2, Hits2 &hf, Hits2 &h1, Hits2 &h2) {
    // where the three classes' buckets are (picked per lane below between the full tag's and the first half tag's)
    const uint32_t so_f = V.bk_mask[c_full], ko_f = V.bk_kw_off[c_full], po_f = V.bk_pk_off[c_full];
    const uint32_t so_1 = V.bk_mask[c_h1], ko_1 = V.bk_kw_off[c_h1], po_1 = V.bk_pk_off[c_h1];
    const uint32_t so_2 = V.bk_mask[c_h2], ko_2 = V.bk_kw_off[c_h2], po_2 = V.bk_pk_off[c_h2];
    auto lookup = [&](const uint32_t mask, const uint32_t ko, const uint32_t po, const uint64_t val) {
      uint32_t s1, s2;
      v2_slots(val, mask, s1, s2);
      const uint64_t *pk = reinterpret_cast<const uint64_t *>(V.bk + po);
      const uint16_t *kws = reinterpret_cast<const uint16_t *>(V.bk + ko);
      const uint64_t p1 = pk[s1], p2 = pk[s2];
      const uint32_t k1 = kws[s1], k2 = kws[s2];
      int kw = -1;
      if (p2 == val && k2 != V2_PH_EMPTY) kw = (int)k2;
      if (p1 == val && k1 != V2_PH_EMPTY) kw = (int)k1;
      return kw;
    };
    for (int x = 0; x < ne; x++) {
      const int i = REV ? ne - 1 - x : x;
      const uint32_t e = E.get(i);
      const uint32_t fl = e >> 8;
      if (!(fl & (f_full | f_half))) continue;
      const bool full = (fl & f_full) != 0u, half = (fl & f_half) != 0u;
      const int f1 = 2 * (int)(e & 0xFFu) + 1;               // the pair's second stored base
      const int xs = f1 >= 31 ? f1 - 31 : 0;                 // the window: stored bases [xs, xs + 32)
      const uint64_t X = frame_stored64(F, nwords, xs);
      for (int y = 0; y < 2; y++) {
        const int f = REV ? f1 - y : f1 - 1 + y;             // ascending end position in the frame
        // keyword (L long) whose occurrence in the stored read ends at base f: its packed form, or ~0 when it cannot be there
        auto slice = [&](const int L) -> uint64_t {
          const int s = f - L + 1;
          if (s < 0 || f >= n) return ~0ull;
          const int p = REV ? n - s - L : s;
          if (F.has_exc() && !F.clean(p, p + L)) return ~0ull;   // an exception byte sends the automata back to their roots
          const uint64_t mask = L >= 32 ? ~0ull : ((1ull << (2 * L)) - 1ull);
          return (X >> (2 * (s - xs))) & mask;
        };
        {   // the full tag where the pair carries its flag; else the first half tag
          const int L = full ? Lf : L1;
          const uint64_t val = slice(L);
          const bool can = L >= 32 || val != ~0ull;            // (a 32-base keyword of all T packs to ~0: let it through)
          const int kw = can ? lookup(full ? so_f : so_1, full ? ko_f : ko_1, full ? po_f : po_1, val) : -1;
          if (kw >= 0) { if (full) hf.add(kw, f); else h1.add(kw, f); }
        }
        if (half) {
          const uint64_t val = slice(L2);
          const bool can = L2 >= 32 || val != ~0ull;
          const int kw = can ? lookup(so_2, ko_2, po_2, val) : -1;
          if (kw >= 0) h2.add(kw, f);
          if (full) {                                          // full-tag and half-tag flags on one pair: the first half tag as well
            const uint64_t v1 = slice(L1);
            const int k1 = (L1 >= 32 || v1 != ~0ull) ? lookup(so_1, ko_1, po_1, v1) : -1;
            if (k1 >= 0) h1.add(k1, f);
          }
        }
      }
    }
  };
  sweep(V2_F_VF, V2_F_VH, K_VFULL, K_VH1, K_VH2, Lvf, Lv1, Lv2, hvf, hv1, hv2);
  sweep(V2_F_JF, V2_F_JH, K_JFULL, K_JH1, K_JH2, Ljf, Lj1, Lj2, hjf, hj1, hj2);
  if (cfg.flags & DCRX_F_PROFILE_RESCUE_HITS_ONLY) {   // profiling aid: price the sweep alone (records are NOT results)
    rec.v = (uint16_t)(hvf.n + hjf.n + hv1.n + hv2.n + hj1.n + hj2.n); rec.j = (uint16_t)(hvf.a ^ hjf.a ^ hv1.a ^ hv2.a ^ hj1.a ^ hj2.a);
    return 254;
  }
  // a half-tag list that will be walked and does not fit: nothing has been counted yet
  if ((hvf.n == 0 && (hv1.n > V2_MAX_HITS || (hv1.n == 0 && hv2.n > V2_MAX_HITS))) ||
      (hjf.n == 0 && !jmulti && (hj1.n > V2_MAX_HITS || (hj1.n == 0 && hj2.n > V2_MAX_HITS))))
    return DCRX_S_DEFER;
  auto frame_start = [&](const int f, const int L) { const int s = f - L + 1; return REV ? n - s - L : s; };
  XDat vdat{0, 0, 0, 0}, jdat{0, 0, 0, 0};

  // ---- stage 2: vanalysis ----
  if (hvf.n > 1) { C.add(DCRX_C_MULTIPLE_V_MATCHES); return DCRX_S_V_MULTI; }      // :278-280
  if (hvf.n == 1) {
    const uint32_t h = hvf.at(0);
    const int v = (int)T.kw_first[T.kw_base[K_VFULL] + (h >> 16)];                 // v_seqs.index(tag) :282
    const int vp = frame_start((int)(h & 0xFFFFu), Lvf);
    const int te = vp + GV.jump[v] - 1;                                            // :283-285
    int end_v, dels;
    if (!get_v_deletions(GV, F, v, te, end_v, dels, C))                            // :288-290
      return (te >= n) ? DCRX_S_V_WALK_FAIL_AT_END : DCRX_S_V_WALK_FAIL;
    vdat = XDat{v, end_v, dels, vp};
  } else {
    const int half = hv1.n ? 1 : 2;                          // half 2 only when no half-1 keyword occurs (:292-294, :337-339)
    const Hits2 hh{hv1.n ? hv1.a : hv2.a, hv1.n ? hv1.b : hv2.b, hv1.n ? hv1.c : hv2.c, hv1.n ? hv1.d : hv2.d, hv1.n ? hv1.n : hv2.n};   // by value: a reference picked at run time would put both lists in memory
    if (hh.n == 0) { C.add(DCRX_C_NO_VTAGS_FOUND); return DCRX_S_V_NONE; }         // :393-394
    const int cls = half == 1 ? K_VH1 : K_VH2, L = half == 1 ? Lv1 : Lv2;
    bool done = false;
    for (int k = 0; k < hh.n && !done; k++) {
      const uint32_t h = hh.at(k);
      done = rescue_candidates(T, F, 0, half, T.kw_base[cls] + (h >> 16), L, frame_start((int)(h & 0xFFFFu), L), 0, vdat, C);
    }
    if (!done) {
      C.add(half == 1 ? DCRX_C_FOUNDV1NOTV2 : DCRX_C_FOUNDV2NOTV1);                 // :334 / :389
      return half == 1 ? DCRX_S_V_HALF1_EXHAUSTED : DCRX_S_V_HALF2_EXHAUSTED;
    }
  }
  const int end_of_v = vdat.pos + 1;                                               // :547

  // ---- janalysis ----
  int jstatus = DCRX_S_OK;
  const int jcount = jmulti ? 2 : hjf.n;
  if (jcount > 1) { C.add(DCRX_C_MULTIPLE_J_MATCHES); jstatus = DCRX_S_J_MULTI; }  // :402-404
  else if (jcount == 1) {
    const uint32_t h = hjf.at(0);
    const int j = (int)T.kw_first[T.kw_base[K_JFULL] + (h >> 16)];                 // j_seqs.index(tag) :406
    const int jp = frame_start((int)(h & 0xFFFFu), Ljf);
    const int Lj = (int)GJ.tag_len[j];
    const int ts = jp - GJ.jump[j];                                                // :407-409
    int start_j, dels;
    if (get_j_deletions(GJ, F, j, ts, end_of_v, start_j, dels, C)) jdat = XDat{j, start_j, dels, jp + Lj};  // :411-418
    else jstatus = DCRX_S_J_WALK_FAIL;
  } else {
    const int half = hj1.n ? 1 : 2;
    const Hits2 hh{hj1.n ? hj1.a : hj2.a, hj1.n ? hj1.b : hj2.b, hj1.n ? hj1.c : hj2.c, hj1.n ? hj1.d : hj2.d, hj1.n ? hj1.n : hj2.n};
    if (hh.n == 0) { C.add(DCRX_C_NO_J_ASSIGNED); jstatus = DCRX_S_J_NONE; }       // :530-531
    else {
      const int cls = half == 1 ? K_JH1 : K_JH2, L = half == 1 ? Lj1 : Lj2;
      bool done = false;
      for (int k = 0; k < hh.n && !done; k++) {
        const uint32_t h = hh.at(k);
        done = rescue_candidates(T, F, 1, half, T.kw_base[cls] + (h >> 16), L, frame_start((int)(h & 0xFFFFu), L), end_of_v, jdat, C);
      }
      if (!done) {
        C.add(half == 1 ? DCRX_C_FOUNDJ1NOTJ2 : DCRX_C_FOUNDV2NOTV1);               // :469 / :526 (the reference bumps the V key)
        jstatus = half == 1 ? DCRX_S_J_HALF1_EXHAUSTED : DCRX_S_J_HALF2_EXHAUSTED;
      }
    }
  }
  if (jstatus != DCRX_S_OK) { C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); return jstatus; }  // :583-585
  return dcr_filters(T, F, vdat, jdat, cfg, C, rec);
}

// One read from its events to its record, the read's words loaded into registers once (NW words;
// typed global loads on the device).  false: the read goes to the three-launch form (see dcr_frame3).  [x0, x1): the read's slice of the exception list (x0 == x1: a clean read).
template <bool UNIFORM_LEN, int NW, int ORI = -1>
DCRX_DEV bool finish2_words(const DevTables &T, const V2Ori &V, const BatchDev &B, const CfgDev &cfg, const uint64_t r,
                            const uint32_t (&w)[NW], const uint32_t (&ev)[3], const bool jmulti, const int x0, const int x1,
                            const Counters &C, dcrx_record_t *records);

// the first NW packed words of read r into registers: typed global loads on the device, two words
// per load (reads start on 8-byte boundaries: stride is a multiple of 8).  Lanes of a wave hold
// scattered reads, so every load instruction visits about as many cache lines as there are lanes:
// few, wide loads.
template <int NW>
DCRX_DEV void load_words(const BatchDev &B, const uint64_t r, uint32_t (&w)[NW]) {
  const uint32_t nw = B.stride >> 2;
#ifndef DCRX_HOST_EMUL
  typedef uint32_t dcrx_v2u __attribute__((ext_vector_type(2)));
  const __attribute__((address_space(1))) dcrx_v2u *gw =
      reinterpret_cast<const __attribute__((address_space(1))) dcrx_v2u *>(reinterpret_cast<uintptr_t>(B.packed + r * B.stride));
#pragma unroll
  for (int k = 0; k < NW / 2; k++) {
    dcrx_v2u t = {0u, 0u};
    if ((uint32_t)(2 * k) < nw) t = gw[k];
    w[2 * k] = t.x; w[2 * k + 1] = t.y;
  }
#else
  const dcrx_gwords gw = dcrx_gwords_of(B.packed + r * B.stride);
#pragma unroll
  for (int k = 0; k < NW; k++) w[k] = (uint32_t)k < nw ? gw[k] : 0u;
#endif
}

template <bool UNIFORM_LEN, int NW>
DCRX_DEV bool finish2_reg(const DevTables &T, const V2Ori &V, const BatchDev &B, const CfgDev &cfg, const uint64_t r,
                          const uint32_t (&ev)[3], const bool jmulti, const int x0, const int x1, const Counters &C,
                          dcrx_record_t *records) {
  uint32_t w[NW];
  load_words<NW>(B, r, w);
  return finish2_words<UNIFORM_LEN, NW>(T, V, B, cfg, r, w, ev, jmulti, x0, x1, C, records);
}

// ORI: -1 the frame cfg.orientation names, picked at run time; 0 / 1 only the forward / reverse frame's code (the
// event kernel is instantiated per frame: half the code, and it is the code's size that a short list pays for)
template <bool UNIFORM_LEN, int NW, int ORI>
DCRX_DEV bool finish2_words(const DevTables &T, const V2Ori &V, const BatchDev &B, const CfgDev &cfg, const uint64_t r,
                            const uint32_t (&w)[NW], const uint32_t (&ev)[3], const bool jmulti, const int x0, const int x1,
                            const Counters &C, dcrx_record_t *records) {
  const uint32_t nw = B.stride >> 2;
  ReadView rv;
  rv.comp = T.comp;
  rv.words = nullptr;
  rv.n = UNIFORM_LEN ? (int)B.read_len : (int)B.lens[r];
  rv.e0 = x0; rv.e1 = x1;
  rv.exc_pos = B.exc_pos; rv.exc_chr = B.exc_chr;
  __align__(16) dcrx_record_t rec;
  rec.v = rec.j = rec.v_start = rec.j_end = rec.ins_start = rec.ins_len = 0;
  rec.vdel = rec.jdel = 0;
  int status, frame;
  if (ORI == 0 || (ORI < 0 && cfg.orientation == DCRX_ORIENT_FORWARD)) {
    status = dcr_frame3(T, V, FrameReg<false, NW>(w, rv), (int)nw, ev, jmulti, cfg, C, rec); frame = 1;
  } else {
    status = dcr_frame3(T, V, FrameReg<true, NW>(w, rv), (int)nw, ev, jmulti, cfg, C, rec); frame = 0;
  }
  if (status == DCRX_S_DEFER) return false;                           // to the three-launch form; nothing counted, nothing written
  C.add(DCRX_C_READ_COUNT);                                           // :991
  if (status == DCRX_S_OK) {
    C.add(DCRX_C_VJ_COUNT);                                           // :1013
    if (frame) C.add(DCRX_C_FRAME_FORWARD);
  }
  rec.status = (uint8_t)status; rec.frame = (uint8_t)frame;
  dcrx_store_record_cached(records + r, rec);      // (its neighbours in the line come from other kernels: merged in the caches)
  return true;
}

// ------------------------------------------------------------------------------
// The lean tail: a read with exactly one V-tag pair whose J side is one J-tag pair, several,
// or none, finished in straight-line code — two rounds of loads (the two tag windows, then the
// two walk windows), keyword look-up by window comparison, the bit-parallel germline walks of
// get_v_deletions / get_j_deletions, the filters.  Whatever does not fit that mould (a window
// that leaves the read, a walk the 32-base form does not settle, two tags inside one pair, ...)
// returns TAIL2_SLOW and takes the general form (dcr_frame3) from the event stack.
// ------------------------------------------------------------------------------
struct Tail2Tabs {       // where the lean tail finds its tables (frame in use); index 0 = V, 1 = J
  uint32_t bk_mask[2];                             // full-tag tables: slots - 1 ...
  dcrx_ldsaddr bk_tag[2], bk_pk[2];                // ... the first tag per slot (uint16; V2_PH_EMPTY: an empty slot), the packed keyword per slot (uint64)
  dcrx_ldsaddr jump[2];                            // int32 per tag
  dcrx_ldsaddr w64[2];                             // uint64 per tag: the walk window as the stored read shows it in this frame
  dcrx_ldsaddr w64_ok[2];                          // uint8 per tag
  uint32_t L[2];                                   // tag length of the class
};

// `side` / `bk`: where the side tables (image[dfa_bytes ..)) and the frame's bucket image were staged
DCRX_DEV Tail2Tabs tail2_tabs(const DevTables &T0, const V2Ori &V0, const uint8_t *side, const uint8_t *bk, const bool rev) {
  Tail2Tabs t;
  auto at_side = [&](const void *p) { return dcrx_ldsaddr_of(side + ((reinterpret_cast<const uint8_t *>(p) - T0.image) - T0.dfa_bytes)); };
  for (int g = 0; g < 2; g++) {
    const int cls = g == 0 ? K_VFULL : K_JFULL;
    t.bk_mask[g] = V0.bk_mask[cls];
    t.bk_tag[g] = dcrx_ldsaddr_of(bk + V0.bk_tag_off[cls]);
    t.bk_pk[g] = dcrx_ldsaddr_of(bk + V0.bk_pk_off[cls]);
    t.jump[g] = at_side(T0.g[g].jump);
    t.w64[g] = at_side(rev ? T0.g[g].w64_rc : T0.g[g].w64_fwd);
    t.w64_ok[g] = at_side(T0.g[g].w64_ok);
    t.L[g] = T0.kw_len[cls];
  }
  return t;
}

constexpr int TAIL2_SLOW = -1;

// 32 bases of the stored read from base s; caller guarantees 0 <= s and s + 32 <= n
DCRX_DEV uint64_t tail2_load64(const dcrx_gwords words, const int s) {
  const int i = s >> 4, sh = (s & 15) * 2;
  const uint32_t w0 = words[i], w1 = words[i + 1];
  const uint32_t w2 = sh ? words[i + 2] : 0u;
  return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
}

// N bucket look-ups in lockstep: the bucket bounds of all of them are requested together, then per round
// one slot of each (packed keyword and its id).  A look-up alone is a chain of dependent LDS reads (bounds,
// slot, id, next slot ...); side by side the chains cost one wait per round for all of them.
// shift count for a window offset in bases (a look-up that is switched off may carry an offset outside the window)
DCRX_DEV int v2_sh(const int bases) { return 2 * min(max(bases, 0), 31); }
struct LookupQ {
  uint32_t mask;                 // the class's slots - 1
  dcrx_ldsaddr ids, pk;          // per slot an id (uint16; V2_PH_EMPTY: an empty slot) and the packed keyword (uint64)
  uint64_t val;                  // the window to find
  bool on;                       // false: no look-up (the result is -1)
};
// N look-ups side by side: the two slots of each are fetched together (keyword and id), one wait for all of them
template <int N>
DCRX_DEV void lookup_lockstep(const LookupQ (&q)[N], int (&out)[N]) {
  uint64_t p1[N], p2[N];
  uint32_t i1[N], i2[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    uint32_t s1, s2;
    v2_slots(q[k].val, q[k].mask, s1, s2);
    p1[k] = dcrx_lds_at<uint64_t>(q[k].pk, s1); p2[k] = dcrx_lds_at<uint64_t>(q[k].pk, s2);
    i1[k] = dcrx_lds_at<uint16_t>(q[k].ids, s1); i2[k] = dcrx_lds_at<uint16_t>(q[k].ids, s2);
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    int r = -1;
    if (p2[k] == q[k].val && i2[k] != V2_PH_EMPTY) r = (int)i2[k];
    if (p1[k] == q[k].val && i1[k] != V2_PH_EMPTY) r = (int)i1[k];
    out[k] = q[k].on ? r : -1;
  }
}

// first tag holding the full-tag keyword whose packed form is `val`, or -1
DCRX_DEV int tail2_lookup(const Tail2Tabs &tt, const int g, const uint64_t val) {
  const LookupQ q[1] = {{tt.bk_mask[g], tt.bk_tag[g], tt.bk_pk[g], val, true}};
  int out[1];
  lookup_lockstep<1>(q, out);
  return out[0];
}

// 32 bases of a read held in registers (NW words) from stored base s >= 0: one select chain per
// word, the three chains share their compares
template <int NW>
DCRX_DEV uint64_t reg_stored64(const uint32_t (&w)[NW], const int s) {
  const int i = s >> 4, sh = (s & 15) * 2;
  uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const bool at = i == k;
    w0 = at ? w[k] : w0;
    if (k + 1 < NW) w1 = at ? w[k + 1] : w1;
    if (k + 2 < NW) w2 = at ? w[k + 2] : w2;
  }
  return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
}

// Where a lean kernel keeps the read in hand: in registers (a 32-base window is a select chain over the
// words, ~45 VALU instructions) or in a per-lane strip of LDS (three ds_read_b32 at lane-varying addresses;
// the strip holds the NW words and two zero words, at an odd stride in words so that the lanes of a wave
// spread over the banks).
template <int NW>
struct RegWords {
  const uint32_t (&w)[NW];
  DCRX_DEV uint64_t stored64(const int s) const { return reg_stored64<NW>(w, s); }
};
struct LdsWords {
  dcrx_ldsaddr strip;      // this lane's first word
  DCRX_DEV uint64_t stored64(const int s) const {
    const uint32_t i = (uint32_t)s >> 4;
    const int sh = (s & 15) * 2;
    const uint32_t w0 = dcrx_lds_at<uint32_t>(strip, i), w1 = dcrx_lds_at<uint32_t>(strip, i + 1), w2 = dcrx_lds_at<uint32_t>(strip, i + 2);
    return (uint64_t)dcrx_funnel_r(w0, w1, sh) | ((uint64_t)dcrx_funnel_r(w1, w2, sh) << 32);
  }
};
template <int NW>
constexpr int lds_words_stride() { return ((NW + 2) | 1); }      // words per lane: NW + 2 zero words, made odd

// A clean read (no exception bytes) behind a word source WS (RegWords / LdsWords), seen in one frame: what the
// walks and comparisons of dcrx_dcr_device.h ask of a frame.  The lean forms hand a walk that leaves its first
// 32-base window to get_v_deletions / get_j_deletions through this (a rare branch: 0.02 % of their entries).
template <bool REV_, class WS>
struct FrameWS {
  static constexpr bool kRev = REV_;
  static constexpr bool kWindowedWalks = true;
  const WS &w;
  int len;
  DCRX_DEV int n() const { return len; }
  DCRX_DEV bool has_exc() const { return false; }
  DCRX_DEV bool clean(int, int) const { return true; }
  DCRX_DEV uint64_t exc_slots(int) const { return 0ull; }
  DCRX_DEV bool has_N(int, int) const { return false; }
  DCRX_DEV uint64_t load64(int b) const { return w.stored64(b); }
  DCRX_DEV uint32_t window(int a, int l) const {
    const int lo = REV_ ? len - a - l : a;
    const uint32_t v = (uint32_t)w.stored64(lo);
    return v & ((l >= 16) ? 0xFFFFFFFFu : ((1u << (2 * l)) - 1u));
  }
  DCRX_DEV int code(int i) const {
    const int m = REV_ ? len - 1 - i : i;
    const int c = (int)(w.stored64(m) & 3ull);
    return REV_ ? (c ^ 3) : c;
  }
  DCRX_DEV uint8_t chr(int i) const { return (uint8_t)("ACGT"[code(i)]); }
};

// One tail entry (`digest`: tail2_pack) on a read whose words sit in registers (the finishing
// kernel loads them one batch ahead: nothing here waits for global memory).  Returns the read's
// status with `rec` filled (status and frame left to the caller), or TAIL2_SLOW with nothing
// decided.  No counter is touched here: the caller tallies by status (each status of this form
// implies its counters).
template <bool REV, class WS>
DCRX_DEV int tail2_fast(const Tail2Tabs &tt, const WS &w, const int n, const uint32_t digest, const CfgDev &cfg,
                        dcrx_record_t &rec, const DevTables &T, const Counters &C) {
  const int vpair = (int)(digest & 0xFFu), jpair = (int)((digest >> 8) & 0xFFu), jc = (int)((digest >> 16) & 3u);
  const int Lv = (int)tt.L[0], Lj = (int)tt.L[1];
  // Straight-line up to the walks: what does not fit the lean form only sets `slow` (every load below stays inside the
  // strip whatever the read looks like), and one branch at the end leaves — a wave pays for its branches, not its flags.
  bool slow = n < 32 || Lv > 31 || Lj > 31;
  const int top = max(n, 32) - 32;                         // the last place a 32-base window can start
  // ---- the windows that hold the tags (both candidate ends of a pair share one window) ----
  const int sva = 2 * vpair - Lv + 1;                      // the V tag starts here (ends at the pair's first base) or one base on
  const int wsv = min(max(sva, 0), top);
  const uint64_t Wv = w.stored64(wsv);
  const int sja = 2 * jpair - Lj + 1;
  const int wsj = min(max(sja, 0), top);
  const uint64_t Wj = w.stored64(wsj);
  const uint64_t mv = (1ull << (2 * (Lv & 31))) - 1ull, mj = (1ull << (2 * (Lj & 31))) - 1ull;
  int v, sv, j = 0, sj = 0;
  {   // both candidate ends of the V pair and of the J pair, side by side
    const LookupQ q[4] = {
        {tt.bk_mask[0], tt.bk_tag[0], tt.bk_pk[0], (Wv >> v2_sh(sva - wsv)) & mv, sva >= 0},
        {tt.bk_mask[0], tt.bk_tag[0], tt.bk_pk[0], (Wv >> v2_sh(sva + 1 - wsv)) & mv, sva + 1 + Lv <= n},
        {tt.bk_mask[1], tt.bk_tag[1], tt.bk_pk[1], (Wj >> v2_sh(sja - wsj)) & mj, jc == 1 && sja >= 0},
        {tt.bk_mask[1], tt.bk_tag[1], tt.bk_pk[1], (Wj >> v2_sh(sja + 1 - wsj)) & mj, jc == 1 && sja + 1 + Lj <= n}};
    int t[4];
    lookup_lockstep<4>(q, t);
    slow |= (t[0] >= 0) == (t[1] >= 0);                   // none (cannot be) or two V tags inside the pair
    v = max(t[0] >= 0 ? t[0] : t[1], 0); sv = t[0] >= 0 ? sva : sva + 1;
    slow |= jc == 1 && (t[2] >= 0) == (t[3] >= 0);
    j = max(t[2] >= 0 ? t[2] : t[3], 0); sj = t[2] >= 0 ? sja : sja + 1;
  }
  // ---- the walk windows ----
  const int jumpv = dcrx_lds_at<int32_t>(tt.jump[0], (uint32_t)v);
  const int vp = REV ? n - sv - Lv : sv;                   // where the tag starts in the frame (hold_v[0][1])
  const int te = vp + jumpv - 1;                           // decombine.py:283-285
  const int fv = te + 1;
  // (a walk whose first window does not lie inside the read, or whose gene has no packed window, takes the general function)
  const bool vedge = !(fv >= 32 && fv < n) || !dcrx_lds_at<uint8_t>(tt.w64_ok[0], (uint32_t)v);
  const uint64_t rwv = w.stored64(min(max(REV ? n - fv : fv - 32, 0), top));
  const int jumpj = dcrx_lds_at<int32_t>(tt.jump[1], (uint32_t)j);
  const int jp = REV ? n - sj - Lj : sj;
  const int ts = jp - jumpj;                               // :407-409
  const bool jedge = !(ts >= 0 && ts + 32 <= n) || !dcrx_lds_at<uint8_t>(tt.w64_ok[1], (uint32_t)j);
  const uint64_t rwj = w.stored64(min(max(REV ? n - ts - 32 : ts, 0), top));
  if (slow) return TAIL2_SLOW;
  // (nothing has been counted up to here, and from here on no path returns TAIL2_SLOW)
  // get_v_deletions (:749-785), the 32-base form; a walk that leaves that window goes on in the general function
  const uint64_t yv = mismatch_slots(rwv, dcrx_lds_at<uint64_t>(tt.w64[0], (uint32_t)v));
  int kv = REV ? first_clean_up(or10_up(yv), 0) : first_clean_down(or10_down(yv), 0);
  if (vedge) kv = -1;
  int end_v = te - kv;
  if (kv < 0) {
    const FrameWS<REV, WS> F{w, n};
    if (!get_v_deletions(T.g[0], F, v, te, end_v, kv, C)) return te >= n ? DCRX_S_V_WALK_FAIL_AT_END : DCRX_S_V_WALK_FAIL;     // :288-290 (counted inside)
  }
  // get_j_deletions (:788-817), the 32-base form (for a read without exactly one J-tag pair the lanes compute on whatever the
  // strip holds and the result is dropped below)
  const int end_of_v = end_v + 1;                          // :547
  const int k0 = end_of_v > ts ? end_of_v - ts : 0;
  const uint64_t yj = mismatch_slots(rwj, dcrx_lds_at<uint64_t>(tt.w64[1], (uint32_t)j));
  int kj = REV ? first_clean_down(or10_down(yj), k0) : first_clean_up(or10_up(yj), k0);
  if (jedge) kj = -1;
  int start_j = ts + kj;
  if (jc == 1 && kj < 0) {
    const FrameWS<REV, WS> F{w, n};
    if (!get_j_deletions(T.g[1], F, j, ts, end_of_v, start_j, kj, C)) return DCRX_S_J_WALK_FAIL;   // :413-418 (j_del_failed counted inside)
  }
  const int jend = jp + Lj;
  // the exits in the reference's order (:402-404, :530-531, then the filters :553-569: a clean read holds no N), picked
  // without branching; a record that is not a decombined read keeps no field
  int status = DCRX_S_OK;
  if (vp + Lv > jend + Lj) status = DCRX_S_F_OVERLAP;
  if (kv > jumpv - Lv || kj > jumpj) status = DCRX_S_F_IMPOSS_DEL;
  if ((vp - jend) >= cfg.lenthreshold) status = DCRX_S_F_TOOLONG;
  if (jc == 2) status = DCRX_S_J_MULTI;
  if (jc == 0) status = DCRX_S_J_NONE;
  int lo, hi;
  pyslice(n, end_v + 1, start_j, lo, hi);                  // read[vdat[1]+1 : jdat[1]] :577
  const bool ok = status == DCRX_S_OK;
  rec.v = (uint16_t)(ok ? v : 0); rec.j = (uint16_t)(ok ? j : 0);
  rec.v_start = (uint16_t)(ok ? vp : 0); rec.j_end = (uint16_t)(ok ? jend : 0);
  rec.ins_start = (uint16_t)(ok ? lo : 0); rec.ins_len = (uint16_t)(ok ? hi - lo : 0);
  rec.vdel = (uint8_t)(ok ? kv : 0); rec.jdel = (uint8_t)(ok ? kj : 0);
  return status;
}

// the counters a status of the lean tail stands for (besides read_count)
DCRX_DEV void tail2_count(const Counters &C, const int status, const bool forward) {
  C.add(DCRX_C_READ_COUNT);
  if (status == DCRX_S_OK) { C.add(DCRX_C_VJ_COUNT); if (forward) C.add(DCRX_C_FRAME_FORWARD); return; }
  if (status == DCRX_S_J_NONE) { C.add(DCRX_C_NO_J_ASSIGNED); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); }
  else if (status == DCRX_S_J_MULTI) { C.add(DCRX_C_MULTIPLE_J_MATCHES); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); }
  else if (status == DCRX_S_J_WALK_FAIL) C.add(DCRX_C_VJ_ASSIGNMENT_FAILED);        // (j_del_failed / v_del_failed were counted by the walk itself)
  else if (status == DCRX_S_F_TOOLONG) C.add(DCRX_C_DCRFILTER_TOOLONG_INTERTAG);
  else if (status == DCRX_S_F_IMPOSS_DEL) C.add(DCRX_C_DCRFILTER_IMPOSS_DELETION);
  else if (status == DCRX_S_F_OVERLAP) C.add(DCRX_C_DCRFILTER_TAG_OVERLAP);
}

// ------------------------------------------------------------------------------
// The lean rescue: an event entry whose V and J sides are each one of
//   - one pair that holds the full tag (as the lean tail resolves it),
//   - no full tag and at most four pairs with a half-tag flag: the half-tag rescue of
//     vanalysis :292-394 / janalysis :420-531 on the hits of those pairs, in findall order,
//   - (J only) no J flag at all, or several J-tag pairs,
// finished without the general form's hit lists: per flagged pair one window of the stored read,
// per base of the pair the bucket look-ups of the two half-tag classes; a half-1 hit tries its
// candidate tags at once (Hamming <= 1 of the whole tag window, then the 32-base walk), half-2
// hits wait in four register slots and are tried only when no half-1 keyword occurred (:337-339).
// Everything else — exception bytes, a window that leaves the read, a walk the 32-base form does
// not settle, a candidate that passes the Hamming test but whose walk fails (the reference then
// goes on with the next candidate) — returns RESCUE2_SLOW with nothing decided and no counter
// touched; the caller sends the entry to the general form (dcr_frame3).
// `errs`: bits 0-1 the V rescue that succeeded (1 = second half matched: verr1, 2 = first half
// matched: verr2), bits 2-3 likewise for J (jerr1 / jerr2) — the counters :318/:370/:445/:504.
// ------------------------------------------------------------------------------
struct Rescue2Tabs {
  Tail2Tabs t;
  uint32_t h_mask[2][2];                     // [gene][half - 1]: the half-tag classes' tables: slots - 1 ...
  dcrx_ldsaddr h_kw[2][2], h_pk[2][2];       // ... class-local keyword per slot (uint16; V2_PH_EMPTY: empty), packed keyword per slot (uint64)
  dcrx_ldsaddr kw_begin, kw_tags;                        // keyword -> its tags, ascending (uint32 CSR)
  dcrx_ldsaddr tag_pk[2];                                // uint64 per tag: the tag as the stored read shows it in this frame
  uint32_t kw_base[2][2];                                // first global keyword id of the class
  uint32_t Lh[2][2];                                     // keyword length of the class
  int32_t split[2];
};

DCRX_DEV Rescue2Tabs rescue2_tabs(const DevTables &T0, const V2Ori &V0, const uint8_t *side, const uint8_t *bk, const bool rev,
                                  const uint32_t (&kw_base)[K_NCLASS]) {
  Rescue2Tabs r;
  r.t = tail2_tabs(T0, V0, side, bk, rev);
  auto at_side = [&](const void *p) { return dcrx_ldsaddr_of(side + ((reinterpret_cast<const uint8_t *>(p) - T0.image) - T0.dfa_bytes)); };
  for (int g = 0; g < 2; g++) {
    for (int h = 0; h < 2; h++) {
      const int cls = g == 0 ? (h == 0 ? K_VH1 : K_VH2) : (h == 0 ? K_JH1 : K_JH2);
      r.h_mask[g][h] = V0.bk_mask[cls];
      r.h_kw[g][h] = dcrx_ldsaddr_of(bk + V0.bk_kw_off[cls]);
      r.h_pk[g][h] = dcrx_ldsaddr_of(bk + V0.bk_pk_off[cls]);
      r.kw_base[g][h] = kw_base[cls];
      r.Lh[g][h] = T0.kw_len[cls];
    }
    r.tag_pk[g] = at_side(rev ? T0.g[g].tag_pk_rc : T0.g[g].tag_pk_fwd);
    r.split[g] = T0.g[g].split;
  }
  r.kw_begin = at_side(T0.kw_begin);
  r.kw_tags = at_side(T0.kw_tags);
  return r;
}

constexpr int RESCUE2_SLOW = -1;
// (host emulation, tools/r2_profile.py: the trips of the lean rescue's loops per entry, from which a wave's trips — the longest
// of its lanes' — are counted without a GPU)
#if defined(DCRX_HOST_EMUL) && defined(DCRX_R2_PROFILE)
void r2_prof(int what, int a);
#define R2P(what, a) r2_prof(what, a)
#else
#define R2P(what, a) ((void)0)
#endif
#ifdef DCRX_R2_REASONS
extern unsigned long long g_r2_reasons[32];
#define R2S(k) (g_r2_reasons[k]++, RESCUE2_SLOW)
#else
#define R2S(k) RESCUE2_SLOW
#endif

// Which gene a half-tag sweep serves: gene G (0 = V, 1 = J) for the whole wave, or (G = -1) picked lane by lane — for an
// entry whose two genes differ in kind (one has its full tag, the other needs the rescue) one sweep then serves the V
// rescues and the J rescues of a wave side by side, the same instructions on per-lane tables, instead of one sweep after
// the other.  The tables' places are picked where they are used (two scalars and the lane's bit), not held per lane.
template <int G>
struct GeneOf {
  const Rescue2Tabs &rt;
  bool j;                                       // G = -1: this lane's gene
  DCRX_DEV uint32_t h_mask(int h) const { return G >= 0 ? rt.h_mask[G < 0 ? 0 : G][h] : (j ? rt.h_mask[1][h] : rt.h_mask[0][h]); }
  DCRX_DEV dcrx_ldsaddr h_kw(int h) const { return G >= 0 ? rt.h_kw[G < 0 ? 0 : G][h] : (j ? rt.h_kw[1][h] : rt.h_kw[0][h]); }
  DCRX_DEV dcrx_ldsaddr h_pk(int h) const { return G >= 0 ? rt.h_pk[G < 0 ? 0 : G][h] : (j ? rt.h_pk[1][h] : rt.h_pk[0][h]); }
  DCRX_DEV uint32_t kw_base(int h) const { return G >= 0 ? rt.kw_base[G < 0 ? 0 : G][h] : (j ? rt.kw_base[1][h] : rt.kw_base[0][h]); }
  DCRX_DEV int Lh(int h) const { return (int)(G >= 0 ? rt.Lh[G < 0 ? 0 : G][h] : (j ? rt.Lh[1][h] : rt.Lh[0][h])); }
  DCRX_DEV dcrx_ldsaddr tag_pk() const { return G >= 0 ? rt.tag_pk[G < 0 ? 0 : G] : (j ? rt.tag_pk[1] : rt.tag_pk[0]); }
  DCRX_DEV int Lt() const { return (int)(G >= 0 ? rt.t.L[G < 0 ? 0 : G] : (j ? rt.t.L[1] : rt.t.L[0])); }
  DCRX_DEV int split() const { return G >= 0 ? rt.split[G < 0 ? 0 : G] : (j ? rt.split[1] : rt.split[0]); }
  DCRX_DEV uint32_t fbit() const { return G >= 0 ? (G == 0 ? V2_F_VH : V2_F_JH) : (j ? V2_F_JH : V2_F_VH); }   // the gene's half-tag flag in a nibble of the log
};

// The candidate tags of one half-tag hit (keyword gk of half `half` of the gene, starting at frame
// position p): the first whose whole tag window is within Hamming distance 1 — the `indices`
// loops of :298-317 / :342-369 / :425-444 / :476-503.  1 with k / q (tag start in the frame),
// 0 none.
template <bool REV, class WS, int G>
DCRX_DEV int rescue2_candidates(const GeneOf<G> &g, const WS &w, const int n, const int half, const uint32_t gk,
                                const int p, int &k_out, int &q_out) {
  const Rescue2Tabs &rt = g.rt;
  const int L = g.Lt();
  const int q = half == 1 ? p : p - g.split();
  // a window that leaves the read: the reference's slice read[q:q+L] then comes out shorter than the tag (empty when q < 0
  // and q + L >= 0, as n > L) and the candidates are passed over (:302-307 and siblings); q + L < 0 would wrap around
  if (q + L < 0) return R2S(1);
  if (q < 0 || q + L > n) return 0;
  const int b = REV ? n - q - L : q;                           // where the window starts in the stored read
  const int ws = min(b, n - 32);
  const uint64_t mask = (1ull << (2 * L)) - 1ull;
  const uint64_t val = (w.stored64(ws) >> (2 * (b - ws))) & mask;
  const uint32_t x0 = dcrx_lds_at<uint32_t>(rt.kw_begin, gk), x1 = dcrx_lds_at<uint32_t>(rt.kw_begin, gk + 1);
  const dcrx_ldsaddr tpk = g.tag_pk();
  int found = 0;
  for (uint32_t x = x0; x < x1 && !found; x++) {
    R2P(5, half);
    const int k = (int)dcrx_lds_at<uint32_t>(rt.kw_tags, x);
    const uint64_t y = mismatch_slots(val, dcrx_lds_at<uint64_t>(tpk, (uint32_t)k)) & mask;
    if (dcrx_popc64(y) <= 1) { found = 1; k_out = k; q_out = q; }
  }
  return found;
}

// word kk of a flag log held in registers (a lane-varying index: a select chain over the words)
template <int NW>
DCRX_DEV uint32_t log_word(const uint32_t (&lg)[NW], const int kk) {
  uint32_t l = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) l = (kk == k) ? lg[k] : l;
  return l;
}

// The half-tag rescue of a gene over the pairs whose nibble of the flag log has the gene's half-tag bit, however many
// there are.
// 1: a candidate passed the Hamming test (k, q, p = the keyword's frame start, half); 0: none did
// (half = the list the reference walks: 1 when a half-1 keyword occurred, else 2, 0 when no half-tag
// keyword occurred at all); RESCUE2_SLOW.
// One sweep in findall order (ascending end position in the frame: the words of the log, and the pairs inside a word,
// downwards in the reverse frame): a half-1 hit tries its candidates at once and a success ends the sweep; half-2 hits wait in
// four register slots and are tried only when no half-1 keyword occurred — the reference consults the half-2 list only
// then (:337-339 / :471-473).
template <bool REV, int NW, class WS, int G>
DCRX_DEV int rescue2_half(const GeneOf<G> &g, const WS &w, const uint32_t (&lg)[NW], const int n,
                          int &k_out, int &q_out, int &p_out, int &half_out) {
  const int L1 = g.Lh(0), L2 = g.Lh(1);
  const uint64_t m1 = (1ull << (2 * L1)) - 1ull, m2 = (1ull << (2 * L2)) - 1ull;
  const uint32_t mask8 = g.fbit() * 0x11111111u;
  uint32_t nz = 0;                       // the words of the log that hold such a pair
#pragma unroll
  for (int kk = 0; kk < NW; kk++) nz |= (lg[kk] & mask8) ? (1u << kk) : 0u;
  uint64_t h2lo = 0, h2hi = 0;           // half-2 hits in findall order: slot i = (keyword + 1) << 16 | stored end base
  int h2n = 0;
  bool any1 = false;
  int res = 0;
  R2P(1, G);
  while (nz && res == 0) {
    R2P(2, 0);
    const int kk = REV ? 31 - dcrx_clz32(nz) : dcrx_ctz32(nz);
    nz &= ~(1u << kk);
    uint32_t m = log_word<NW>(lg, kk) & mask8;
    while (m && res == 0) {
      R2P(3, 0);
      const int bit = REV ? 31 - dcrx_clz32(m) : dcrx_ctz32(m);
      m &= ~(1u << bit);
      const int pair = 8 * kk + (bit >> 2);
      const int f1 = 2 * pair + 1;                                 // the pair's second stored base
      const int xs = f1 >= 31 ? f1 - 31 : 0;                       // window: stored bases [xs, xs + 32)
      const uint64_t X = w.stored64(xs);
      for (int y = 0; y < 2 && res == 0; y++) {
        const int f = REV ? f1 - y : f1 - 1 + y;                   // ascending end position in the frame
        if (f >= n) continue;
        R2P(4, y);
        const int s1 = f - L1 + 1, s2 = f - L2 + 1;
        const LookupQ q[2] = {{g.h_mask(0), g.h_kw(0), g.h_pk(0), (X >> v2_sh(s1 - xs)) & m1, s1 >= 0},
                              {g.h_mask(1), g.h_kw(1), g.h_pk(1), (X >> v2_sh(s2 - xs)) & m2, s2 >= 0}};
        int kw[2];
        lookup_lockstep<2>(q, kw);
        const int kw1 = kw[0], kw2 = kw[1];
        if (kw2 >= 0) {
          const uint64_t e = ((uint64_t)(uint32_t)(kw2 + 1) << 16) | (uint64_t)(uint32_t)f;
          if (h2n < 2) h2lo |= e << (32 * h2n); else if (h2n < 4) h2hi |= e << (32 * (h2n - 2));
          h2n++;
        }
        if (kw1 >= 0) {
          R2P(8, 0);
          any1 = true;
          const int p = REV ? n - s1 - L1 : s1;
          res = rescue2_candidates<REV, WS, G>(g, w, n, 1, g.kw_base(0) + (uint32_t)kw1, p, k_out, q_out);
          p_out = p;
        }
      }
    }
  }
  half_out = 1;
  if (res != 0 || any1) return res;
  if (h2n > 4) return R2S(13);
  half_out = h2n ? 2 : 0;
  for (int i = 0; i < h2n && res == 0; i++) {
    R2P(6, 0);
    const uint32_t e = (uint32_t)((i < 2 ? h2lo : h2hi) >> (32 * (i & 1)));
    const int f = (int)(e & 0xFFFFu), kw2 = (int)(e >> 16) - 1;
    const int s2 = f - L2 + 1;
    const int p = REV ? n - s2 - L2 : s2;
    res = rescue2_candidates<REV, WS, G>(g, w, n, 2, g.kw_base(1) + (uint32_t)kw2, p, k_out, q_out);
    p_out = p;
  }
  return res;
}

// The shapes of an event entry, as the scan kernel sorts them:
//   V2_SHAPE_ONE   one gene has its full tag (or, J: several, or no flag at all) and the other needs the half-tag rescue:
//                  the full tag by look-up as in the lean tail, ONE sweep for the other gene, the gene picked lane by lane
//   V2_SHAPE_BOTH  no V tag, no J tag, half-tag flags of both genes: two sweeps
//   V2_SHAPE_ANY   (the test build) decided per read
enum { V2_SHAPE_ANY = -1, V2_SHAPE_ONE = 0, V2_SHAPE_BOTH = 1 };
DCRX_DEV int shape2(const uint32_t vf_n, const uint32_t jf_n, const uint32_t any) {
  return (vf_n == 0u && jf_n == 0u && (any & V2_F_JH)) ? V2_SHAPE_BOTH : V2_SHAPE_ONE;
}

// T: the tables in global memory (a walk that leaves its first window reads the packed regions there);
// C: the block's counters (a full-tag walk that fails is final and counts inside the walk); Cdry: a scratch block of
// counters nobody reads (a half-tag candidate's walk that fails is not final: the read then takes the general form, and
// nothing may have been counted).
// what the scan's digest of a flag log (Digest2) says, in one word of the event entry: the lean rescue then skips its own pass
// over the log's words (a hundred instructions per batch).  RESCUE2_NO_DIGEST: none given.
constexpr uint32_t RESCUE2_NO_DIGEST = 0xFFFFFFFFu;
DCRX_DEV uint32_t rescue2_digest_pack(const Digest2 &d) {
  return min(d.vf_n, 3u) | (min(d.jf_n, 3u) << 2) | ((d.any & 0xFu) << 4) | ((d.vf_pair & 0xFFu) << 8) | ((d.jf_pair & 0xFFu) << 16);
}
// rescue2_fast_to: the fields of a decombined read go to `on_ok` where they are known (the kernel stores the record there): a
// caller that took them back as results paid for it at every early exit — the function has two dozen, and each nesting level
// set all eight fields, the rescue bits and the tuple to their defaults again (15 moves a level: a fifth of the lean rescue's
// vector instructions were such moves).
template <bool REV, int NW, int SHAPE, class WS, class OnOk>
DCRX_DEV int rescue2_fast_to(const Rescue2Tabs &rt, const WS &w, const uint32_t (&lg)[NW], const int n, const CfgDev &cfg,
                             OnOk &&on_ok, uint32_t &errs, const DevTables &T, const Counters &C, const Counters &Cdry,
                             const uint32_t dg = RESCUE2_NO_DIGEST) {
  const Tail2Tabs &tt = rt.t;
  const int Lv = (int)tt.L[0], Lj = (int)tt.L[1];
  errs = 0;
  if (n < 32 || Lv > 31 || Lj > 31) return R2S(2);
  // ---- what the flag log holds: pairs with a V tag / a J tag (number and the first), any half-tag flag ----
  uint32_t vfn = 0, jfn = 0, any = 0, vf1 = 0xFFFFFFFFu, jf1 = 0xFFFFFFFFu;
  if (dg != RESCUE2_NO_DIGEST) {         // (wave-uniform: a list's entries all carry one, or none does)
    vfn = dg & 3u; jfn = (dg >> 2) & 3u; any = (dg >> 4) & 0xFu;
    vf1 = ((dg >> 8) & 0xFFu) << 2; jf1 = ((dg >> 16) & 0xFFu) << 2;
  } else {
#pragma unroll
    for (int kk = 0; kk < NW; kk++) {
      const uint32_t l = lg[kk];
      any |= l;
      if (SHAPE != V2_SHAPE_BOTH) {
        const uint32_t tv = l & 0x11111111u, tj = l & 0x22222222u;
        const uint32_t kb = (uint32_t)kk << 5;
        vfn += (uint32_t)dcrx_popc64(tv);
        vf1 = min(vf1, (tv ? (uint32_t)dcrx_ctz32(tv) : 0xFFFFFFFFu) | kb);
        jfn += (uint32_t)dcrx_popc64(tj);
        jf1 = min(jf1, (tj ? (uint32_t)dcrx_ctz32(tj) : 0xFFFFFFFFu) | kb);
      }
    }
    any |= any >> 16; any |= any >> 8; any |= any >> 4; any &= 0xFu;
  }
  if ((n & 1) && log_nibble<NW>(lg, n >> 1)) return R2S(3);       // a flag on the half pair at the end of an odd-length read may not stand
  if (vfn > 1 || (vfn == 0 && !(any & V2_F_VH))) return R2S(4);   // (entries of the scan kernel never look like this)
  const bool vfull = vfn == 1;
  const bool jsweep = jfn == 0 && (any & V2_F_JH) != 0u;          // J by half-tag rescue (else: one J tag, several, or no J flag at all)
  if (SHAPE == V2_SHAPE_ONE && vfull != jsweep) return R2S(4);    // (exactly one gene is swept in this shape: V unless it has its tag, then J)
  if (SHAPE == V2_SHAPE_BOTH && (vfull || !jsweep)) return R2S(4);
  const uint64_t mv = (1ull << (2 * Lv)) - 1ull, mj = (1ull << (2 * Lj)) - 1ull;

  // ---- the full tag of the gene that has one (its pair: both candidate ends) ----
  int ftag = -1, fs = 0;                 // tag and where it starts in the stored read
  const bool jfull = !vfull && jfn == 1;
  if (SHAPE != V2_SHAPE_BOTH && (vfull || jfull)) {
    const int Lf = vfull ? Lv : Lj;
    const uint64_t mf = vfull ? mv : mj;
    const int pair = (int)((vfull ? vf1 : jf1) >> 2);
    const int sa = 2 * pair - Lf + 1;
    const int ws = min(max(sa, 0), n - 32);
    const uint64_t W = w.stored64(ws);
    const uint32_t bs = vfull ? tt.bk_mask[0] : tt.bk_mask[1];
    const dcrx_ldsaddr bt = vfull ? tt.bk_tag[0] : tt.bk_tag[1], bp = vfull ? tt.bk_pk[0] : tt.bk_pk[1];
    const LookupQ q[2] = {{bs, bt, bp, (W >> v2_sh(sa - ws)) & mf, sa >= 0}, {bs, bt, bp, (W >> v2_sh(sa + 1 - ws)) & mf, sa + 1 + Lf <= n}};
    int t[2];
    lookup_lockstep<2>(q, t);
    if ((t[0] >= 0) == (t[1] >= 0)) return vfull ? R2S(5) : R2S(9);
    ftag = t[0] >= 0 ? t[0] : t[1];
    fs = t[0] >= 0 ? sa : sa + 1;
  }
  // ---- the sweeps (nothing is counted here: what they find is used below in the reference's order, V before J) ----
  int vres = 0, vk = 0, vq = 0, vpp = 0, vhalf = 0, jres = 0, jk = 0, jq = 0, jpp = 0, jhalf = 0;
  if (SHAPE == V2_SHAPE_ONE) {
    const GeneOf<-1> g{rt, vfull};                     // the J rescue where V has its tag, else the V rescue
    int k = 0, q = 0, p = 0, half = 0;
    const int res = rescue2_half<REV, NW, WS, -1>(g, w, lg, n, k, q, p, half);
    if (vfull) { jres = res; jk = k; jq = q; jpp = p; jhalf = half; } else { vres = res; vk = k; vq = q; vpp = p; vhalf = half; }
  } else {
    if (!vfull) vres = rescue2_half<REV, NW, WS, 0>(GeneOf<0>{rt, false}, w, lg, n, vk, vq, vpp, vhalf);
    if (vres < 0) return R2S(6);
    if (jsweep && (vfull || vres > 0)) jres = rescue2_half<REV, NW, WS, 1>(GeneOf<1>{rt, true}, w, lg, n, jk, jq, jpp, jhalf);
  }

  // ---- vanalysis ----
  int v = -1, vp = 0, te = 0;
  if (vfull) {
    v = ftag;
    vp = REV ? n - fs - Lv : fs;
    te = vp + dcrx_lds_at<int32_t>(tt.jump[0], (uint32_t)v) - 1;                       // :283-285
  } else {
    if (vres < 0) return R2S(6);
    if (vres == 0) return vhalf == 1 ? DCRX_S_V_HALF1_EXHAUSTED : (vhalf == 2 ? DCRX_S_V_HALF2_EXHAUSTED : DCRX_S_V_NONE);   // :334 / :389 / :393
    v = vk; vp = vq;
    const int jump = dcrx_lds_at<int32_t>(tt.jump[0], (uint32_t)vk);
    te = vhalf == 1 ? vpp + jump - 1 : vpp + jump - rt.split[0] - 1;                  // :320-322 / :372-377
    errs |= vhalf == 1 ? 2u : 1u;
  }
  const int jumpv = dcrx_lds_at<int32_t>(tt.jump[0], (uint32_t)v);
  const int fv = te + 1;
  // (a walk whose first window does not lie inside the read, or whose gene has no packed window, takes the general function)
  const bool vedge = !(fv >= 32 && fv < n) || !dcrx_lds_at<uint8_t>(tt.w64_ok[0], (uint32_t)v);
  const uint64_t rwv = w.stored64(min(max(REV ? n - fv : fv - 32, 0), n - 32));
  const uint64_t yv = mismatch_slots(rwv, dcrx_lds_at<uint64_t>(tt.w64[0], (uint32_t)v));
  int kv = REV ? first_clean_up(or10_up(yv), 0) : first_clean_down(or10_down(yv), 0);
  if (vedge) kv = -1;
  int end_v = te - kv;
  // (a full-tag walk that fails is final and counts inside the walk: whatever may still send the read to the general form
  // because of the sweep has to be looked at before; the J side's own checks below come before anything they could count)
  if (jsweep && jres < 0) return R2S(10);
  if (kv < 0) {      // the walk leaves its first window: the general function (rare)
    const FrameWS<REV, WS> F{w, n};
    if (!get_v_deletions(T.g[0], F, v, te, end_v, kv, vfull ? C : Cdry)) {
      if (!vfull) return R2S(8);                       // the reference goes on with the next candidate: the general form
      return te >= n ? DCRX_S_V_WALK_FAIL_AT_END : DCRX_S_V_WALK_FAIL;      // :288-290
    }
  }
  const int end_of_v = end_v + 1;                                                     // :547

  // ---- janalysis ----
  int j = -1, jend = 0, ts = 0;
  if (!jsweep) {
    if (jfn >= 2) return DCRX_S_J_MULTI;                                              // :402-404
    if (jfn == 0) return DCRX_S_J_NONE;                                               // :530-531 (no J flag of any kind)
    j = ftag;
    const int jp = REV ? n - fs - Lj : fs;
    ts = jp - dcrx_lds_at<int32_t>(tt.jump[1], (uint32_t)j);                          // :407-409
    jend = jp + Lj;
  } else {
    if (jres == 0) return jhalf == 1 ? DCRX_S_J_HALF1_EXHAUSTED : (jhalf == 2 ? DCRX_S_J_HALF2_EXHAUSTED : DCRX_S_J_NONE);   // :469 / :526 / :530
    j = jk;
    const int jump = dcrx_lds_at<int32_t>(tt.jump[1], (uint32_t)jk);
    ts = jhalf == 1 ? jpp - jump : jpp - jump - rt.split[1];                          // :447-449 / :506-510
    jend = jhalf == 1 ? jpp + (int)rt.Lh[1][0] + rt.split[1] : jpp + (int)rt.Lh[1][1];   // :450-454 / :511
    errs |= jhalf == 1 ? 8u : 4u;
  }
  const int jumpj = dcrx_lds_at<int32_t>(tt.jump[1], (uint32_t)j);
  const bool jedge = !(ts >= 0 && ts + 32 <= n) || !dcrx_lds_at<uint8_t>(tt.w64_ok[1], (uint32_t)j);
  const uint64_t rwj = w.stored64(min(max(REV ? n - ts - 32 : ts, 0), n - 32));
  const int k0 = end_of_v > ts ? end_of_v - ts : 0;
  const uint64_t yj = mismatch_slots(rwj, dcrx_lds_at<uint64_t>(tt.w64[1], (uint32_t)j));
  int kj = REV ? first_clean_down(or10_down(yj), k0) : first_clean_up(or10_up(yj), k0);
  if (jedge) kj = -1;
  int start_j = ts + kj;
  if (kj < 0) {
    const FrameWS<REV, WS> F{w, n};
    if (!get_j_deletions(T.g[1], F, j, ts, end_of_v, start_j, kj, jsweep ? Cdry : C)) {
      if (jsweep) return R2S(12);
      return DCRX_S_J_WALK_FAIL;                       // :413-418
    }
  }
  // ---- filters :553-569 (a clean read holds no N) ----
  if ((vp - jend) >= cfg.lenthreshold) return DCRX_S_F_TOOLONG;
  if (kv > jumpv - Lv || kj > jumpj) return DCRX_S_F_IMPOSS_DEL;
  if (vp + Lv > jend + Lj) return DCRX_S_F_OVERLAP;
  int lo, hi;
  pyslice(n, end_v + 1, start_j, lo, hi);                                             // read[vdat[1]+1 : jdat[1]] :577
  dcrx_record_t rec;
  rec.v = (uint16_t)v; rec.j = (uint16_t)j;
  rec.v_start = (uint16_t)vp; rec.j_end = (uint16_t)jend;
  rec.ins_start = (uint16_t)lo; rec.ins_len = (uint16_t)(hi - lo);
  rec.vdel = (uint8_t)kv; rec.jdel = (uint8_t)kj;
  rec.status = (uint8_t)DCRX_S_OK; rec.frame = 0;
  on_ok(rec, errs);
  return DCRX_S_OK;
}
// ... and with the fields handed back in `rec` (untouched unless the read decombined)
template <bool REV, int NW, int SHAPE, class WS>
DCRX_DEV int rescue2_fast(const Rescue2Tabs &rt, const WS &w, const uint32_t (&lg)[NW], const int n, const CfgDev &cfg,
                          dcrx_record_t &rec, uint32_t &errs, const DevTables &T, const Counters &C, const Counters &Cdry,
                          const uint32_t dg = RESCUE2_NO_DIGEST) {
  return rescue2_fast_to<REV, NW, SHAPE>(rt, w, lg, n, cfg, [&](const dcrx_record_t &r, const uint32_t) {
    rec.v = r.v; rec.j = r.j; rec.v_start = r.v_start; rec.j_end = r.j_end; rec.ins_start = r.ins_start; rec.ins_len = r.ins_len;
    rec.vdel = r.vdel; rec.jdel = r.jdel; }, errs, T, C, Cdry, dg);
}

// the counters a status of the lean rescue stands for (besides read_count); errs as rescue2_fast leaves them
DCRX_DEV void rescue2_count(const Counters &C, const int status, const uint32_t errs, const bool forward) {
  C.add(DCRX_C_READ_COUNT);
  if (errs & 1u) C.add(DCRX_C_VERR1);
  if (errs & 2u) C.add(DCRX_C_VERR2);
  if (errs & 4u) C.add(DCRX_C_JERR1);
  if (errs & 8u) C.add(DCRX_C_JERR2);
  switch (status) {
    case DCRX_S_OK: C.add(DCRX_C_VJ_COUNT); if (forward) C.add(DCRX_C_FRAME_FORWARD); break;
    case DCRX_S_V_HALF1_EXHAUSTED: C.add(DCRX_C_FOUNDV1NOTV2); break;
    case DCRX_S_V_HALF2_EXHAUSTED: C.add(DCRX_C_FOUNDV2NOTV1); break;
    case DCRX_S_V_NONE: C.add(DCRX_C_NO_VTAGS_FOUND); break;
    case DCRX_S_J_MULTI: C.add(DCRX_C_MULTIPLE_J_MATCHES); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); break;
    case DCRX_S_J_NONE: C.add(DCRX_C_NO_J_ASSIGNED); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); break;
    case DCRX_S_J_HALF1_EXHAUSTED: C.add(DCRX_C_FOUNDJ1NOTJ2); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); break;
    case DCRX_S_J_HALF2_EXHAUSTED: C.add(DCRX_C_FOUNDV2NOTV1); C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); break;   // the reference bumps the V key (:526)
    case DCRX_S_J_WALK_FAIL: C.add(DCRX_C_VJ_ASSIGNMENT_FAILED); break;     // (j_del_failed / v_del_failed were counted by the walk itself)
    case DCRX_S_F_TOOLONG: C.add(DCRX_C_DCRFILTER_TOOLONG_INTERTAG); break;
    case DCRX_S_F_IMPOSS_DEL: C.add(DCRX_C_DCRFILTER_IMPOSS_DELETION); break;
    case DCRX_S_F_OVERLAP: C.add(DCRX_C_DCRFILTER_TAG_OVERLAP); break;
    default: break;
  }
}

}  // namespace dcrx
