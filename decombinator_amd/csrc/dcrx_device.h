// dcrx_device.h — the table image as the kernels see it: resolved device
// pointers into the single blob that dcrx::compile_tables laid out.
#pragma once

#include <cstdint>

namespace dcrx {

// output classes of the merged automaton (one per reference automaton)
enum KwClass { K_VFULL = 0, K_JFULL = 1, K_VH1 = 2, K_VH2 = 3, K_JH1 = 4, K_JH2 = 5, K_NCLASS = 6 };

// ---- transition entry layout (uint32) -------------------------------------
// bits 0..17  byte offset of the target state's row (state * 16)
// bit  18     >=1 V tag ends at the target state          (v_key.findall hit)
// bit  19     >=1 J tag ends there                         (j_key)
// bit  20..23 a V half1 / V half2 / J half1 / J half2 keyword ends there
// bit  24     >=2 V tags end there (unequal-length tag sets only)
// bit  25     >=2 J tags end there
constexpr uint32_t TE_ROW_MASK = 0x3FFFFu;
constexpr int TE_VFULL_BIT = 18, TE_JFULL_BIT = 19;
constexpr int TE_VH1_BIT = 20, TE_VH2_BIT = 21, TE_JH1_BIT = 22, TE_JH2_BIT = 23;
constexpr int TE_VMULTI_BIT = 24, TE_JMULTI_BIT = 25;
constexpr uint32_t MAX_STATES = 16383;

// ---- two-bases-per-step table (fast kernel): entry for (state, first base * 4 + second base) ------
// bits 0..17  byte offset of the row of the state after both bases (state * 64)
// bit  18/19  a V / J tag ends at the first or the second base of the pair
// bit  20..23 half-tag classes ending at either base (OR)
// bit  24/25  two or more V / J tags end inside the pair
// bit  26/27  the V / J tag of bit 18/19 ends at the SECOND base
constexpr int TE16_V2_BIT = 26, TE16_J2_BIT = 27;
constexpr int TE16_H2_SHIFT = 28;   // pair entries: bits 20..23 = half-tag classes (VH1 VH2 JH1 JH2) that end at the FIRST base of the
                                    // pair, bits 28..31 = those that end at the SECOND base
// OR-ed pair entries -> the one-base table's flag layout (half-tag classes of either base in bits 20..23)
#define DCRX_ACC16_FLAGS(acc) (((acc) | ((((acc) >> dcrx::TE16_H2_SHIFT) & 0xFu) << dcrx::TE_VH1_BIT)) & 0x03FFFFFFu)
constexpr uint32_t MAX_STATES16 = 4095;
constexpr uint32_t MAX_TAG_LEN = 32;

// entry of the per-state output list: class | len<<3 | kw<<9 | last<<31
inline uint32_t out_pack(int cls, int len, int kw, bool last) {
  return (uint32_t)cls | ((uint32_t)len << 3) | ((uint32_t)kw << 9) | (last ? 0x80000000u : 0u);
}

// ---- v2 scan (dcrx_v2_device.h): forward-only filter automaton + event log -------------------
// One table per frame.  The stored (FASTQ-frame) read is always scanned forwards from its first
// base: for the forward frame the automaton holds the keywords themselves, for the reverse frame
// their reverse complements (a keyword K occurs in revcomp(read) exactly where revcomp(K) occurs
// in the read).  Entry (uint16) for (state, raw nibble of the packed read = first base | second
// base << 2): next state << 4 | V2_F_* flags "a keyword of that group ends at the first or the
// second base of the pair".  The flags select the reads and positions that are then resolved by
// comparing the read's window with the packed keywords (bucket tables below): the automaton is
// the filter, the comparison is what decides.
constexpr uint32_t V2_MAX_STATES = 4095;
constexpr uint32_t V2_F_VF = 1u, V2_F_JF = 2u, V2_F_VH = 4u, V2_F_JH = 8u;
// The keyword tables: per class a two-choice (cuckoo) hash table of S = 2^k slots — a packed keyword lies in one of the two
// slots its hash names, so a look-up is two probes side by side and one wait: no chain to walk, no bounds to fetch first (the
// 64-bucket chains of rounds 2-4 cost a look-up two dependent LDS round trips and a loop over the longest chain of the
// look-ups in flight).  An empty slot carries the id V2_PH_EMPTY.
constexpr uint32_t V2_PH_EMPTY = 0xFFFFu;
constexpr uint32_t V2_PH_MAX_SLOTS = 8192;
// the two slots of a packed keyword / read window (2 bits per base, first base lowest) in a table of mask + 1 slots
inline
#ifdef __HIPCC__
__host__ __device__
#endif
void v2_slots(const uint64_t v, const uint32_t mask, uint32_t &s1, uint32_t &s2) {
  // three 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 run at full rate, v_mul_lo_u32 at a quarter of it): the key's
  // bits 0-23, 24-47 and 48-63 each times an odd 24-bit constant, summed.  A product's low bits know only the low bits of
  // the key: both slots are cut from the sum's upper half.
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  const uint32_t a = lo & 0xFFFFFFu, b = ((lo >> 24) | (hi << 8)) & 0xFFFFFFu, c = hi >> 16;
  const uint32_t s = a * 0x9E3779u + b * 0x85EBCBu + c * 0xC2B2AFu;      // (24-bit operands: the device compiles these to the u24 forms)
  s1 = (s >> 19) & mask;
  s2 = ((s >> 12) ^ (s >> 25)) & mask;
}

struct V2Ori {
  const uint16_t *trans;          // [n_states][16]
  uint32_t trans_bytes;
  uint32_t n_states;
  uint32_t narrow;                // 1: entries are state << 5 | flags (<= 2047 states: the state bits are the row's byte offset); 0: state << 4 | flags
  const uint8_t *bk;              // keyword tables' image (staged in LDS behind the side tables)
  uint32_t bk_bytes;
  // per keyword class: slots - 1 (a power of two), and byte offsets inside bk of the slots' packed keywords as the stored read
  // shows them (uint64[slots]), of their class-local keyword indices (uint16[slots]; V2_PH_EMPTY: an empty slot) and of the
  // first tag that holds each keyword (uint16[slots]: list.index, decombine.py:282 / :406)
  uint32_t bk_mask[K_NCLASS];
  uint32_t bk_kw_off[K_NCLASS];
  uint32_t bk_pk_off[K_NCLASS];
  uint32_t bk_tag_off[K_NCLASS];
};

struct GeneDevPtrs {
  uint32_t n;
  int32_t split;               // v_half_split / j_half_split (decombine.py:657-661)
  const uint8_t *tag_len;      // len(v_seqs[k])
  const int32_t *jump;         // jump_to_end_v[k] / jump_to_start_j[k]
  const uint8_t *tag_ascii;    // [k*32], the tag's characters
  const uint64_t *tag_pk_fwd;  // tag 2-bit packed: slot s = tag[s]
  const uint64_t *tag_pk_rc;   // slot y = complement of tag[len-1-y] (the tag as the packed FORWARD read shows it in the reverse frame)
  const uint32_t *reg_off;     // byte offset of region k inside reg_bytes
  const uint32_t *reg_len;     // len(v_regions[k])
  const uint8_t *reg_bytes;    // upper-cased germline characters
  const uint32_t *reg_pk_off;  // word offset of region k inside reg_pk / reg_pk_rc
  const uint32_t *reg_pk;      // 2-bit packed region (base i at bits 2(i%16) of word i/16)
  const uint32_t *reg_pk_rc;   // 2-bit packed reverse complement of the region
  const uint8_t *reg_clean;    // 1 when the region is pure ACGT (packed compare allowed)
  // 32-base window at the end the walk starts from (V: last 32 nt, J: first 32 nt), for
  // the bit-parallel walk: w64_fwd slot s (bits 2s..2s+1) = window base s; w64_rc slot y =
  // complement of window base 31-y (the window as the packed FORWARD read shows it when the
  // frame is the reverse complement); w64_ok = region has >= 32 nt and the window is pure ACGT
  const uint64_t *w64_fwd;
  const uint64_t *w64_rc;
  const uint8_t *w64_ok;
};

struct DevTables {
  uint32_t n_states;
  uint32_t first_out;         // states >= first_out are the ones where some keyword ends
  uint32_t row0;              // what a transition entry's row field is relative to: 0 in global memory,
                              // the staged table's LDS address inside a kernel
  uint32_t dfa_bytes;
  uint32_t lds_image_bytes;   // image[0 .. lds_image_bytes) = DFA + side tables: what the kernels stage in LDS
  uint32_t lds_image2_bytes;  // image[lds_image_bytes .. lds_image2_bytes) = packed germline regions (+ offsets, purity flags) of both genes
  const uint8_t *image;       // start of the table blob (== trans)
  const uint32_t *trans;      // [n_states*4] transition entries
  const uint32_t *trans16;    // [n_states*16] two-bases-per-step entries (null when the automaton has > 4095 states)
  uint32_t dfa16_bytes;
  uint32_t row16_0;           // like row0, for trans16
  uint32_t max_half_len;      // longest half-tag keyword (the rescue kernel re-derives hit states from a window of 16 bases)
  uint32_t pair_rescue;       // 1 when the rescue kernel's pair form applies: trans16 exists, max_half_len <= 15, every keyword >= 2 nt
  const uint32_t *st_full;    // [state - first_out] V tag | J tag << 16 ending at the state (0xFFFF none)
  const uint32_t *st_out;     // [state - first_out (+1)] CSR into outs
  const uint32_t *outs;       // per-state output list, longest keyword first
  const uint32_t *kw_base;    // [K_NCLASS] first global keyword id of each class
  const uint32_t *kw_first;   // [n_kw_total] first tag index holding the keyword (list.index)
  const uint32_t *kw_begin;   // [n_kw_total+1] CSR into kw_tags
  const uint32_t *kw_tags;    // tag indices holding the keyword, ascending
  const uint8_t *comp;        // [256] Biopython ambiguous-DNA complement, both cases (decombine.py:184)
  GeneDevPtrs g[2];           // 0 = V, 1 = J
  // v2 scan: built when every class has keywords of one length and each frame's automaton has <= V2_MAX_STATES states
  uint32_t v2_ok;
  uint32_t kw_len[K_NCLASS];  // the common keyword length of each class (v2_ok)
  V2Ori v2[2];                // [0] forward frame, [1] reverse frame
};

}  // namespace dcrx
