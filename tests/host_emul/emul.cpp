// emul.cpp — TEST-ONLY host build of the per-read device code
// (decombinator_amd/csrc/dcrx_dcr_device.h) so that the exact functions the HIP
// kernel runs can be checked against the oracle, and run under ASan/UBSan,
// without a GPU.  It is never linked into libdcrx.so and nothing under
// decombinator_amd/ refers to it.  It does NOT replace the GPU parity tests:
// the kernel's launch geometry, LDS staging, counter reduction and the real
// ISA are only exercised by `pytest -m gpu`.
#define DCRX_HOST_EMUL 1
#define DCRX_R2_REASONS 1
#include <cstring>
#include <string>
#include <vector>

#include "../../decombinator_amd/csrc/dcrx_dcr_device.h"
#include "../../decombinator_amd/csrc/dcrx_v2_device.h"
#include "../../decombinator_amd/csrc/dcrx_tables.h"

using namespace dcrx;

template <bool UNIFORM>
static int fast_one(bool pair_scan, const DevTables &T, const BatchDev &B, const CfgDev &C, uint64_t r, uint32_t nw,
                    const Counters &CC, dcrx_record_t *records) {
  if (pair_scan)
    return B.stride <= 40 ? decombine_fast_one<false, UNIFORM, 10, 16>(T, nullptr, B, C, r, nw, CC, records)
                          : decombine_fast_one<false, UNIFORM, DCRX_NWMAX, 16>(T, nullptr, B, C, r, nw, CC, records);
  return B.stride <= 40 ? decombine_fast_one<false, UNIFORM, 10, 4>(T, nullptr, B, C, r, nw, CC, records)
                        : decombine_fast_one<false, UNIFORM, DCRX_NWMAX, 4>(T, nullptr, B, C, r, nw, CC, records);
}

// a deferred clean read: the rescue kernel's pair form when the launch would use it, else the list kernel's
template <bool UNIFORM>
static void rescue_one(bool pair_rescue, const DevTables &T, const BatchDev &B, const CfgDev &C, uint64_t r, uint32_t nw,
                       const Counters &CC, dcrx_record_t *records, uint32_t *slot) {
  if (!pair_rescue) { decombine_list_one<false, UNIFORM>(T, nullptr, B, C, r, CC, records, slot, false); return; }
  if (B.stride <= 40) decombine_rescue16_one<false, UNIFORM, 10>(T, B, C, r, 3u, nw, CC, records, slot);
  else decombine_rescue16_one<false, UNIFORM, DCRX_NWMAX>(T, B, C, r, 3u, nw, CC, records, slot);
}

// a read with exception bytes: the rescue kernel's general form when the launch would use it
template <bool UNIFORM>
static void general_one(bool pair_rescue, const DevTables &T, const BatchDev &B, const CfgDev &C, uint64_t r, uint32_t nw,
                        const Counters &CC, dcrx_record_t *records, uint32_t *slot) {
  if (!pair_rescue) { decombine_list_one<false, UNIFORM>(T, nullptr, B, C, r, CC, records, slot, true); return; }
  uint32_t e0 = 0;
  while (e0 < B.n_exc && B.exc_read[e0] < (uint32_t)r) e0++;
  if (B.stride <= 40) decombine_general16_one<false, UNIFORM, 10>(T, B, C, r, e0, nw, CC, records, slot);
  else decombine_general16_one<false, UNIFORM, DCRX_NWMAX>(T, B, C, r, e0, nw, CC, records, slot);
}

constexpr int FAST_DIGEST_MISMATCH = -77;      // (emul_decombine then returns -100)
static uint64_t g_v2_lean = 0;    // ... of them settled by the lean tail
extern "C" uint64_t emul_v2_lean(void) { const uint64_t v = g_v2_lean; g_v2_lean = 0; return v; }
namespace dcrx { unsigned long long g_r2_reasons[32]; unsigned long long g_walk_steps[2]; }
static unsigned long long g_walk_hist[2][16];
extern "C" void emul_walk_hist(uint64_t *out) { for (int i = 0; i < 32; i++) { out[i] = g_walk_hist[i / 16][i % 16]; g_walk_hist[i / 16][i % 16] = 0; } }
extern "C" void emul_r2_reasons(uint64_t *out) { for (int i = 0; i < 32; i++) { out[i] = dcrx::g_r2_reasons[i]; dcrx::g_r2_reasons[i] = 0; } }
static uint64_t g_v2_stats[64];  // [what] ; 8 + min(#events, 15) ; 24 + kind of event entry; 32 + status of event entries
extern "C" void emul_v2_stats(uint64_t *out) { for (int i = 0; i < 64; i++) { out[i] = g_v2_stats[i]; g_v2_stats[i] = 0; } }
#ifdef DCRX_R2_PROFILE
// tools/r2_profile.py: (what, a) per loop trip of the lean rescue, an entry's events behind a (0, read) marker; every clean event
// entry then takes the form compiled for its shape, as the kernel's do
static std::vector<int32_t> g_r2_trace;
namespace dcrx { void r2_prof(int what, int a) { g_r2_trace.push_back(what); g_r2_trace.push_back(a); } }
extern "C" uint64_t emul_r2_trace(int32_t *out, uint64_t cap) {
  const uint64_t n = g_r2_trace.size();
  if (out) { for (uint64_t i = 0; i < n && i < cap; i++) out[i] = g_r2_trace[i]; g_r2_trace.clear(); }
  return n;
}
#endif
static uint64_t g_v2_reads = 0;   // reads that took the v2 form since the last emul_v2_reads() call
extern "C" uint64_t emul_v2_reads(void) { const uint64_t v = g_v2_reads; g_v2_reads = 0; return v; }

// What one lane of the v2 kernel does with a clean read: scan, digest, classification, and the
// finishing of tail / event entries (the kernel batches those in stacks; here they run at once).
// Returns FAST_TO_RESCUE for a read with more flagged pairs than an event entry holds.
// [x0, x1): the read's slice of the exception list (x0 == x1: a clean read).  A read with exception bytes the v2 kernels
// hand over returns FAST_TO_GENERAL.
template <bool UNIFORM, int NW>
static int v2_one(const DevTables &T, const BatchDev &B, const CfgDev &C, uint64_t r, int x0, int x1, const Counters &CC,
                  dcrx_record_t *records) {
  const bool exc = x1 > x0;
  const int o = C.orientation == DCRX_ORIENT_FORWARD ? 0 : 1;
  const V2Ori &V = T.v2[o];
  const V2Tab tab{V.trans};
  g_v2_reads++;
  const uint32_t nw = B.stride >> 2;
  uint32_t w[1][NW], lg[1][NW];
  const uint32_t *words = reinterpret_cast<const uint32_t *>(B.packed + r * B.stride);
  for (int k = 0; k < NW; k++) w[0][k] = (uint32_t)k < nw ? words[k] : 0u;
  const int n = UNIFORM ? (int)B.read_len : (int)B.lens[r];
  const int npairs = UNIFORM ? (int)((B.read_len + 1u) >> 1) : 8 * (int)(nw < (uint32_t)NW ? nw : (uint32_t)NW);
  if (V.narrow) scan2<NW, 1, true>(tab, w, lg, npairs); else scan2<NW, 1, false>(tab, w, lg, npairs);
  if (!UNIFORM) mask_log2<NW>(lg[0], n);
  const Digest2 d = digest2_scan<NW>(lg[0]);      // (the scan kernel's digest; the fields anyone reads must be digest2's)
  {
    const Digest2 d0 = digest2<NW>(lg[0]);
    if (d.any != d0.any || d.vf_n != d0.vf_n || d.jf_n != d0.jf_n || (d0.vf_n == 1 && d.vf_pair != d0.vf_pair) ||
        (d0.jf_n == 1 && d.jf_pair != d0.jf_pair))
      return FAST_DIGEST_MISMATCH;
  }
  const uint32_t bnd = (n & 1) ? log_nibble<NW>(lg[0], n >> 1) : 0u;
  int what = classify2(d, bnd);
  if (exc && what != V2_VNONE) what = V2_EVENTS;
  g_v2_stats[what]++;
  if (what == V2_EVENTS && x1 - x0 > V2_MAX_EXC) {      // more exception bytes than the register frame holds beside a run of Ns
    ExcLayout xl;
    if (!exc_layout(B.exc_pos, B.exc_chr, x0, x1, xl)) return FAST_TO_GENERAL;
  }
  if (what == V2_VNONE || what == V2_VMULTI) {
    dcrx_record_t rec;
    std::memset(&rec, 0, sizeof rec);
    rec.status = (uint8_t)(what == V2_VNONE ? DCRX_S_V_NONE : DCRX_S_V_MULTI);
    rec.frame = (uint8_t)(o == 0 ? 1 : 0);
    records[r] = rec;
    CC.add(what == V2_VNONE ? DCRX_C_NO_VTAGS_FOUND : DCRX_C_MULTIPLE_V_MATCHES);
    CC.add(DCRX_C_READ_COUNT);
    return FAST_DONE;
  }
  uint32_t ev[3];
  bool jmulti = false;
  if (what == V2_TAIL) {
    // the lean tail first; what it does not settle takes the general form from the same entry
    const Tail2Tabs tt = tail2_tabs(T, V, T.image + T.dfa_bytes, V.bk, o == 1);
    dcrx_record_t rec;
    std::memset(&rec, 0, sizeof rec);
    // (the words in a strip of memory, as the kernel keeps them in LDS)
    uint32_t strip[NW + 2];
    for (int k = 0; k < NW; k++) strip[k] = w[0][k];
    strip[NW] = strip[NW + 1] = 0u;
    const LdsWords lw{dcrx_ldsaddr_of(strip)};
    const int st = o ? tail2_fast<true>(tt, lw, n, tail2_pack(d), C, rec, T, CC) : tail2_fast<false>(tt, lw, n, tail2_pack(d), C, rec, T, CC);
    if (st != TAIL2_SLOW) {
      rec.status = (uint8_t)st; rec.frame = (uint8_t)(o ? 0 : 1);
      records[r] = rec;
      tail2_count(CC, st, o == 0);
      g_v2_lean++;
      return FAST_DONE;
    }
    tail2_events(tail2_pack(d), ev, jmulti);
    return finish2_reg<UNIFORM, NW>(T, V, B, C, r, ev, jmulti, 0, 0, CC, records) ? FAST_DONE : FAST_TO_RESCUE;
  }
  if (!exc && !(C.flags & DCRX_F_V2_NO_LEAN_RESCUE)) {
    // the lean rescue first (the kernel's order): what it settles is final
    uint32_t kwb[K_NCLASS];
    for (int c = 0; c < K_NCLASS; c++) kwb[c] = T.kw_base[c];
    const Rescue2Tabs rt = rescue2_tabs(T, V, T.image + T.dfa_bytes, V.bk, o == 1, kwb);
    dcrx_record_t rec;
    std::memset(&rec, 0, sizeof rec);
    uint32_t errs = 0;
    const RegWords<NW> rw{w[0]};      // (and from registers here: both word sources are exercised)
    uint32_t dry[DCRX_N_COUNTERS] = {0};
    const Counters Cdry{dry};
    // (the kernel sorts event entries by shape and runs the form compiled for the shape; here every other read takes that
    // form and the rest the form that decides per read: both must agree with the oracle)
    const int shp = shape2(d.vf_n, d.jf_n, d.any);
    int st;
#define DCRX_EMUL_R2(SH) (o ? rescue2_fast<true, NW, SH>(rt, rw, lg[0], n, C, rec, errs, T, CC, Cdry) : rescue2_fast<false, NW, SH>(rt, rw, lg[0], n, C, rec, errs, T, CC, Cdry))
#ifdef DCRX_R2_PROFILE
    dcrx::r2_prof(0, (int)r); dcrx::r2_prof(9, shp);
    if (bnd) st = DCRX_EMUL_R2(V2_SHAPE_ANY);
#else
    if ((r & 1) || bnd) st = DCRX_EMUL_R2(V2_SHAPE_ANY);
#endif
    else if (shp == V2_SHAPE_ONE) st = DCRX_EMUL_R2(V2_SHAPE_ONE);
    else st = DCRX_EMUL_R2(V2_SHAPE_BOTH);
#undef DCRX_EMUL_R2
#ifdef DCRX_R2_PROFILE
    dcrx::r2_prof(7, st);
#endif
    if (st >= 0) {
      rec.status = (uint8_t)st; rec.frame = (uint8_t)(o ? 0 : 1);
      records[r] = rec;
      rescue2_count(CC, st, errs, o == 0);
      g_v2_stats[62]++;
      return FAST_DONE;
    }
  }
  if (!events2<NW>(lg[0], d, exc ? 0xFu : bnd, ev)) return exc ? FAST_TO_GENERAL : FAST_TO_RESCUE;
  {
    const Events2 E{(uint64_t)ev[0] | ((uint64_t)ev[1] << 32), ev[2]};
    g_v2_stats[8 + E.count()]++;
    g_v2_stats[24 + (d.vf_n == 1 ? 0 : 1) + (d.jf_n == 1 ? 0 : (d.jf_n == 0 ? 2 : 4))]++;   // 24 V1 J1(bnd) ; 25 V0 J1 ; 26 V1 J0 ; 27 V0 J0 ; 28/29 J multi
    dcrx::g_walk_steps[0] = dcrx::g_walk_steps[1] = 0;
    const bool ok = finish2_reg<UNIFORM, NW>(T, V, B, C, r, ev, false, x0, x1, CC, records);
    for (int g = 0; g < 2; g++) { unsigned long long st = dcrx::g_walk_steps[g]; int b = 0; while (st) { b++; st >>= 1; } g_walk_hist[g][b]++; }
    if (ok) g_v2_stats[32 + (records[r].status < 30 ? records[r].status : 30)]++; else g_v2_stats[63]++;
    return ok ? FAST_DONE : (exc ? FAST_TO_GENERAL : FAST_TO_RESCUE);
  }
  return finish2_reg<UNIFORM, NW>(T, V, B, C, r, ev, false, x0, x1, CC, records) ? FAST_DONE : (exc ? FAST_TO_GENERAL : FAST_TO_RESCUE);
}

extern "C" int emul_decombine(const dcrx_tagset_t *ts, const dcrx_cfg_t *cfg, const dcrx_batch_t *b,
                              dcrx_record_t *records, uint64_t *counters, char *err, int err_cap) {
  HostTables H;
  std::string e;
  int rc = compile_tables(ts, &H, &e);
  if (rc) { std::strncpy(err, e.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; return rc; }
  const DevTables T = H.resolve(H.blob.data());
  std::vector<uint32_t> flag((b->n_reads + 31) / 32 + 1, 0);
  for (uint64_t i = 0; i < b->n_exc; i++) flag[b->exc_read[i] >> 5] |= 1u << (b->exc_read[i] & 31);
  // reads are copied with a padded tail so that the word-pair loads stay in bounds
  std::vector<uint8_t> packed(b->n_reads * (size_t)b->stride + 4 * DCRX_V2_NWLONG + 16, 0);
  std::memcpy(packed.data(), b->packed, b->n_reads * (size_t)b->stride);
  BatchDev B;
  B.packed = packed.data(); B.stride = b->stride; B.read_len = b->read_len; B.lens = b->lens;
  B.n_reads = b->n_reads; B.n_exc = b->n_exc; B.exc_read = b->exc_read; B.exc_pos = b->exc_pos;
  B.exc_chr = b->exc_chr; B.exc_flag = flag.data();
  CfgDev C{cfg->orientation, cfg->allow_ns, cfg->lenthreshold, cfg->flags};
  uint32_t counts[DCRX_N_COUNTERS] = {0};
  Counters CC{counts};
  for (int c = 0; c < DCRX_N_COUNTERS; c++) counters[c] = 0;
  for (uint64_t r = 0; r < b->n_reads; r++) {
    // what the launch does: reads of the general list (exception bytes; every read for `both` /
    // forced slow reader) take the general kernel's form; the rest go through the fast kernel and,
    // when deferred, the rescue kernel's form.
    uint32_t slot[DCRX_GENERAL_SLOT + 2];
    const uint32_t nw = b->stride / 4;
    if (B.stride > 4 * DCRX_V2_NWLONG) {      // reads of 512 nt and more: the long form (dcrx_kernels.hip, launch_long), every read of the batch
      uint32_t lslot[DCRX_LONG_SLOT_MAX];
      const int sl = (r & 1) ? DCRX_LONG_SLOT_MIN : DCRX_LONG_SLOT_MAX;      // (the launch's choice by the tables' size: both ends of it)
      if (b->lens) decombine_long_one<false, false>(T, nullptr, B, C, r, CC, records, lslot, sl);
      else decombine_long_one<true, false>(T, nullptr, B, C, r, CC, records, lslot, sl);
      for (int c = 0; c < DCRX_N_COUNTERS; c++) { counters[c] += counts[c]; counts[c] = 0; }
      continue;
    }
    const bool pair_scan = T.dfa16_bytes != 0 && !(C.flags & DCRX_F_ONE_BASE_SCAN);
    const bool pair_rescue = pair_scan && T.pair_rescue && !(C.flags & DCRX_F_LIST_RESCUE);
    // the launch's choice of kernels (dcrx_kernels.hip, v2_applies); reads beyond the three-launch form's 320 nt stay on the
    // v2 functions (32 words per read) where those apply, and all take the general list form otherwise
    const bool v2_able = T.v2_ok && !(C.flags & (DCRX_F_V1_KERNELS | DCRX_F_ONE_BASE_SCAN | DCRX_F_LIST_RESCUE | DCRX_F_PROFILE_LIST_SCAN_ONLY |
                                                  DCRX_F_PROFILE_RESCUE_HITS_ONLY | DCRX_F_FORCE_SLOW_READER));
    const bool all_general = C.orientation == DCRX_ORIENT_BOTH || (C.flags & DCRX_F_FORCE_SLOW_READER) || (B.stride > 4 * DCRX_NWMAX && !v2_able);
    const bool general = all_general || ((flag[r >> 5] >> (r & 31)) & 1u);
    const bool v2 = v2_able && !all_general;
    // the v2 kernels take every read of the batch, those with exception bytes included (with their slice of the list)
    int x0 = 0, x1 = 0;
    if (v2 && general) {
      while ((uint64_t)x0 < B.n_exc && B.exc_read[x0] < (uint32_t)r) x0++;
      x1 = x0;
      while ((uint64_t)x1 < B.n_exc && B.exc_read[x1] == (uint32_t)r) x1++;
    }
    if (b->lens) {
      int what = FAST_TO_GENERAL;
      if (v2) what = B.stride <= 40 ? v2_one<false, 10>(T, B, C, r, x0, x1, CC, records) : B.stride <= 4 * DCRX_NWMAX ? v2_one<false, DCRX_NWMAX>(T, B, C, r, x0, x1, CC, records) : v2_one<false, DCRX_V2_NWLONG>(T, B, C, r, x0, x1, CC, records);
      else if (!general) what = fast_one<false>(pair_scan, T, B, C, r, nw, CC, records);
      if (what == FAST_TO_GENERAL) general_one<false>(pair_rescue && !all_general, T, B, C, r, nw, CC, records, slot);
      else if (what == FAST_TO_RESCUE) rescue_one<false>(pair_rescue, T, B, C, r, nw, CC, records, slot);
      else if (what != FAST_DONE) return -100;
    } else {
      int what = FAST_TO_GENERAL;
      if (v2) what = B.stride <= 40 ? v2_one<true, 10>(T, B, C, r, x0, x1, CC, records) : B.stride <= 4 * DCRX_NWMAX ? v2_one<true, DCRX_NWMAX>(T, B, C, r, x0, x1, CC, records) : v2_one<true, DCRX_V2_NWLONG>(T, B, C, r, x0, x1, CC, records);
      else if (!general) what = fast_one<true>(pair_scan, T, B, C, r, nw, CC, records);
      if (what == FAST_TO_GENERAL) general_one<true>(pair_rescue && !all_general, T, B, C, r, nw, CC, records, slot);
      else if (what == FAST_TO_RESCUE) rescue_one<true>(pair_rescue, T, B, C, r, nw, CC, records, slot);
      else if (what != FAST_DONE) return -100;
    }
    for (int c = 0; c < DCRX_N_COUNTERS; c++) { counters[c] += counts[c]; counts[c] = 0; }
  }
  return 0;
}
