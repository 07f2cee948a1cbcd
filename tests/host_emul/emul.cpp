// emul.cpp — TEST-ONLY host build of the per-read device code
// (decombinator_amd/csrc/dcrx_dcr_device.h) so that the exact functions the HIP
// kernel runs can be checked against the oracle, and run under ASan/UBSan,
// without a GPU.  It is never linked into libdcrx.so and nothing under
// decombinator_amd/ refers to it.  It does NOT replace the GPU parity tests:
// the kernel's launch geometry, LDS staging, counter reduction and the real
// ISA are only exercised by `pytest -m gpu`.
#define DCRX_HOST_EMUL 1
#include <cstring>
#include <string>
#include <vector>

#include "../../decombinator_amd/csrc/dcrx_dcr_device.h"
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
  if (!pair_rescue) { decombine_list_one<false, UNIFORM>(T, nullptr, B, C, r, CC, records, slot); return; }
  if (B.stride <= 40) decombine_rescue16_one<false, UNIFORM, 10>(T, B, C, r, 3u, nw, CC, records, slot);
  else decombine_rescue16_one<false, UNIFORM, DCRX_NWMAX>(T, B, C, r, 3u, nw, CC, records, slot);
}

// a read with exception bytes: the rescue kernel's general form when the launch would use it
template <bool UNIFORM>
static void general_one(bool pair_rescue, const DevTables &T, const BatchDev &B, const CfgDev &C, uint64_t r, uint32_t nw,
                        const Counters &CC, dcrx_record_t *records, uint32_t *slot) {
  if (!pair_rescue) { decombine_list_one<false, UNIFORM>(T, nullptr, B, C, r, CC, records, slot); return; }
  uint32_t e0 = 0;
  while (e0 < B.n_exc && B.exc_read[e0] < (uint32_t)r) e0++;
  if (B.stride <= 40) decombine_general16_one<false, UNIFORM, 10>(T, B, C, r, e0, nw, CC, records, slot);
  else decombine_general16_one<false, UNIFORM, DCRX_NWMAX>(T, B, C, r, e0, nw, CC, records, slot);
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
  std::vector<uint8_t> packed(b->n_reads * (size_t)b->stride + 4 * DCRX_NWMAX + 16, 0);
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
    const bool pair_scan = T.dfa16_bytes != 0 && !(C.flags & DCRX_F_ONE_BASE_SCAN);
    const bool pair_rescue = pair_scan && T.pair_rescue && !(C.flags & DCRX_F_LIST_RESCUE);
    const bool all_general = C.orientation == DCRX_ORIENT_BOTH || (C.flags & DCRX_F_FORCE_SLOW_READER);
    const bool general = all_general || ((flag[r >> 5] >> (r & 31)) & 1u);
    if (b->lens) {
      if (general) general_one<false>(pair_rescue && !all_general, T, B, C, r, nw, CC, records, slot);
      else {
        const int what = fast_one<false>(pair_scan, T, B, C, r, nw, CC, records);
        if (what == FAST_TO_RESCUE) rescue_one<false>(pair_rescue, T, B, C, r, nw, CC, records, slot);
        else if (what != FAST_DONE) return -100;
      }
    } else {
      if (general) general_one<true>(pair_rescue && !all_general, T, B, C, r, nw, CC, records, slot);
      else {
        const int what = fast_one<true>(pair_scan, T, B, C, r, nw, CC, records);
        if (what == FAST_TO_RESCUE) rescue_one<true>(pair_rescue, T, B, C, r, nw, CC, records, slot);
        else if (what != FAST_DONE) return -100;
      }
    }
    for (int c = 0; c < DCRX_N_COUNTERS; c++) { counters[c] += counts[c]; counts[c] = 0; }
  }
  return 0;
}
