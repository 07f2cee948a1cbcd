// dcrx_collapse.cpp — the front half of the reference's `collapse` stage over a batch of `.n12` rows (host, threaded):
// what read_in_data (src/decombinator/collapse.py:482-565) does to every row before it starts grouping rows —
//   get_barcode_positions :367-479 (spacer searches :192-236, filters :240-253, the pass counters :256-275),
//   set_barcode :278-326, check_umi_quality :343-353, the inter-tag length filter :553-556 —
// on the text libdcrx assembled (dcrx_assemble_rows: fields separated by `field_sep`, one row per line).
//
// The spacer searches are the reference's three steps (:204-212), all decided here: the spacer verbatim; else with up to two
// substitutions (regex "(spacer){1s<=2}": the leftmost windows within Hamming distance 2, non-overlapping, left to right);
// else the indel form "(spacer){2i+2d+1s<=2}" — what it adds to the second is the spacer with one base inserted (a window
// of m + 1 bases) or one base deleted (m - 1 bases): a substitution beside an indel would cost 3.  regex.findall takes the
// leftmost start at which either holds, non-overlapping, left to right; where both hold at one start it returns the
// insertion (its backtracking tries the deepest error position first and, there, insertion before deletion — and the
// alignments of one kind that a start admits all reach down to the first mismatch).  Pinned against the regex module
// itself: tests/test_collapse_front.py (differential, mutated barcode regions over the five oligos) and
// tools/fuzz_spacer_search.py (10^6 and more per oligo).  DCRX_CF_DEFER is left for rows that are not 10 fields of ASCII.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <cstring>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

#include "../../include/dcrx.h"

namespace dcrx { int set_err(int code, const char *msg); }
using dcrx::set_err;

namespace {

struct Oligo { const char *s1; const char *s2; };
const Oligo kOligos[5] = {
    {"GTCGTGACTGGGAAAACCCTGG", "GTCGTGAT"},   // m13      (collapse.py:174-189)
    {"GTCGTGAT", "GTCGTGAT"},                 // i8
    {"ATCACGAC", nullptr},                    // i8_single
    {"TACGGG", nullptr},                      // nebio
    {"GTACGGG", nullptr},                     // takara
};

// str.find(sub, from) on [s, s + n)
inline int find_from(const char *s, int n, const char *sub, int m, int from) {
  if (from < 0) from = 0;
  if (m == 0) return from <= n ? from : -1;
  if (from > n - m) return -1;
  const void *p = memmem(s + from, (size_t)(n - from), sub, (size_t)m);
  return p ? (int)((const char *)p - s) : -1;
}

// spacerSearch(spacer, s[0..n)) :204-212.  Returns the number of non-overlapping matches (left to right); the first `cap` of
// them go to starts[] / lens[] (length m; m + 1 / m - 1 for an indel match); kind: 0 verbatim, 1 substitutions, 2 indel.
int spacer_search_all(const char *s, int n, const char *sp, int m, int *starts, int *lens, int cap, int &kind) {
  int cnt = 0;
  auto put = [&](int at, int len) { if (cnt < cap) { starts[cnt] = at; lens[cnt] = len; } cnt++; };
  kind = 0;
  for (int at = find_from(s, n, sp, m, 0); at >= 0; at = find_from(s, n, sp, m, at + m)) put(at, m);     // regex.findall(spacer, s)
  if (cnt) return cnt;
  kind = 1;
  for (int i = 0; i + m <= n;) {                                                               // {1s<=2}
    int d = 0;
    for (int k = 0; k < m && d <= 2; k++) d += s[i + k] != sp[k];
    if (d <= 2) { put(i, m); i += m; } else i++;
  }
  if (cnt) return cnt;
  // {2i+2d+1s<=2} (:198-201), given that neither search above found anything
  kind = 2;
  for (int i = 0; i < n;) {
    int a = 0;
    while (a < m && i + a < n && s[i + a] == sp[a]) a++;                 // the spacer's prefix at i
    int span = 0;
    if (i + m + 1 <= n) {                                                // one base inserted: prefix + one base + the rest
      int b = 0;
      while (b < m && s[i + m - b] == sp[m - 1 - b]) b++;
      if (a + b >= m) span = m + 1;
    }
    if (!span && m >= 2 && i + m - 1 <= n) {                             // one base deleted
      int b = 0;
      while (b < m - 1 && s[i + m - 2 - b] == sp[m - 1 - b]) b++;
      if (a + b >= m - 1) span = m - 1;
    }
    if (span) { put(i, span); i += span; } else i++;
  }
  return cnt;
}
// ... as get_barcode_positions uses it: the count, the first match, whether the matches are verbatim
int spacer_search(const char *s, int n, const char *sp, int m, int &first_at, int &first_len, bool &exact) {
  int at = -1, len = m, kind = 0;
  const int cnt = spacer_search_all(s, n, sp, m, &at, &len, 1, kind);
  first_at = cnt ? at : -1; first_len = cnt ? len : m; exact = kind == 0;
  return cnt;
}

struct Cfg { int oligo, allow_ns, lenthreshold; double min_q, below_min, avg_q; const char *sep; int sep_len; };

// one row [row, row + len) (no newline).  Returns the status; counters only for a decided row.
uint8_t front_row(const char *row, int len, const Cfg &c, dcrx_collapse_row_t &o, uint64_t *cnt) {
  std::memset(&o, 0, sizeof o);
  o.b1start = o.b1end = o.b2start = o.b2end = -1;
  // fields: 0-4 dcr, 5 id, 6 inter-tag seq, 7 its quality, 8 barcode region, 9 its quality (, 10 v_tail)
  int fstart[12], nf = 0;
  fstart[nf++] = 0;
  for (int i = 0; i + c.sep_len <= len && nf < 11;) {          // (hops from one occurrence of the separator's first byte to the next)
    const void *hit = memchr(row + i, c.sep[0], (size_t)(len - c.sep_len + 1 - i));
    if (!hit) break;
    i = (int)((const char *)hit - row);
    if (std::memcmp(row + i, c.sep, (size_t)c.sep_len) == 0) { i += c.sep_len; fstart[nf++] = i; } else i++;
  }
  if (nf < 10) return DCRX_CF_DEFER;                    // not a 10-field row: the caller's own error path
  auto fend = [&](int f) { return f + 1 < nf ? fstart[f + 1] - c.sep_len : len; };
  const char *bc = row + fstart[8]; const int nbc = fend(8) - fstart[8];
  const char *bq = row + fstart[9]; const int nbq = fend(9) - fstart[9];
  const int seq_len = fend(6) - fstart[6];
  {      // non-ASCII text: Python counts characters, not bytes
    uint64_t hi = 0;
    int i = 0;
    for (; i + 8 <= len; i += 8) { uint64_t w; std::memcpy(&w, row + i, 8); hi |= w; }
    for (; i < len; i++) hi |= (uint64_t)(unsigned char)row[i];
    if (hi & 0x8080808080808080ull) return DCRX_CF_DEFER;
  }
  uint64_t add[DCRX_CF_N_COUNTERS] = {0};
  auto done = [&](uint8_t st) { for (int k = 0; k < DCRX_CF_N_COUNTERS; k++) cnt[k] += add[k]; return st; };
  add[DCRX_CF_C_INPUT_DCRS]++;
  // ---- get_barcode_positions :367-479 ----
  const Oligo &ol = kOligos[c.oligo];
  const int m1 = (int)std::strlen(ol.s1), m2 = ol.s2 ? (int)std::strlen(ol.s2) : 0;
  bool have = true;
  if (!c.allow_ns && memchr(bc, 'N', (size_t)nbc)) { add[DCRX_CF_C_FAIL_N]++; have = false; }                     // :390-394
  int p0 = -1, p1 = -1, l0 = m1, l1 = m2;
  bool ex0 = true, ex1 = true;
  if (have) {
    const int ws = c.oligo == 3 ? 18 : 0, we = c.oligo == 3 ? 28 : (c.oligo == 4 ? 19 : 10 + m1);               // :398-409
    const int a = ws < nbc ? ws : nbc, b = we < nbc ? we : nbc;
    int at = -1;
    const int n1 = spacer_search(bc + a, b - a, ol.s1, m1, at, l0, ex0);
    if (n1 != 1) { add[DCRX_CF_C_FAIL_NOSPACER]++; have = false; }                                             // :413-416
    else p0 = a + at;
  }
  if (have && ol.s2) {
    const int a = m1 < nbc ? m1 : nbc;
    int at = -1;
    const int n2 = spacer_search(bc + a, nbc - a, ol.s2, m2, at, l1, ex1);
    if (n2 != 1) { add[DCRX_CF_C_FAIL_NOT2SPACERS]++; have = false; }                                          // :423-426
    else p1 = a + at;
  }
  int b1s = 0, b1e = 0, b2s = -1, b2e = -1;
  if (have) {
    // getSpacerPositions :230-237: bcseq.find(found string, startpos), startpos advancing by the found strings' lengths
    const int q0 = find_from(bc, nbc, bc + p0, l0, 0);
    int q1 = -1;
    if (ol.s2) q1 = find_from(bc, nbc, bc + p1, l1, l0);
    const bool all_exact = ex0 && (!ol.s2 || ex1);
    if (c.oligo >= 3) {                                                                                        // nebio / takara :431-437
      const int bclength = c.oligo == 3 ? 17 : 12;
      b1s = 0; b1e = bclength;
      add[all_exact ? DCRX_CF_C_PASS_EXACT : DCRX_CF_C_PASS_REGEX]++;
      const int b1len = bclength;
      if (!all_exact) add[DCRX_CF_C_PASS_FUZZY_RIGHTLEN]++; else add[DCRX_CF_C_PASS_OTHER]++;                  // :263-275 with b1len == bclength
      (void)b1len; (void)q0;
    } else {
      const int bclength = 6;
      if (c.oligo == 2) { b1s = 0; b1e = q0; b2s = q0 + l0; }                                                   // i8_single :440-443
      else { b1s = q0 + l0; b1e = q1; b2s = q1 + l1; }                                                          // :445-447
      b2e = b2s + bclength;
      const int b1len = b1e - b1s;
      if (b1len <= 3) { add[DCRX_CF_C_FAIL_N1SHORT]++; have = false; }                                         // :240-253
      else if (b1len >= 9) { add[DCRX_CF_C_FAIL_N1LONG]++; have = false; }
      else if (b2e > nbc) { add[DCRX_CF_C_FAIL_N2PASTEND]++; have = false; }
      else {
        add[all_exact ? DCRX_CF_C_PASS_EXACT : DCRX_CF_C_PASS_REGEX]++;                                         // :256-260
        if (b1len == bclength && !all_exact) add[DCRX_CF_C_PASS_FUZZY_RIGHTLEN]++;                             // :263-275
        else if ((b1len == 4 || b1len == 5) && !all_exact) add[DCRX_CF_C_PASS_FUZZY_SHORT]++;
        else if (b1len >= 7 && !all_exact) add[DCRX_CF_C_PASS_FUZZY_LONG]++;
        else if (b1len == bclength) add[DCRX_CF_C_PASS_OTHER]++;
      }
    }
  }
  if (!have) { add[DCRX_CF_C_FAIL_NO_BCLOCS]++; return done(DCRX_CF_NO_BCLOCS); }                                // :545-547
  o.b1start = (int16_t)b1s; o.b1end = (int16_t)b1e; o.b2start = (int16_t)b2s; o.b2end = (int16_t)b2e;
  // ---- set_barcode :278-326 (Python slices clamp to the string) ----
  auto cut = [](const char *s, int n, int a, int b, char *dst) { if (a < 0) a = 0; if (b > n) b = n; int k = 0; for (int i = a; i < b; i++) dst[k++] = s[i]; return k; };
  int nb = 0, nq = 0;
  if (c.oligo >= 3) {
    nb = cut(bc, nbc, b1s, b1e, o.barcode); nq = cut(bq, nbq, b1s, b1e, o.barcode_qual);
  } else {
    const int n1 = b1e - b1s;
    if (n1 == 6) {
      nb = cut(bc, nbc, b1s, b1e, o.barcode); nb += cut(bc, nbc, b2s, b2e, o.barcode + nb);
      nq = cut(bq, nbq, b1s, b1e, o.barcode_qual); nq += cut(bq, nbq, b2s, b2e, o.barcode_qual + nq);
    } else if (n1 < 6) {
      nb = cut(bc, nbc, b1s, b1e, o.barcode); for (int k = 0; k < 6 - n1; k++) o.barcode[nb++] = 'S'; nb += cut(bc, nbc, b2s, b2e, o.barcode + nb);
      nq = cut(bq, nbq, b1s, b1e, o.barcode_qual); for (int k = 0; k < 6 - n1; k++) o.barcode_qual[nq++] = '?'; nq += cut(bq, nbq, b2s, b2e, o.barcode_qual + nq);
      add[DCRX_CF_C_SHORT_BARCODE]++;
    } else {                                           // "?" * (6 - n1) is the empty string for n1 > 6 (:318): reproduced
      nb = cut(bc, nbc, b1s, b1s + 5, o.barcode); o.barcode[nb++] = 'L'; nb += cut(bc, nbc, b2s, b2e, o.barcode + nb);
      nq = cut(bq, nbq, b1s, b1s + 5, o.barcode_qual); nq += cut(bq, nbq, b2s, b2e, o.barcode_qual + nq);
      add[DCRX_CF_C_LONG_BARCODE]++;
    }
  }
  o.barcode_len = (uint8_t)nb; o.barcode_qual_len = (uint8_t)nq;
  if (nq == 0) return DCRX_CF_DEFER;                   // (the reference divides by the length: its own error)
  // ---- check_umi_quality :343-353 ----
  long sum = 0; int below = 0;
  for (int k = 0; k < nq; k++) { const int q = (int)(unsigned char)o.barcode_qual[k] - 33; sum += q; below += q < c.min_q; }
  if (below > c.below_min || (double)sum / (double)nq < c.avg_q) { add[DCRX_CF_C_FAIL_LOW_QUALITY]++; return done(DCRX_CF_LOW_QUALITY); }
  if (seq_len > c.lenthreshold) { add[DCRX_CF_C_FAIL_OVERLONG]++; return done(DCRX_CF_OVERLONG); }               // :553-556
  add[DCRX_CF_C_SUCCESS]++;
  return done(DCRX_CF_OK);
}

}  // namespace

static int64_t collapse_front(const char *text, uint64_t n_bytes, const dcrx_collapse_cfg_t *cfg, dcrx_collapse_row_t *rows,
                              uint64_t rows_cap, uint64_t *row_offsets, uint64_t *counters, int n_threads);

extern "C" int32_t dcrx_spacer_search(const char *seq, int32_t n, const char *spacer, int32_t m, int32_t *starts, int32_t *lens, int32_t cap,
                                      int32_t *kind) {
  if (!seq || !spacer || n < 0 || m < 1 || cap < 0 || (cap && (!starts || !lens))) return set_err(DCRX_E_INVALID, "dcrx_spacer_search: bad argument");
  int k = 0;
  const int cnt = spacer_search_all(seq, n, spacer, m, starts, lens, cap, k);
  if (kind) *kind = k;
  return cnt;
}

// (nothing throws across the C ABI: include/dcrx.h; allocation failures of the row index and of the per-thread tallies
// come back as DCRX_E_NOMEM, a worker thread that cannot be started leaves its share to the calling thread)
extern "C" int64_t dcrx_collapse_front(const char *text, uint64_t n_bytes, const dcrx_collapse_cfg_t *cfg, dcrx_collapse_row_t *rows,
                                       uint64_t rows_cap, uint64_t *row_offsets, uint64_t *counters, int n_threads) {
  try { return collapse_front(text, n_bytes, cfg, rows, rows_cap, row_offsets, counters, n_threads); }
  catch (const std::bad_alloc &) { return set_err(DCRX_E_NOMEM, "out of memory in dcrx_collapse_front"); }
  catch (...) { return set_err(DCRX_E_NOMEM, "dcrx_collapse_front: unexpected exception"); }
}

static int64_t collapse_front(const char *text, uint64_t n_bytes, const dcrx_collapse_cfg_t *cfg, dcrx_collapse_row_t *rows,
                              uint64_t rows_cap, uint64_t *row_offsets, uint64_t *counters, int n_threads) {
  if (!cfg || !counters || (n_bytes && !text)) return set_err(DCRX_E_INVALID, "null argument");
  if (cfg->oligo < 0 || cfg->oligo > 4) return set_err(DCRX_E_INVALID, "oligo must be 0 (m13), 1 (i8), 2 (i8_single), 3 (nebio) or 4 (takara)");
  if (!cfg->field_sep[0]) return set_err(DCRX_E_INVALID, "field_sep is empty");
  Cfg c{cfg->oligo, cfg->allow_ns, cfg->lenthreshold, cfg->min_bc_q, cfg->bc_q_below_min, cfg->avg_q_threshold, cfg->field_sep, (int)strnlen(cfg->field_sep, sizeof cfg->field_sep)};
  // rows = lines (every row ends with a newline; text after the last newline is one more row)
  std::vector<uint64_t> starts;
  for (uint64_t at = 0; at < n_bytes;) {
    starts.push_back(at);
    const void *nl = memchr(text + at, '\n', (size_t)(n_bytes - at));
    at = nl ? (uint64_t)((const char *)nl - text) + 1 : n_bytes;
  }
  const uint64_t n = starts.size();
  if (!rows) return (int64_t)n;                           // sizing call
  if (n > rows_cap) return set_err(DCRX_E_INVALID, "rows_cap is smaller than the number of rows");
  starts.push_back(n_bytes);
  if (row_offsets) for (uint64_t r = 0; r <= n; r++) row_offsets[r] = starts[r];
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (const char *e = std::getenv("DCRX_HOST_THREADS")) nt = (unsigned)std::atoi(e);
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  if (n < 4096) nt = 1;
  std::vector<std::vector<uint64_t>> part(nt, std::vector<uint64_t>(DCRX_CF_N_COUNTERS, 0));
  auto work = [&](unsigned t) {
    const uint64_t lo = n * t / nt, hi = n * (t + 1) / nt;
    for (uint64_t r = lo; r < hi; r++) {
      uint64_t e = starts[r + 1];
      if (e > starts[r] && text[e - 1] == '\n') e--;
      if (e - starts[r] > 0x7FFFFFFFull) { rows[r].status = DCRX_CF_DEFER; continue; }
      rows[r].status = front_row(text + starts[r], (int)(e - starts[r]), c, rows[r], part[t].data());
    }
  };
  std::thread th[64];
  bool own[64] = {false};
  for (unsigned t = 1; t < nt; t++) {
    try { th[t] = std::thread(work, t); }
    catch (...) { own[t] = true; }          // this share stays with the calling thread
  }
  work(0);
  for (unsigned t = 1; t < nt; t++) if (own[t]) work(t);
  for (unsigned t = 1; t < nt; t++) if (th[t].joinable()) th[t].join();
  for (int k = 0; k < DCRX_CF_N_COUNTERS; k++) { uint64_t s = 0; for (unsigned t = 0; t < nt; t++) s += part[t][k]; counters[k] += s; }
  return (int64_t)n;
}
