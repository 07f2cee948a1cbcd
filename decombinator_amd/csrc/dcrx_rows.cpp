// Bulk `.n12` row assembly (host only): what the reference's loop body builds for every
// decombined read (src/decombinator/decombine.py:1012-1039) —
//   [v, j, vdel, jdel, insert, read id, inter-tag sequence, inter-tag quality, barcode,
//    barcode quality (, v_tail with sampling_analysis)]
// from the 16-byte records and the byte spans of the FASTQ batch.  Strings are cut with
// Python's slice clamping; the reverse frame is revcomp(vdj) / vdjqual[::-1] (:1015-1017)
// with Bio.Seq's complement table (ambiguous codes, both cases, other bytes unchanged).
// Output: one line per decombined read, in read order, fields separated by `field_sep`.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   // memmem
#endif
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/dcrx.h"
#include "../../include/dcrx_codes.h"

namespace dcrx { int set_err(int code, const char *msg); }
using dcrx::set_err;

namespace {
struct Comp {
  uint8_t t[256];
  Comp() {
    for (int c = 0; c < 256; c++) t[c] = (uint8_t)c;
    const char *ck = "ACGTMRWSYKVHDBXNU", *cv = "TGCAKYWSRMBDHVXNA";
    for (int i = 0; ck[i]; i++) {
      t[(uint8_t)ck[i]] = (uint8_t)cv[i];
      t[(uint8_t)(ck[i] + 32)] = (uint8_t)(cv[i] + 32);
    }
  }
};
const Comp g_comp;

inline uint32_t ndigits(unsigned v) { return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : 5; }
inline uint32_t clamp_len(uint32_t len, uint32_t a, uint32_t b) { if (b > len) b = len; return a < b ? b - a : 0; }

struct Out {
  char *p; uint64_t n;
  const char *sep; size_t sep_len; bool clash;      // clash: a field holds the separator itself
  bool holds_sep(const char *s, uint64_t k) const { return k >= sep_len && memmem(s, k, sep, sep_len) != nullptr; }
  void put(const char *s, uint64_t k) {
    if (holds_sep(s, k)) clash = true;
    std::memcpy(p + n, s, k);
    n += k;
  }
  void ch(char c) { p[n++] = c; }
  void fs() { std::memcpy(p + n, sep, sep_len); n += sep_len; }
  void num(unsigned v) {     // v < 65536
    char b[8]; int k = 0;
    do { b[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) p[n++] = b[--k];
  }
  // frame string [a, b) of a span of length len: forward = the bytes, reverse = reversed (and complemented)
  void cut(const char *s, uint32_t len, bool rev, bool comp, uint32_t a, uint32_t b) {
    if (b > len) b = len;
    if (a >= b) return;
    const uint64_t k = b - a;
    char *o = p + n;
    if (!rev) std::memcpy(o, s + a, k);
    else if (comp) for (uint32_t i = a; i < b; i++) *o++ = (char)g_comp.t[(uint8_t)s[len - 1 - i]];
    else for (uint32_t i = a; i < b; i++) *o++ = s[len - 1 - i];
    if (holds_sep(p + n, k)) clash = true;
    n += k;
  }
};
}  // namespace

static int64_t assemble_rows(const dcrx_record_t *records, uint64_t n_reads, const dcrx_spans_t *vdj,
                                      const dcrx_spans_t *qual, const dcrx_spans_t *id, const dcrx_spans_t *bc,
                                      const dcrx_spans_t *bcq, const dcrx_spans_t *tail, const char *field_sep, char *out,
                                      uint64_t out_cap, uint64_t *n_rows) {
  if ((n_reads && !records) || !vdj || !qual || !id || !bc || !bcq || !field_sep || !field_sep[0])
    return set_err(DCRX_E_INVALID, "null argument to dcrx_assemble_rows");
  const size_t sep_len = std::strlen(field_sep);
  // Reads are split into contiguous ranges over a few threads: a sizing pass (arithmetic only)
  // gives every range its place in the output, then the ranges are written independently.
  unsigned nt = 1;
  if (n_reads >= (1u << 16)) {
    nt = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("DCRX_HOST_THREADS")) nt = (unsigned)std::atoi(e);
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
  }
  std::vector<uint64_t> r_need(nt, 0), r_rows(nt, 0);
  std::vector<int> r_bad(nt, 0), r_clash(nt, 0);
  auto range = [&](unsigned k, uint64_t &lo, uint64_t &hi) { lo = n_reads * k / nt; hi = n_reads * (k + 1) / nt; };
  auto size_pass = [&](unsigned k) {
    uint64_t lo, hi; range(k, lo, hi);
    uint64_t rows = 0, need = 0;
    for (uint64_t r = lo; r < hi; r++) {
      const dcrx_record_t &c = records[r];
      if (c.status != DCRX_S_OK) continue;
      if (qual->len[r] == DCRX_FASTQ_NO_QUAL) { r_bad[k] = 1; return; }
      need += ndigits(c.v) + ndigits(c.j) + ndigits(c.vdel) + ndigits(c.jdel) + 9 * sep_len + 1
            + clamp_len(vdj->len[r], c.ins_start, (uint32_t)c.ins_start + c.ins_len) + id->len[r]
            + clamp_len(vdj->len[r], c.v_start, c.j_end) + clamp_len(qual->len[r], c.v_start, c.j_end)
            + bc->len[r] + bcq->len[r] + (tail ? sep_len + tail->len[r] : 0);
      rows++;
    }
    r_need[k] = need; r_rows[k] = rows;
  };
  auto run = [&](auto &&fn) {
    if (nt == 1) { fn(0u); return; }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++) th.emplace_back(fn, k);
    for (auto &x : th) x.join();
  };
  run(size_pass);
  uint64_t rows = 0, need = 0;
  for (unsigned k = 0; k < nt; k++) {
    if (r_bad[k]) return set_err(DCRX_E_INVALID, "decombined read without a quality string");
    rows += r_rows[k]; need += r_need[k];
  }
  if (n_rows) *n_rows = rows;
  if (!out || need > out_cap) return (int64_t)need;
  std::vector<uint64_t> r_off(nt, 0);
  for (unsigned k = 1; k < nt; k++) r_off[k] = r_off[k - 1] + r_need[k - 1];
  auto write_pass = [&](unsigned k) {
    uint64_t lo, hi; range(k, lo, hi);
    Out o{out + r_off[k], 0, field_sep, sep_len, false};
    for (uint64_t r = lo; r < hi; r++) {
      const dcrx_record_t &c = records[r];
      if (c.status != DCRX_S_OK) continue;
      const bool rev = c.frame == 0;
      const char *s = vdj->text + vdj->start[r];
      const char *q = qual->text + qual->start[r];
      o.num(c.v); o.fs(); o.num(c.j); o.fs(); o.num(c.vdel); o.fs(); o.num(c.jdel); o.fs();
      o.cut(s, vdj->len[r], rev, true, c.ins_start, (uint32_t)c.ins_start + c.ins_len); o.fs();
      o.put(id->text + id->start[r], id->len[r]); o.fs();
      o.cut(s, vdj->len[r], rev, true, c.v_start, c.j_end); o.fs();
      o.cut(q, qual->len[r], rev, false, c.v_start, c.j_end); o.fs();
      o.put(bc->text + bc->start[r], bc->len[r]); o.fs();
      o.put(bcq->text + bcq->start[r], bcq->len[r]);
      if (tail) { o.fs(); o.put(tail->text + tail->start[r], tail->len[r]); }
      o.ch('\n');
    }
    r_clash[k] = o.clash ? 1 : 0;
  };
  run(write_pass);
  bool clash = false;
  for (unsigned k = 0; k < nt; k++) clash = clash || r_clash[k];
  if (clash) return set_err(DCRX_E_UNSUPPORTED, "a field contains the field separator");
  return (int64_t)need;
}

extern "C" int64_t dcrx_assemble_rows(const dcrx_record_t *records, uint64_t n_reads, const dcrx_spans_t *vdj,
                                      const dcrx_spans_t *qual, const dcrx_spans_t *id, const dcrx_spans_t *bc,
                                      const dcrx_spans_t *bcq, const dcrx_spans_t *tail, const char *field_sep, char *out,
                                      uint64_t out_cap, uint64_t *n_rows) {
  try {
    return assemble_rows(records, n_reads, vdj, qual, id, bc, bcq, tail, field_sep, out, out_cap, n_rows);
  } catch (...) { return set_err(DCRX_E_NOMEM, "out of memory in dcrx_assemble_rows"); }
}

// ------------------------------------------------------------------------------
// The intermediate files' gzip step (reference io.py:497-506: the text file is re-read and written through gzip.open,
// one thread at level 9 — minutes for a run's `.n12`): the text goes out as a multi-member gzip file, its pieces deflated
// by several threads.  Any gzip reader (zcat, Python's gzip, the reference's own opener) reads the members as one stream;
// the decompressed bytes are the reference's, the compressed ones are not meant to be.
// ------------------------------------------------------------------------------
#include <zlib.h>
#include <atomic>
#include <cstdio>

namespace {
struct GzWriter {
  FILE *f = nullptr;
  int level = 6;
  unsigned threads = 1;
};
constexpr size_t GZ_PIECE = 4u << 20;

// one gzip member for data[0, n)
bool gz_member(const uint8_t *data, size_t n, int level, std::vector<uint8_t> &out) {
  z_stream z;
  std::memset(&z, 0, sizeof z);
  if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  out.resize(deflateBound(&z, (uLong)n) + 64);
  z.next_in = const_cast<Bytef *>(data); z.avail_in = (uInt)n;
  z.next_out = out.data(); z.avail_out = (uInt)out.size();
  const int rc = deflate(&z, Z_FINISH);
  const size_t got = out.size() - z.avail_out;
  deflateEnd(&z);
  if (rc != Z_STREAM_END) return false;
  out.resize(got);
  return true;
}
}  // namespace

extern "C" int dcrx_gzip_open(const char *path, int level, int n_threads, void **writer) {
  if (!path || !writer) return set_err(DCRX_E_INVALID, "null argument to dcrx_gzip_open");
  *writer = nullptr;
  if (level < 1 || level > 9) return set_err(DCRX_E_INVALID, "gzip level must be 1..9");
  FILE *f = std::fopen(path, "wb");
  if (!f) return set_err(DCRX_E_INVALID, "cannot open the gzip output file");
  GzWriter *w = new GzWriter;
  w->f = f; w->level = level;
  unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::thread::hardware_concurrency();
  if (const char *e = std::getenv("DCRX_HOST_THREADS")) { if (n_threads <= 0) nt = (unsigned)std::atoi(e); }
  w->threads = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
  *writer = w;
  return DCRX_OK;
}

extern "C" int dcrx_gzip_write(void *writer, const void *data, uint64_t n_bytes) {
  GzWriter *w = static_cast<GzWriter *>(writer);
  if (!w || !w->f || (n_bytes && !data)) return set_err(DCRX_E_INVALID, "null argument to dcrx_gzip_write");
  const uint8_t *p = static_cast<const uint8_t *>(data);
  const size_t pieces = (size_t)((n_bytes + GZ_PIECE - 1) / GZ_PIECE);
  try {
    // rounds of `threads` x 4 pieces: deflated side by side, written in order (memory: a round's output, not the file's)
    const size_t round = (size_t)w->threads * 4;
    std::vector<std::vector<uint8_t>> out(std::min(round, std::max<size_t>(pieces, 1)));
    for (size_t p0 = 0; p0 < pieces; p0 += round) {
      const size_t cnt = std::min(round, pieces - p0);
      std::atomic<size_t> next{0};
      std::atomic<int> bad{0};
      auto work = [&]() {
        for (size_t k; (k = next.fetch_add(1)) < cnt;) {
          const size_t off = (p0 + k) * GZ_PIECE;
          try {
            if (!gz_member(p + off, (size_t)std::min<uint64_t>(GZ_PIECE, n_bytes - off), w->level, out[k])) bad = 1;
          } catch (...) { bad = 1; }      // (out of memory inside a worker: reported, not thrown across the thread)
        }
      };
      const unsigned nt = (unsigned)std::min<size_t>(w->threads, cnt);
      std::vector<std::thread> th;
      for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
      work();
      for (auto &x : th) x.join();
      if (bad) return set_err(DCRX_E_INVALID, "deflate failed");
      for (size_t k = 0; k < cnt; k++)
        if (std::fwrite(out[k].data(), 1, out[k].size(), w->f) != out[k].size()) return set_err(DCRX_E_INVALID, "short write to the gzip output file");
    }
  } catch (...) { return set_err(DCRX_E_NOMEM, "out of memory in dcrx_gzip_write"); }
  return DCRX_OK;
}

extern "C" int dcrx_gzip_close(void *writer) {
  GzWriter *w = static_cast<GzWriter *>(writer);
  if (!w) return DCRX_OK;
  int rc = DCRX_OK;
  if (w->f) {
    // (an empty file is not a gzip stream: the reference's gzip.open writes a header and trailer even for no data)
    if (std::ftell(w->f) == 0) {
      std::vector<uint8_t> out;
      if (!gz_member(reinterpret_cast<const uint8_t *>(""), 0, w->level, out) || std::fwrite(out.data(), 1, out.size(), w->f) != out.size())
        rc = set_err(DCRX_E_INVALID, "short write to the gzip output file");
    }
    if (std::fclose(w->f) != 0 && rc == DCRX_OK) rc = set_err(DCRX_E_INVALID, "closing the gzip output file failed");
  }
  delete w;
  return rc;
}
