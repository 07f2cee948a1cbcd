// Native FASTQ / FASTA batch reader: the host side that feeds dcrx_pack_reads_span.
//
// Behaviour follows the reference's generator readfq (src/decombinator/decombine.py:228-265)
// over a file opened by its opener (opener_check, :118-123: gzip.open for *.gz, else open,
// both in text mode, i.e. with universal newlines):
//   * lines are skipped until one starts with '>' or '@';
//   * name = header without its first character, up to the first SPACE;
//   * sequence lines are joined until a line starts with '@', '+' or '>';
//   * '+' starts the quality: lines are joined until their total length reaches the
//     sequence length (a quality line may start with '@');
//   * no '+' -> a FASTA record (no quality); a truncated quality -> the record is
//     yielded without quality and reading stops;
//   * every line loses its LAST character (the reference writes l[:-1]), which is the
//     newline for all but an unterminated last line — that one loses a real character.
// Records come out in batches as offsets into one library-owned text buffer, so that the
// Python host slices strings only for the reads that decombine.
//
// Fast path (plain four-line FASTQ, the common case): a block of the file holding the batch is
// cut at record starts and its pieces are parsed by several threads with a strict grammar
// ('@' header / one sequence line / '+' line / one quality line at least as long); the batch's
// text buffer is the block itself (no per-line copies).  The pieces must chain — each piece's
// parse has to end exactly where the next one was cut — and any line that does not fit the
// grammar (wrapped records, FASTA, a truncated or unterminated tail, '>' headers) sends the
// whole batch through the line-by-line generator below, which is the definition.
// A plain file is mapped and, while everything so far was strict, parsed straight from the page cache (the
// batch's text is a piece of the mapping); the first block that is not ends that for good and the buffered
// paths go on from the first record not handed out.
// DCRX_FASTQ_SERIAL=1 switches the fast paths off, DCRX_FASTQ_NO_MMAP=1 the mapping (tests compare them).
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <future>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dcrx.h"

namespace dcrx { int set_err(int code, const char *msg); }
using dcrx::set_err;

struct dcrx_fastq {
  gzFile gz = nullptr;
  FILE *fp = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false, pending_cr = false;
  // generator state between batches: the header line seen ahead (readfq's `last`)
  bool have_last = false;
  std::string last;
  bool finished = false;
  // Batch storage, twice: while the caller works on one parsed chunk a worker thread parses the
  // next one into the other (the file is only ever touched by one of them at a time).
  struct Store {
    std::vector<char> text;
    std::vector<uint64_t> name_off, seq_off, qual_off;
    std::vector<uint32_t> name_len, seq_len, qual_len;
    // the fast path's text buffer: a block of the file as it is (grown with realloc, never zero-filled, kept between batches)
    char *raw = nullptr;
    size_t raw_cap = 0, raw_len = 0;
    bool use_raw = false;
    const char *ext = nullptr;      // ... or a piece of the mapped file (not owned)
    size_t ext_len = 0;
    ~Store() { std::free(raw); }
    bool raw_reserve(size_t cap) {
      if (cap <= raw_cap) return true;
      char *p = (char *)std::realloc(raw, cap);
      if (!p) return false;
      raw = p; raw_cap = cap;
      return true;
    }
    void clear() {
      text.clear(); name_off.clear(); seq_off.clear(); qual_off.clear();
      name_len.clear(); seq_len.clear(); qual_len.clear();
      use_raw = false; ext = nullptr; ext_len = 0;
    }
    void emit(uint64_t no, uint32_t nl, uint64_t so, uint32_t sl, uint64_t qo, uint32_t ql) {
      name_off.push_back(no); name_len.push_back(nl);
      seq_off.push_back(so); seq_len.push_back(sl);
      qual_off.push_back(qo); qual_len.push_back(ql);
    }
  };
  Store st[2];
  int cur = 0;               // the store being handed out
  size_t cursor = 0;         // records of st[cur] already handed out
  bool cur_valid = false;
  std::future<int> ahead;    // parse of st[cur ^ 1] in flight
  bool ahead_valid = false;
  const char *err_msg = nullptr;

  int parse(Store &S, uint64_t max_records);
  bool parse_fast(Store &S, uint64_t max_records);
  size_t est_record_bytes = 0;     // running estimate for sizing the fast path's block (0: not known yet)
  bool fast_ok = true;             // false: DCRX_FASTQ_SERIAL was set when the file was opened
  // the plain file mapped: the fast path parses straight from the page cache while everything so far was strict records
  const char *mm = nullptr;
  size_t mm_size = 0, mm_off = 0;      // (mm_size: where reading ends — the file's end, or a shard's: dcrx_fastq_open_range)
  size_t mm_map = 0;                   // bytes mapped
  bool mm_ok = false;
  bool ranged = false;                 // a byte range of a plain file: strict records only, nothing read beyond the range
  int strict_block(const char *b, size_t len, bool at_eof, uint64_t max_records, Store &S, size_t &used);
  bool parse_mapped(Store &S, uint64_t max_records);

  long raw_read(char *dst, size_t cap) {
    if (gz) {
      const int k = gzread(gz, dst, (unsigned)(cap > (1u << 30) ? (1u << 30) : cap));
      return k;
    }
    const size_t k = fread(dst, 1, cap, fp);
    if (k == 0 && ferror(fp)) return -1;
    return (long)k;
  }

  // Appends one chunk to buf[end..), newline-normalised ("\r\n" and lone "\r" -> "\n").
  // Returns false on a read error.
  bool refill() {
    if (pos > 0) {
      std::memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos; pos = 0;
    }
    if (buf.size() - end < (1u << 20)) buf.resize(buf.size() * 2 > (4u << 20) ? buf.size() * 2 : (4u << 20));
    char *dst = buf.data() + end;
    size_t w = 0;
    if (pending_cr) {        // a '\r' ended the previous chunk: room for its '\n' is kept in front
      dst[0] = '\n';         // overwritten below when the chunk starts with '\n'
      w = 1;
    }
    const long k = raw_read(dst + w, buf.size() - end - w - 1);
    if (k < 0) return false;
    if (k == 0) {
      eof = true;
      if (pending_cr) { end += 1; pending_cr = false; }
      return true;
    }
    char *chunk = dst + w;
    size_t n = (size_t)k;
    if (pending_cr) {
      pending_cr = false;
      if (chunk[0] == '\n') { std::memmove(dst, chunk, n); chunk = dst; w = 0; }
    }
    if (std::memchr(chunk, '\r', n)) {
      size_t o = 0;
      for (size_t i = 0; i < n; i++) {
        const char c = chunk[i];
        if (c != '\r') { chunk[o++] = c; continue; }
        if (i + 1 == n) { pending_cr = true; break; }
        if (chunk[i + 1] != '\n') chunk[o++] = '\n';
      }
      n = o;
    }
    end += w + n;
    return true;
  }

  // Reads up to cap - 1 bytes of the file to dst, newline-normalised like refill(); n = bytes stored (0 at the end of
  // the file, then eof is set).  False on a read error.
  bool read_into(char *dst, size_t cap, size_t &n) {
    n = 0;
    if (eof || cap < 2) return true;
    size_t w = 0;
    if (pending_cr) { dst[0] = '\n'; w = 1; }
    const long k = raw_read(dst + w, cap - w - 1);
    if (k < 0) return false;
    if (k == 0) {
      eof = true;
      if (pending_cr) { n = 1; pending_cr = false; }
      return true;
    }
    char *chunk = dst + w;
    size_t m = (size_t)k;
    if (pending_cr) {
      pending_cr = false;
      if (chunk[0] == '\n') { std::memmove(dst, chunk, m); chunk = dst; w = 0; }
    }
    if (std::memchr(chunk, '\r', m)) {
      size_t o = 0;
      for (size_t i = 0; i < m; i++) {
        const char c = chunk[i];
        if (c != '\r') { chunk[o++] = c; continue; }
        if (i + 1 == m) { pending_cr = true; break; }
        if (chunk[i + 1] != '\n') chunk[o++] = '\n';
      }
      m = o;
    }
    n = w + m;
    return true;
  }

  // One line of the file: [p, p+len) INCLUDING its '\n' when it has one.  False at EOF.
  bool next_line(const char *&p, size_t &len, int &err) {
    for (;;) {
      if (pos < end) {
        const char *s = buf.data() + pos;
        const char *nl = (const char *)std::memchr(s, '\n', end - pos);
        if (nl) { p = s; len = (size_t)(nl - s) + 1; pos += len; return true; }
        if (eof) { p = s; len = end - pos; pos = end; return true; }
      } else if (eof) {
        return false;
      }
      if (!refill()) { err = 1; return false; }
    }
  }
};


namespace {

struct Piece {          // records of one piece of a block, offsets relative to the block
  std::vector<uint64_t> name_off, seq_off, qual_off;
  std::vector<uint32_t> name_len, seq_len, qual_len;
  size_t end = 0;       // where the piece's parse stopped (the next record's header, or the block's end)
  bool ok = true;
};

// Strict four-line records from block[from ..): stops before the first record that starts at or after `to`
// or is not complete inside the block (then P.end is that record's start).
void parse_piece(const char *b, size_t from, size_t to, size_t block_end, Piece &P) {
  size_t p = from;
  while (p < to) {
    if (b[p] != '@') { P.ok = false; break; }
    const char *l1 = (const char *)std::memchr(b + p, '\n', block_end - p);
    if (!l1) break;
    const size_t s = (size_t)(l1 - b) + 1;
    const char *l2 = s < block_end ? (const char *)std::memchr(b + s, '\n', block_end - s) : nullptr;
    if (!l2) break;
    const size_t t = (size_t)(l2 - b) + 1;
    const char *l3 = t < block_end ? (const char *)std::memchr(b + t, '\n', block_end - t) : nullptr;
    if (!l3) break;
    const size_t q = (size_t)(l3 - b) + 1;
    const char *l4 = q < block_end ? (const char *)std::memchr(b + q, '\n', block_end - q) : nullptr;
    if (!l4) break;
    const size_t nx = (size_t)(l4 - b) + 1;
    const size_t seq_len = t - 1 - s, qual_len = nx - 1 - q;
    // the generator's rules that a strict record satisfies: a sequence line that is not a header or '+' line, followed
    // directly by the '+' line; one quality line that reaches the sequence's length (readfq :239-260)
    if (seq_len == 0 || b[s] == '@' || b[s] == '+' || b[s] == '>' || b[t] != '+' || qual_len < seq_len || seq_len > 0xFFFFFFF0ull ||
        qual_len > 0xFFFFFFF0ull) { P.ok = false; break; }
    const char *h = b + p + 1;
    const size_t hl = s - 1 - (p + 1);
    const void *sp = std::memchr(h, ' ', hl);
    P.name_off.push_back(p + 1); P.name_len.push_back((uint32_t)(sp ? (size_t)((const char *)sp - h) : hl));
    P.seq_off.push_back(s); P.seq_len.push_back((uint32_t)seq_len);
    P.qual_off.push_back(q); P.qual_len.push_back((uint32_t)qual_len);
    p = nx;
  }
  P.end = p;
}

}  // namespace

// Strict records of block b[0 .. len): cut at record starts, pieces parsed by several threads, chained.  0: the batch
// (up to max_records records, S filled with offsets relative to b, `used` = bytes of b they take); 1: the block holds
// fewer than max_records and the file goes on (est_record_bytes raised: come back with a bigger block); 2: not strict.
int dcrx_fastq::strict_block(const char *b, size_t len, bool at_eof, uint64_t max_records, Store &S, size_t &used) {
  unsigned nt = std::thread::hardware_concurrency() / 2;
  if (nt > 16) nt = 16;
  if (nt < 1) nt = 1;
  // cut at record starts: "\n@" where the line after next starts with '+'
  const unsigned np_want = len < (1u << 20) ? 1u : nt;
  std::vector<size_t> cut{0};
  for (unsigned k = 1; k < np_want; k++) {
    size_t o = len / np_want * k;
    if (o <= cut.back()) continue;
    bool found = false;
    for (int tries = 0; tries < 64 && !found; tries++) {
      const char *nl = (const char *)std::memchr(b + o, '\n', len - o);
      if (!nl || (size_t)(nl - b) + 1 >= len) break;
      o = (size_t)(nl - b) + 1;
      if (b[o] != '@') continue;
      const char *l1 = (const char *)std::memchr(b + o, '\n', len - o);
      const char *l2 = l1 && (size_t)(l1 - b) + 1 < len ? (const char *)std::memchr(l1 + 1, '\n', len - ((size_t)(l1 - b) + 1)) : nullptr;
      if (l2 && (size_t)(l2 - b) + 1 < len && l2[1] == '+') found = true;
    }
    if (found) cut.push_back(o);
  }
  cut.push_back(len);
  const size_t np = cut.size() - 1;
  std::vector<Piece> pieces(np);
  {
    std::vector<std::thread> th;
    for (size_t k = 1; k < np; k++) th.emplace_back(parse_piece, b, cut[k], cut[k + 1], len, std::ref(pieces[k]));
    parse_piece(b, cut[0], cut[1], len, pieces[0]);
    for (auto &t : th) t.join();
  }
  // the pieces must chain, each ending where the next was cut (the last one wherever its last whole record ends)
  uint64_t total = 0;
  bool ok = true;
  for (size_t k = 0; k < np && ok; k++) {
    ok = pieces[k].ok && (k + 1 == np || pieces[k].end == cut[k + 1]);
    total += pieces[k].name_off.size();
  }
  if (!ok) return 2;
  const bool whole_tail = pieces[np - 1].end == len;
  if (total < max_records && !(at_eof && whole_tail)) {
    if (at_eof) return 2;                                       // a tail that is not whole strict records: the generator's business
    if (total > 0) est_record_bytes = len / total + 16;         // records are longer than estimated: a bigger block
    else est_record_bytes *= 4;
    return 1;
  }
  // take max_records of them; the block up to the end of the last one is the batch's text
  const uint64_t take = total < max_records ? total : max_records;
  S.name_off.reserve(take); S.name_len.reserve(take); S.seq_off.reserve(take); S.seq_len.reserve(take);
  S.qual_off.reserve(take); S.qual_len.reserve(take);
  uint64_t left = take;
  used = 0;
  for (size_t k = 0; k < np && left; k++) {
    const Piece &P = pieces[k];
    const size_t m = P.name_off.size() < left ? P.name_off.size() : (size_t)left;
    S.name_off.insert(S.name_off.end(), P.name_off.begin(), P.name_off.begin() + m);
    S.name_len.insert(S.name_len.end(), P.name_len.begin(), P.name_len.begin() + m);
    S.seq_off.insert(S.seq_off.end(), P.seq_off.begin(), P.seq_off.begin() + m);
    S.seq_len.insert(S.seq_len.end(), P.seq_len.begin(), P.seq_len.begin() + m);
    S.qual_off.insert(S.qual_off.end(), P.qual_off.begin(), P.qual_off.begin() + m);
    S.qual_len.insert(S.qual_len.end(), P.qual_len.begin(), P.qual_len.begin() + m);
    if (m) used = (size_t)(P.qual_off[m - 1] + P.qual_len[m - 1]) + 1;      // past the last record's newline
    left -= m;
  }
  return 0;
}

// The next batch straight from the mapped file (no read, no copy): true when S holds it.  The first block that is not
// strict records (or holds a '\r') ends the mapped mode for good: the stream is positioned at the first record not
// handed out and the buffered paths take over.
bool dcrx_fastq::parse_mapped(Store &S, uint64_t max_records) {
  if (!mm_ok || !fast_ok || finished || max_records == 0) return false;
  auto leave = [&]() {
    mm_ok = false;
    if (ranged) {      // a shard is read by the mapped path alone (the buffered paths know no end but the file's)
      finished = true; err_msg = "the FASTQ shard is not plain four-line records (multi-line records, FASTA records, carriage returns): read the file unsharded";
      return false;
    }
    if (fseeko(fp, (off_t)mm_off, SEEK_SET) != 0) { finished = true; err_msg = "cannot seek in the FASTQ file"; }
    pos = end = 0; eof = false; pending_cr = false; have_last = false; last.clear();
    return false;
  };
  if (mm_off >= mm_size) { finished = true; S.clear(); return true; }
  if (est_record_bytes == 0) {
    const size_t n = mm_size - mm_off < (1u << 20) ? mm_size - mm_off : (1u << 20);
    Piece probe;
    parse_piece(mm + mm_off, 0, n, n, probe);
    if (!probe.ok || probe.name_off.empty()) return leave();
    est_record_bytes = probe.end / probe.name_off.size() + 8;
  }
  for (int attempt = 0; attempt < 8; attempt++) {
    const size_t want = (size_t)max_records * est_record_bytes + (1u << 16);
    const size_t len = mm_size - mm_off < want ? mm_size - mm_off : want;
    const char *b = mm + mm_off;
    if (std::memchr(b, '\r', len)) return leave();          // universal newlines: the buffered paths rewrite them
    S.clear();
    size_t used = 0;
    const int st = strict_block(b, len, mm_off + len == mm_size, max_records, S, used);
    if (st == 2) return leave();
    if (st == 1) continue;
    S.ext = b; S.ext_len = used;
    mm_off += used;
    if (!S.name_off.empty()) est_record_bytes = used / S.name_off.size() + 8;
    if (mm_off == mm_size) finished = true;
    return true;
  }
  return leave();
}

// The next batch by the fast path: true when S holds it (S.text = the block's bytes, file position advanced);
// false with nothing consumed when the block does not fit the strict grammar (or the path does not apply).
bool dcrx_fastq::parse_fast(Store &S, uint64_t max_records) {
  if (!fast_ok || finished || (have_last && !last.empty()) || max_records == 0) return false;
  have_last = false; last.clear();
  // the block is assembled in the store's own buffer: first what the stream buffer still holds, then the file itself
  S.clear();
  S.raw_len = 0;
  auto give_back = [&]() {            // the block returns to the stream buffer: the generator goes on from there
    if (buf.size() < S.raw_len + (1u << 20)) buf.resize(S.raw_len + (4u << 20));
    if (S.raw_len) std::memcpy(buf.data(), S.raw, S.raw_len);
    pos = 0; end = S.raw_len;
    S.raw_len = 0;
    return false;
  };
  if (!S.raw_reserve((end - pos) + (1u << 20))) return false;
  if (end > pos) { std::memcpy(S.raw, buf.data() + pos, end - pos); S.raw_len = end - pos; }
  pos = end = 0;
  if (est_record_bytes == 0) {        // a first look at the file: the size of its records from the first megabyte
    while (!eof && S.raw_len < (1u << 20)) {
      if (!S.raw_reserve((2u << 20))) return give_back();
      size_t n = 0;
      if (!read_into(S.raw + S.raw_len, (1u << 20) + 2 - S.raw_len, n)) return give_back();
      S.raw_len += n;
    }
    Piece probe;
    parse_piece(S.raw, 0, S.raw_len, S.raw_len, probe);
    if (!probe.ok || probe.name_off.empty()) return give_back();
    est_record_bytes = probe.end / probe.name_off.size() + 8;
  }
  for (int attempt = 0; attempt < 8; attempt++) {
    // enough of the file for the batch (by the running estimate of a record's size), or all that is left
    const size_t want = (size_t)max_records * est_record_bytes + (1u << 16);
    while (!eof && S.raw_len < want) {
      if (!S.raw_reserve(want + (1u << 20))) return give_back();
      size_t n = 0;
      if (!read_into(S.raw + S.raw_len, S.raw_cap - S.raw_len, n)) return give_back();      // the generator reports the read error
      S.raw_len += n;
    }
    if (S.raw_len == 0) return give_back();        // end of file: the generator finishes
    const char *b = S.raw;
    const size_t len = S.raw_len;
    size_t used = 0;
    const int st = strict_block(b, len, eof, max_records, S, used);
    if (st == 2) return give_back();
    if (st == 1) continue;
    // what lies behind the batch goes back to the stream buffer (little, once the estimate has settled)
    const size_t rest = len - used;
    if (buf.size() < rest + (1u << 20)) buf.resize(rest + (4u << 20));
    if (rest) std::memcpy(buf.data(), b + used, rest);
    pos = 0; end = rest;
    S.raw_len = used;
    S.use_raw = true;
    if (!S.name_off.empty()) est_record_bytes = used / S.name_off.size() + 8;
    if (eof && rest == 0) finished = true;      // the generator would find no further header
    return true;
  }
  return give_back();
}

int dcrx_fastq::parse(Store &S, uint64_t max_records) {
  dcrx_fastq *f = this;
  if (!gz && (parse_mapped(S, max_records) || parse_fast(S, max_records))) return DCRX_OK;
  if (finished && err_msg) return DCRX_E_INVALID;
  S.clear();
  int err = 0;
  const char *p; size_t len;
  while (!f->finished && S.name_off.size() < max_records) {
    if (!f->have_last || f->last.empty()) {                   // :231-235  look for the next header
      f->have_last = false;
      while (f->next_line(p, len, err)) {
        if (p[0] == '>' || p[0] == '@') { f->last.assign(p, len - 1); f->have_last = true; break; }
      }
    }
    if (err) { err_msg = "read error in FASTQ file"; return DCRX_E_INVALID; }
    if (!f->have_last || f->last.empty()) { f->finished = true; break; }   // :236-237
    // :238  name = last[1:].partition(" ")[0]
    const uint64_t name_off = S.text.size();
    {
      const char *h = f->last.data() + 1;
      const size_t hl = f->last.size() - 1;
      const void *sp = std::memchr(h, ' ', hl);
      const size_t nl = sp ? (size_t)((const char *)sp - h) : hl;
      S.text.insert(S.text.end(), h, h + nl);
    }
    const uint32_t name_len = (uint32_t)(S.text.size() - name_off);
    f->have_last = false; f->last.clear();
    const uint64_t seq_off = S.text.size();
    while (f->next_line(p, len, err)) {                       // :239-243
      if (p[0] == '@' || p[0] == '+' || p[0] == '>') { f->last.assign(p, len - 1); f->have_last = true; break; }
      S.text.insert(S.text.end(), p, p + len - 1);
    }
    if (err) { err_msg = "read error in FASTQ file"; return DCRX_E_INVALID; }
    const uint64_t seq_len64 = S.text.size() - seq_off;
    if (seq_len64 > 0xFFFFFFF0ull) { err_msg = "sequence too long"; return DCRX_E_INVALID; }
    const uint32_t seq_len = (uint32_t)seq_len64;
    const bool last_true = f->have_last && !f->last.empty();
    if (!last_true || f->last[0] != '+') {                    // :244-247  FASTA record
      S.emit(name_off, name_len, seq_off, seq_len, 0, DCRX_FASTQ_NO_QUAL);
      if (!last_true) { f->finished = true; break; }
      continue;
    }
    const uint64_t qual_off = S.text.size();                 // :248-260
    uint64_t leng = 0;
    bool complete = false;
    while (f->next_line(p, len, err)) {
      S.text.insert(S.text.end(), p, p + len - 1);
      leng += len - 1;
      if (leng >= seq_len) { complete = true; break; }
    }
    if (err) { err_msg = "read error in FASTQ file"; return DCRX_E_INVALID; }
    if (complete) {
      f->have_last = false; f->last.clear();
      S.emit(name_off, name_len, seq_off, seq_len, qual_off, (uint32_t)leng);
    } else {                                                  // :261-263  EOF inside the quality
      S.emit(name_off, name_len, seq_off, seq_len, 0, DCRX_FASTQ_NO_QUAL);
      f->finished = true;
    }
  }
  return DCRX_OK;
}

extern "C" {

int dcrx_fastq_open(const char *path, int gzipped, dcrx_fastq_t **out) {
  if (!path || !out) return set_err(DCRX_E_INVALID, "null argument to dcrx_fastq_open");
  *out = nullptr;
  dcrx_fastq *f = new dcrx_fastq();
  if (gzipped) {
    f->gz = gzopen(path, "rb");
    if (!f->gz) { delete f; return set_err(DCRX_E_INVALID, "cannot open FASTQ file"); }
    gzbuffer(f->gz, 1u << 20);
    // gzip.open raises on a file that is not gzip; zlib would pass it through silently
    char probe;
    const int k = gzread(f->gz, &probe, 1);
    if (k < 0 || (k == 1 && gzdirect(f->gz))) {
      gzclose(f->gz); delete f;
      return set_err(DCRX_E_INVALID, "not a gzipped file");
    }
    if (k == 1) gzungetc(probe, f->gz);   // hands the byte back for the first read
  } else {
    f->fp = std::fopen(path, "rb");
    if (!f->fp) { delete f; return set_err(DCRX_E_INVALID, "cannot open FASTQ file"); }
  }
  f->buf.resize(4u << 20);
  f->fast_ok = std::getenv("DCRX_FASTQ_SERIAL") == nullptr;
  if (f->fp && f->fast_ok && !std::getenv("DCRX_FASTQ_NO_MMAP")) {
    struct stat sb;
    if (fstat(fileno(f->fp), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
      void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fileno(f->fp), 0);
      if (m != MAP_FAILED) {
        f->mm = (const char *)m; f->mm_size = f->mm_map = (size_t)sb.st_size; f->mm_ok = true;
        (void)madvise(m, f->mm_size, MADV_SEQUENTIAL);
      }
    }
  }
  *out = f;
  return DCRX_OK;
}

// A byte range [begin, end) of a plain four-line FASTQ file, `begin` a record's first byte (dcrx_fastq_lines finds such
// offsets): what one rank of a sharded stage reads — its records and nothing else of the file.
int dcrx_fastq_open_range(const char *path, uint64_t begin, uint64_t end, dcrx_fastq_t **out) {
  if (!path || !out) return set_err(DCRX_E_INVALID, "null argument to dcrx_fastq_open_range");
  *out = nullptr;
  if (end < begin) return set_err(DCRX_E_INVALID, "dcrx_fastq_open_range: end < begin");
  dcrx_fastq *f = new (std::nothrow) dcrx_fastq();
  if (!f) return set_err(DCRX_E_NOMEM, "out of memory");
  f->fp = std::fopen(path, "rb");
  if (!f->fp) { delete f; return set_err(DCRX_E_INVALID, "cannot open FASTQ file"); }
  struct stat sb;
  if (fstat(fileno(f->fp), &sb) != 0 || !S_ISREG(sb.st_mode) || (uint64_t)sb.st_size < end) {
    std::fclose(f->fp); delete f;
    return set_err(DCRX_E_INVALID, "dcrx_fastq_open_range: the range lies beyond the file");
  }
  f->ranged = true;
  f->fast_ok = true;
  if (end == begin) { f->finished = true; *out = f; return DCRX_OK; }
  // (the mapping starts on the page that holds `begin`; the pages of other ranks' ranges are never touched)
  const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE), base = begin / page * page;
  void *m = mmap(nullptr, (size_t)(end - base), PROT_READ, MAP_PRIVATE, fileno(f->fp), (off_t)base);
  if (m == MAP_FAILED) { std::fclose(f->fp); delete f; return set_err(DCRX_E_INVALID, "cannot map the FASTQ file"); }
  f->mm = (const char *)m; f->mm_map = (size_t)(end - base); f->mm_size = (size_t)(end - base); f->mm_off = (size_t)(begin - base);
  f->mm_ok = true;
  (void)madvise(m, f->mm_map, MADV_SEQUENTIAL);
  *out = f;
  return DCRX_OK;
}

// Newlines of the bytes [begin, end) of a file (mapped, counted at memory speed), and — nth >= 1 — the offset just behind the
// nth of them (where line number nth of the range starts; *nth_off = UINT64_MAX when the range holds fewer); *has_cr tells
// whether the range holds a carriage return (such a file is not read in shards).  How the ranks of a sharded stage agree on
// record boundaries without any of them reading another's bytes: record k of a plain four-line file starts at line 4 k.
int dcrx_fastq_lines(const char *path, uint64_t begin, uint64_t end, uint64_t nth, uint64_t *n_lines, uint64_t *nth_off, int *has_cr,
                     uint64_t *file_size) {
  if (!path || (!n_lines && !nth_off)) return set_err(DCRX_E_INVALID, "null argument to dcrx_fastq_lines");
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return set_err(DCRX_E_INVALID, "cannot open FASTQ file");
  struct stat sb;
  if (fstat(fileno(fp), &sb) != 0 || !S_ISREG(sb.st_mode)) { std::fclose(fp); return set_err(DCRX_E_INVALID, "not a regular file"); }
  if (file_size) *file_size = (uint64_t)sb.st_size;
  if (end > (uint64_t)sb.st_size) end = (uint64_t)sb.st_size;
  if (n_lines) *n_lines = 0;
  if (nth_off) *nth_off = UINT64_MAX;
  if (has_cr) *has_cr = 0;
  if (begin >= end) { std::fclose(fp); return DCRX_OK; }
  const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE), base = begin / page * page;
  void *m = mmap(nullptr, (size_t)(end - base), PROT_READ, MAP_PRIVATE, fileno(fp), (off_t)base);
  if (m == MAP_FAILED) { std::fclose(fp); return set_err(DCRX_E_INVALID, "cannot map the FASTQ file"); }
  (void)madvise(m, (size_t)(end - base), MADV_SEQUENTIAL);
  const char *b = (const char *)m + (begin - base), *e = (const char *)m + (end - base);
  uint64_t n = 0;
  for (const char *p = b; p < e;) {
    const char *q = (const char *)std::memchr(p, '\n', (size_t)(e - p));
    if (!q) break;
    n++;
    if (nth_off && n == nth) { *nth_off = begin + (uint64_t)(q + 1 - b); if (!n_lines) break; }      // (n_lines == NULL: only the offset is wanted)
    p = q + 1;
  }
  if (has_cr && n_lines && std::memchr(b, '\r', (size_t)(e - b))) *has_cr = 1;
  if (n_lines) *n_lines = n;
  munmap(m, (size_t)(end - base));
  std::fclose(fp);
  return DCRX_OK;
}

void dcrx_fastq_close(dcrx_fastq_t *f) {
  if (!f) return;
  if (f->ahead_valid) { f->ahead.get(); f->ahead_valid = false; }
  if (f->gz) gzclose(f->gz);
  // (unmapping gigabytes of touched file pages takes tens of milliseconds — 44 ms of a 330 ms stage over 4 M read pairs; a
  // detached thread doing it only moves the time: the process's other threads wait for the address space's lock meanwhile)
  if (f->mm) munmap(const_cast<char *>(f->mm), f->mm_map);
  if (f->fp) std::fclose(f->fp);
  delete f;
}

int dcrx_fastq_next(dcrx_fastq_t *f, uint64_t max_records, dcrx_fastq_batch_t *out) {
  if (!f || !out) return set_err(DCRX_E_INVALID, "null argument to dcrx_fastq_next");
  std::memset(out, 0, sizeof *out);
  if (max_records == 0) return DCRX_OK;
  if (!f->cur_valid || f->cursor >= f->st[f->cur].name_off.size()) {
    // the current chunk is used up: take the one parsed ahead, or parse now
    int rc;
    if (f->ahead_valid) {
      rc = f->ahead.get();
      f->ahead_valid = false;
      f->cur ^= 1;
    } else {
      try { rc = f->parse(f->st[f->cur], max_records); }
      catch (...) { f->err_msg = "out of memory while reading the FASTQ file"; rc = DCRX_E_NOMEM; }
    }
    if (rc != DCRX_OK) return set_err(rc, f->err_msg ? f->err_msg : "FASTQ reader failed");
    f->cur_valid = true;
    f->cursor = 0;
    if (!f->finished) {      // read ahead while the caller works on this chunk
      dcrx_fastq::Store *other = &f->st[f->cur ^ 1];
      f->ahead = std::async(std::launch::async, [f, other, max_records]() -> int {
        try { return f->parse(*other, max_records); }
        catch (...) { f->err_msg = "out of memory while reading the FASTQ file"; return DCRX_E_NOMEM; }
      });
      f->ahead_valid = true;
    }
  }
  const dcrx_fastq::Store &S = f->st[f->cur];
  const size_t have = S.name_off.size() - f->cursor;
  const size_t k = have < max_records ? have : (size_t)max_records;
  const size_t c = f->cursor;
  out->n_records = k;
  out->text = S.ext ? S.ext : (S.use_raw ? S.raw : S.text.data());
  out->text_bytes = S.ext ? S.ext_len : (S.use_raw ? S.raw_len : S.text.size());
  out->name_off = S.name_off.data() + c; out->name_len = S.name_len.data() + c;
  out->seq_off = S.seq_off.data() + c; out->seq_len = S.seq_len.data() + c;
  out->qual_off = S.qual_off.data() + c; out->qual_len = S.qual_len.data() + c;
  f->cursor += k;
  return DCRX_OK;
}

uint64_t dcrx_count_prefix_byte(const char *text, const uint64_t *start, const uint32_t *len, uint64_t n,
                                uint32_t prefix, int byte) {
  if (!text || !start || !len) return 0;
  auto range = [&](uint64_t lo, uint64_t hi) {
    uint64_t k = 0;
    for (uint64_t r = lo; r < hi; r++) {
      const uint32_t m = len[r] < prefix ? len[r] : prefix;
      if (m && std::memchr(text + start[r], byte, m)) k++;
    }
    return k;
  };
  // (a batch of a million spans: a cache miss per span; a few threads take a range each)
  unsigned nt = n >= (1u << 17) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (const char *e = std::getenv("DCRX_HOST_THREADS")) nt = std::max(1, std::min(8, std::atoi(e)));
  if (nt == 1) return range(0, n);
  // (threads that have started are always joined: a std::thread that is destroyed while joinable ends the process before any
  // handler runs; a range whose thread could not be had is counted here)
  std::vector<uint64_t> part(nt, 0);
  std::vector<std::thread> th;
  unsigned started = 1;                       // ranges [1, started) run on threads of their own
  try {
    th.reserve(nt);
    for (; started < nt; started++) {
      const unsigned t = started;
      th.emplace_back([&part, &range, n, nt, t] { part[t] = range(n * t / nt, n * (t + 1) / nt); });
    }
  } catch (...) {}                            // (no thread to be had: the rest of the ranges in this one)
  part[0] = range(0, n / nt);
  for (unsigned t = started; t < nt; t++) part[t] = range(n * t / nt, n * (t + 1) / nt);
  for (auto &x : th) x.join();
  uint64_t k = 0;
  for (unsigned t = 0; t < nt; t++) k += part[t];
  return k;
}

}  // extern "C"
