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
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <future>
#include <string>
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
    void clear() {
      text.clear(); name_off.clear(); seq_off.clear(); qual_off.clear();
      name_len.clear(); seq_len.clear(); qual_len.clear();
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

int dcrx_fastq::parse(Store &S, uint64_t max_records) {
  dcrx_fastq *f = this;
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
  *out = f;
  return DCRX_OK;
}

void dcrx_fastq_close(dcrx_fastq_t *f) {
  if (!f) return;
  if (f->ahead_valid) { f->ahead.get(); f->ahead_valid = false; }
  if (f->gz) gzclose(f->gz);
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
  out->text = S.text.data();
  out->text_bytes = S.text.size();
  out->name_off = S.name_off.data() + c; out->name_len = S.name_len.data() + c;
  out->seq_off = S.seq_off.data() + c; out->seq_len = S.seq_len.data() + c;
  out->qual_off = S.qual_off.data() + c; out->qual_len = S.qual_len.data() + c;
  f->cursor += k;
  return DCRX_OK;
}

uint64_t dcrx_count_prefix_byte(const char *text, const uint64_t *start, const uint32_t *len, uint64_t n,
                                uint32_t prefix, int byte) {
  if (!text || !start || !len) return 0;
  uint64_t k = 0;
  for (uint64_t r = 0; r < n; r++) {
    const uint32_t m = len[r] < prefix ? len[r] : prefix;
    if (m && std::memchr(text + start[r], byte, m)) k++;
  }
  return k;
}

}  // extern "C"
