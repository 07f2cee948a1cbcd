// dcrx_translate.cpp — dcrx_cdr3_batch: the productivity call and CDR3 extraction of the reference's translate.get_cdr3
// (src/decombinator/translate.py:257-357) for a batch of five-field DCRs, host only (it runs once per
// unique DCR, behind `collapse`).  Written from what the function computes, not from how it is written:
//   sequence     = V region without its last vdel bases + insert + J region from base jdel on          (:296-305)
//   sequence_aa  = its translation, standard table, Biopython's rules for ambiguity codes               (:307-309)
//   in frame     = (len(sequence) - 1) % 3 == 0 — the reference's test, as it stands                    (:312-317)
//   stop codon   = '*' in sequence_aa                                                                   (:320-324)
//   conserved C  = sequence_aa[v_position - 1] is the V gene's residue; the CDR3 then starts there      (:327-335)
//   conserved F  = the J gene's motif occurs in the four residues from j_position on, counted in what
//                  follows the CDR3's start; the CDR3 then ends len + j_position + start + 1            (:338-347)
//   productive   = all four hold; junction_aa / junction are cut only then                              (:350-355)
// Every index and slice behaves as Python's (negative positions count from the end, slices clamp, an index outside the
// string is the reference's IndexError: status 1 here).
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dcrx.h"

namespace dcrx {
int set_err(int code, const char *msg);
}
using dcrx::set_err;

namespace {

// Python's s[start:stop] on a string of n characters: the half-open range it takes
inline void pyslice64(int64_t n, int64_t start, int64_t stop, int64_t &lo, int64_t &hi) {
  if (start < 0) { start += n; if (start < 0) start = 0; }
  if (start > n) start = n;
  if (stop < 0) { stop += n; if (stop < 0) stop = 0; }
  if (stop > n) stop = n;
  lo = start; hi = stop < start ? start : stop;
}

// ---- translation: NCBI table 1, and Bio.Seq.translate's handling of IUPAC ambiguity codes (biopython 1.84) ----
struct Codons {
  char aa[64];          // index 16 a + 4 b + c with T C A G = 0 1 2 3
  uint8_t code[256];    // T C A G -> 0..3, else 255
  uint8_t opts[256];    // bit k set: the letter can stand for base k (0 for letters that are no nucleotide code)
  Codons() {
    static const char *table = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    std::memcpy(aa, table, 64);
    std::memset(code, 255, sizeof code);
    std::memset(opts, 0, sizeof opts);
    const char *b = "TCAG";
    for (int i = 0; i < 4; i++) code[(uint8_t)b[i]] = (uint8_t)i;
    auto set = [&](char c, const char *bases) {
      for (const char *p = bases; *p; p++) opts[(uint8_t)c] |= (uint8_t)(1u << code[(uint8_t)*p]);
    };
    set('A', "A"); set('C', "C"); set('G', "G"); set('T', "T"); set('U', "T");
    set('M', "AC"); set('R', "AG"); set('W', "AT"); set('S', "CG"); set('Y', "CT"); set('K', "GT");
    set('V', "ACG"); set('H', "ACT"); set('D', "AGT"); set('B', "CGT"); set('X', "ACGT"); set('N', "ACGT");
  }
};
const Codons g_codons;

inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// one codon (upper case, U already T); 0 when a letter is no nucleotide code (Biopython raises: "Codon '...' is invalid")
char translate_codon(char a, char b, char c) {
  const Codons &K = g_codons;
  if (a == '-' && b == '-' && c == '-') return '-';      // (Seq.translate's gap = '-': a codon of gaps is a gap, a partial one raises)
  const uint8_t ia = K.code[(uint8_t)a], ib = K.code[(uint8_t)b], ic = K.code[(uint8_t)c];
  if (ia < 4 && ib < 4 && ic < 4) return K.aa[16 * ia + 4 * ib + ic];
  const uint8_t oa = K.opts[(uint8_t)a], ob = K.opts[(uint8_t)b], oc = K.opts[(uint8_t)c];
  if (!oa || !ob || !oc) return 0;
  // the residues of every concrete codon the letters stand for
  uint32_t seen = 0;      // bit per letter 'A'..'Z', bit 26 for '*'
  for (int x = 0; x < 4; x++) if (oa >> x & 1)
    for (int y = 0; y < 4; y++) if (ob >> y & 1)
      for (int z = 0; z < 4; z++) if (oc >> z & 1) {
        const char r = K.aa[16 * x + 4 * y + z];
        seen |= r == '*' ? (1u << 26) : (1u << (r - 'A'));
      }
  auto bit = [](char r) { return 1u << (r - 'A'); };
  if (seen == (1u << 26)) return '*';                 // every reading stops
  if (seen & (1u << 26)) return 'X';                  // stops and residues
  if ((seen & (seen - 1)) == 0) {                     // one residue
    for (int k = 0; k < 26; k++) if (seen >> k & 1) return (char)('A' + k);
  }
  if (seen == (bit('D') | bit('N'))) return 'B';
  if (seen == (bit('E') | bit('Q'))) return 'Z';
  if (seen == (bit('I') | bit('L'))) return 'J';
  return 'X';
}

// ---- the J motif: the subset of Python's `re` the `.translate` files use — literal characters, '.', character classes
// (ranges, a leading '^'), a backslash in front of a literal; anything else (groups, alternation, repetition, anchors) is
// said, not guessed at (DCRX_E_UNSUPPORTED) ----
struct ByteSet {                             // the bytes a position accepts
  std::array<uint64_t, 4> w{{0, 0, 0, 0}};
  void set(int k) { w[(size_t)(k >> 6)] |= 1ull << (k & 63); }
  bool test(uint8_t k) const { return (w[k >> 6] >> (k & 63)) & 1ull; }
  void flip() { for (auto &x : w) x = ~x; }
};
struct Motif {
  std::vector<ByteSet> tok;
  bool ok = true;
  bool parsed = false;                       // (motifs are parsed when a row first asks for them)
};
Motif parse_motif(const char *p, size_t n) {
  Motif m;
  size_t i = 0;
  while (i < n) {
    ByteSet set;
    const char c = p[i];
    if (c == '.') { for (int k = 0; k < 256; k++) if (k != '\n') set.set(k); i++; }
    else if (c == '\\') {
      if (i + 1 >= n) { m.ok = false; return m; }
      const char d = p[i + 1];
      if ((d >= 'a' && d <= 'z') || (d >= 'A' && d <= 'Z') || (d >= '0' && d <= '9')) { m.ok = false; return m; }      // (\d, \w, \1 ...: classes and references)
      set.set((uint8_t)d); i += 2;
    } else if (c == '[') {
      size_t k = i + 1;
      bool neg = false;
      if (k < n && p[k] == '^') { neg = true; k++; }
      bool first = true, closed = false;
      while (k < n) {
        if (p[k] == ']' && !first) { closed = true; k++; break; }
        if (p[k] == '\\' || p[k] == '[') { m.ok = false; return m; }      // (escapes and nested sets inside a class: not taken on)
        if (k + 2 < n && p[k + 1] == '-' && p[k + 2] != ']') {
          const uint8_t lo = (uint8_t)p[k], hi = (uint8_t)p[k + 2];
          if (lo > hi) { m.ok = false; return m; }
          for (int x = lo; x <= hi; x++) set.set(x);
          k += 3;
        } else { set.set((uint8_t)p[k]); k++; }
        first = false;
      }
      if (!closed) { m.ok = false; return m; }
      if (neg) set.flip();
      i = k;
    } else if (std::strchr("()|*+?{}^$", c)) { m.ok = false; return m; }
    else { set.set((uint8_t)c); i++; }
    // a repetition behind the token
    if (i < n && std::strchr("*+?{", p[i])) { m.ok = false; return m; }
    m.tok.push_back(set);
  }
  return m;
}
bool motif_search(const Motif &m, const char *s, int64_t n) {      // re.findall(motif, s) is not empty
  const int64_t k = (int64_t)m.tok.size();
  for (int64_t o = 0; o + k <= n; o++) {
    bool hit = true;
    for (int64_t x = 0; x < k && hit; x++) hit = m.tok[(size_t)x].test((uint8_t)s[o + x]);
    if (hit) return true;
  }
  return false;
}

}  // namespace

extern "C" int64_t dcrx_cdr3_batch(const dcrx_cdr3_genes_t *G, uint64_t n, const int32_t *v, const int32_t *j, const int32_t *vdel,
                                    const int32_t *jdel, const char *ins, const uint64_t *ins_off, dcrx_cdr3_row_t *rows, char *text,
                                    uint64_t text_cap) {
  if (!G || (n && (!v || !j || !vdel || !jdel || !ins_off || !rows))) return set_err(DCRX_E_INVALID, "null argument to dcrx_cdr3_batch");
  if ((G->n_v && (!G->v_regions || !G->v_region_off || !G->v_pos || !G->v_res || !G->v_res_off)) ||
      (G->n_j && (!G->j_regions || !G->j_region_off || !G->j_pos || !G->j_motif || !G->j_motif_off)))
    return set_err(DCRX_E_INVALID, "dcrx_cdr3_batch: a gene table is null");
  try {
    // a J gene's motif is parsed when a row first uses that gene (the reference compiles only the motif it searches with,
    // translate.py:341-343): a motif this parser does not serve costs the rows of that gene their last step, nobody else's
    std::vector<Motif> motifs(G->n_j);
    uint64_t at = 0;
    std::string seq, aa;
    for (uint64_t r = 0; r < n; r++) {
      dcrx_cdr3_row_t R;
      std::memset(&R, 0, sizeof R);
      // the genes, indexed as Python indexes a list
      int64_t vi = v[r], ji = j[r];
      if (vi < 0) vi += G->n_v;
      if (ji < 0) ji += G->n_j;
      if (vi < 0 || vi >= (int64_t)G->n_v || ji < 0 || ji >= (int64_t)G->n_j) { R.status = DCRX_CDR3_INDEX_ERROR; rows[r] = R; continue; }
      const char *vr = G->v_regions + G->v_region_off[vi];
      const int64_t vn = (int64_t)(G->v_region_off[vi + 1] - G->v_region_off[vi]);
      const char *jr = G->j_regions + G->j_region_off[ji];
      const int64_t jn = (int64_t)(G->j_region_off[ji + 1] - G->j_region_off[ji]);
      int64_t lo, hi;
      seq.clear();
      if (vdel[r] == 0) seq.append(vr, (size_t)vn);                                  // :296-299 (vdel == 0 is its own case: [:-0] would be empty)
      else { pyslice64(vn, 0, -(int64_t)vdel[r], lo, hi); seq.append(vr + lo, (size_t)(hi - lo)); }
      if (ins) seq.append(ins + ins_off[r], (size_t)(ins_off[r + 1] - ins_off[r]));
      pyslice64(jn, jdel[r], jn, lo, hi);
      seq.append(jr + lo, (size_t)(hi - lo));
      // translation (str.upper(), U as T; a trailing partial codon is dropped)
      const int64_t sn = (int64_t)seq.size();
      aa.clear();
      bool bad = false;
      for (int64_t i = 0; i + 3 <= sn; i += 3) {
        char c0 = up(seq[(size_t)i]), c1 = up(seq[(size_t)i + 1]), c2 = up(seq[(size_t)i + 2]);
        if (c0 == 'U') c0 = 'T';
        if (c1 == 'U') c1 = 'T';
        if (c2 == 'U') c2 = 'T';
        const char res = translate_codon(c0, c1, c2);
        if (!res) { bad = true; R.bad_codon_at = (uint32_t)i; break; }
        aa.push_back(res);
      }
      if (bad) { R.status = DCRX_CDR3_BAD_CODON; rows[r] = R; continue; }
      const int64_t an = (int64_t)aa.size();
      R.in_frame = ((sn - 1) % 3 == 0) ? 1 : 0;                                      // (Python's %: sn >= 0 here but for the empty sequence, -1 % 3 = 2)
      R.productive = R.in_frame;
      R.stop = aa.find('*') != std::string::npos ? 1 : 0;
      if (R.stop) R.productive = 0;
      // conserved residue of the V gene
      int64_t start = 0;
      {
        int64_t idx = (int64_t)G->v_pos[vi] - 1;
        const int64_t raw = idx;
        if (idx < 0) idx += an;
        if (idx < 0 || idx >= an) { R.status = DCRX_CDR3_INDEX_ERROR; rows[r] = R; continue; }      // (the reference raises IndexError: a sequence shorter than the position)
        const uint32_t rl = G->v_res_off[vi + 1] - G->v_res_off[vi];
        if (rl == 1 && aa[(size_t)idx] == G->v_res[G->v_res_off[vi]]) { start = raw; R.conserved_c = 1; }
        else R.productive = 0;
      }
      // what follows the CDR3's start, and the J motif in it
      int64_t dlo, dhi;
      pyslice64(an, start, an, dlo, dhi);
      const int64_t dn = dhi - dlo;
      const int64_t jp = G->j_pos[ji];
      int64_t slo, shi;
      pyslice64(dn, jp, jp + 4, slo, shi);
      int64_t end = 0;
      Motif &M = motifs[(size_t)ji];
      if (!M.parsed) { M = parse_motif(G->j_motif + G->j_motif_off[ji], G->j_motif_off[ji + 1] - G->j_motif_off[ji]); M.parsed = true; }
      if (!M.ok) {
        // regular-expression syntax beyond literals, '.' and classes: the row keeps everything up to here — its text, the
        // in-frame / stop / conserved-C calls, where the search would look (motif_lo, motif_len inside sequence_aa) — and the
        // caller finishes it with its own regular-expression engine (conserved_f, end_cdr3, productive, the junctions)
        R.status = DCRX_CDR3_MOTIF_LEFT;
        R.junction_aa_off = (uint32_t)(dlo + slo); R.junction_aa_len = (uint32_t)(shi - slo);
      } else if (motif_search(M, aa.data() + dlo + slo, shi - slo)) { end = dn + jp + start + 1; R.conserved_f = 1; }
      else R.productive = 0;
      R.start_cdr3 = (int32_t)start; R.end_cdr3 = (int32_t)end;
      // the row's text: sequence, then sequence_aa; the junctions are slices of them
      R.seq_off = at; R.seq_len = (uint32_t)sn;
      R.aa_off = at + (uint64_t)sn; R.aa_len = (uint32_t)an;
      if (R.productive && R.status == DCRX_CDR3_OK) {
        pyslice64(an, start, end, lo, hi);
        R.junction_aa_off = (uint32_t)lo; R.junction_aa_len = (uint32_t)(hi - lo);
        pyslice64(sn, start * 3, 3 * end, lo, hi);
        R.junction_off = (uint32_t)lo; R.junction_len = (uint32_t)(hi - lo);
      }
      if (text && at + (uint64_t)(sn + an) <= text_cap) {
        std::memcpy(text + at, seq.data(), (size_t)sn);
        std::memcpy(text + at + sn, aa.data(), (size_t)an);
      }
      at += (uint64_t)(sn + an);
      rows[r] = R;
    }
    return (int64_t)at;
  } catch (const std::exception &e) {
    return set_err(DCRX_E_NOMEM, e.what());
  }
}
