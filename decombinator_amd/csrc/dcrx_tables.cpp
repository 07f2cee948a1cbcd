// dcrx_tables.cpp — compiles one chain's tag set into the device image.
//
// Host-only (no HIP calls).  Follows what the reference builds at
// src/decombinator/decombine.py:657-661 (half splits), :690-696 (regions,
// upper-cased), :820-866 (tags, jumps, half tags) and :722-746 (six acora
// automata), but merges the six automata into one goto-only DFA: Aho-Corasick
// reports every occurrence of every keyword independently of which other
// keywords share the machine, so the per-class hit lists of the merged machine
// are exactly the six findall() results.
#include "dcrx_tables.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <queue>

namespace dcrx {

namespace {

int base_code(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

// Python s[a:b]
std::string pyslice(const std::string &s, long a, long b) {
  long n = (long)s.size();
  if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
  if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
  if (b < a) b = a;
  return s.substr((size_t)a, (size_t)(b - a));
}

struct Blob {
  std::vector<uint8_t> bytes;
  uint64_t reserve(size_t n) {  // 16-byte aligned
    size_t off = (bytes.size() + 15) & ~(size_t)15;
    bytes.resize(off + n, 0);
    return off;
  }
  template <typename T>
  uint64_t put(const std::vector<T> &v) {
    uint64_t off = reserve(v.size() * sizeof(T) + 16);  // +16: tail padding for word-pair loads
    if (!v.empty()) std::memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};

template <typename T>
const T *as_off(uint64_t off) { return reinterpret_cast<const T *>(static_cast<uintptr_t>(off)); }

void pack_region(const std::string &s, bool rc, std::vector<uint32_t> *out) {
  size_t n = s.size();
  size_t words = n / 16 + 2;  // one spare word for the funnel-shift pair load
  size_t base = out->size();
  out->resize(base + words, 0);
  for (size_t i = 0; i < n; i++) {
    int c;
    if (!rc) c = base_code(s[i]);
    else { c = base_code(s[n - 1 - i]); if (c >= 0) c ^= 3; }
    if (c < 0) c = 0;
    (*out)[base + i / 16] |= (uint32_t)c << (2 * (i % 16));
  }
}

}  // namespace

DevTables HostTables::resolve(const uint8_t *base) const {
  DevTables d = rel;
  auto fix = [&](auto &p) {
    using P = std::remove_reference_t<decltype(p)>;
    p = reinterpret_cast<P>(base + reinterpret_cast<uintptr_t>(p));
  };
  fix(d.image); fix(d.trans); if (d.dfa16_bytes) fix(d.trans16); fix(d.st_full); fix(d.st_out); fix(d.outs);
  fix(d.kw_base); fix(d.kw_first); fix(d.kw_begin); fix(d.kw_tags); fix(d.comp);
  for (int g = 0; g < 2; g++) {
    fix(d.g[g].tag_len); fix(d.g[g].jump); fix(d.g[g].tag_ascii); fix(d.g[g].tag_pk_fwd); fix(d.g[g].tag_pk_rc); fix(d.g[g].reg_off);
    fix(d.g[g].reg_len); fix(d.g[g].reg_bytes); fix(d.g[g].reg_pk_off); fix(d.g[g].reg_pk);
    fix(d.g[g].reg_pk_rc); fix(d.g[g].reg_clean); fix(d.g[g].w64_fwd); fix(d.g[g].w64_rc); fix(d.g[g].w64_ok);
  }
  if (d.v2_ok) for (int o = 0; o < 2; o++) { fix(d.v2[o].trans); fix(d.v2[o].bk); }
  return d;
}

int compile_tables(const dcrx_tagset_t *ts, HostTables *out, std::string *err) {
  auto fail = [&](int code, const std::string &m) { *err = m; return code; };
  if (!ts || !out) return fail(DCRX_E_INVALID, "null tag set");
  if (ts->n_v == 0 || ts->n_j == 0) return fail(DCRX_E_INVALID, "empty V or J tag list");
  if (ts->n_v > 65534 || ts->n_j > 65534) return fail(DCRX_E_UNSUPPORTED, "more than 65534 genes");

  HostTables &H = *out;
  H = HostTables();
  const char *const *tags_in[2] = {ts->v_tags, ts->j_tags};
  const int32_t *jumps_in[2] = {ts->v_jumps, ts->j_jumps};
  const char *const *regs_in[2] = {ts->v_regions, ts->j_regions};
  const uint32_t n_in[2] = {ts->n_v, ts->n_j};
  const int split_in[2] = {ts->v_half_split, ts->j_half_split};
  const char *gname[2] = {"V", "J"};

  for (int g = 0; g < 2; g++) {
    GeneHost &G = H.g[g];
    G.n = n_in[g];
    G.split = split_in[g];
    if (!tags_in[g] || !jumps_in[g] || !regs_in[g]) return fail(DCRX_E_INVALID, "null tag/jump/region array");
    for (uint32_t k = 0; k < G.n; k++) {
      if (!tags_in[g][k] || !regs_in[g][k]) return fail(DCRX_E_INVALID, "null tag or region string");
      std::string t = tags_in[g][k];
      if (t.empty() || t.size() > MAX_TAG_LEN)
        return fail(DCRX_E_UNSUPPORTED, std::string(gname[g]) + " tag " + std::to_string(k) + ": length must be 1..32");
      for (char c : t)
        if (base_code(c) < 0)
          return fail(DCRX_E_UNSUPPORTED, std::string(gname[g]) + " tag " + std::to_string(k) + " has a character outside ACGT");
      if (G.split <= 0 || (size_t)G.split >= t.size())
        return fail(DCRX_E_UNSUPPORTED, std::string(gname[g]) + " tag " + std::to_string(k) + ": half split leaves an empty half tag");
      int32_t jump = jumps_in[g][k];
      if (jump < -32768 || jump > 32767) return fail(DCRX_E_UNSUPPORTED, "jump outside int16 range");
      // a decombined read reports vdel <= jump_v - len(tag) and jdel <= jump_j (decombine.py:561-565) in a byte
      if (g == 0 && jump - (int)t.size() > 255) return fail(DCRX_E_UNSUPPORTED, "V jump - tag length > 255");
      if (g == 1 && jump > 255) return fail(DCRX_E_UNSUPPORTED, "J jump > 255");
      std::string r = regs_in[g][k];
      if (r.size() > 65535) return fail(DCRX_E_UNSUPPORTED, "region longer than 65535");
      for (char &c : r) if (c >= 'a' && c <= 'z') c = (char)(c - 32);  // .seq.upper(), decombine.py:695
      G.tags.push_back(t);
      G.half1.push_back(pyslice(t, 0, G.split));                 // :841 / :863
      G.half2.push_back(pyslice(t, G.split, (long)t.size()));    // :842 / :864
      G.jumps.push_back(jump);
      G.regions.push_back(r);
      H.max_tag_len = std::max<uint32_t>(H.max_tag_len, (uint32_t)t.size());
    }
  }

  // ---- distinct keywords per class (AcoraBuilder keeps a set) ----------------
  // kw_strings[c][i], kw_tags[c][i] = ascending tag indices holding that string
  std::vector<std::string> kw_str[K_NCLASS];
  std::vector<std::vector<uint32_t>> kw_idx[K_NCLASS];
  auto add_class = [&](int cls, const std::vector<std::string> &list) {
    std::map<std::string, int> seen;
    for (uint32_t k = 0; k < list.size(); k++) {
      auto it = seen.find(list[k]);
      if (it == seen.end()) {
        seen[list[k]] = (int)kw_str[cls].size();
        kw_str[cls].push_back(list[k]);
        kw_idx[cls].push_back({k});
      } else {
        kw_idx[cls][it->second].push_back(k);
      }
    }
    H.n_keywords[cls] = (uint32_t)kw_str[cls].size();
    for (auto &s : kw_str[cls])
      if (s.size() != kw_str[cls][0].size()) H.equal_len_per_automaton = false;
  };
  add_class(K_VFULL, H.g[0].tags); add_class(K_JFULL, H.g[1].tags);
  add_class(K_VH1, H.g[0].half1);  add_class(K_VH2, H.g[0].half2);
  add_class(K_JH1, H.g[1].half1);  add_class(K_JH2, H.g[1].half2);

  // ---- trie ---------------------------------------------------------------------
  struct Node { int next[4] = {-1, -1, -1, -1}; int fail = 0; int out_link = -1; int depth = 0;
                std::vector<std::pair<int, int>> own; };  // own: (class, kw) ending exactly here
  std::vector<Node> nodes(1);
  for (int cls = 0; cls < K_NCLASS; cls++) {
    for (size_t i = 0; i < kw_str[cls].size(); i++) {
      int s = 0;
      for (char ch : kw_str[cls][i]) {
        int c = base_code(ch);
        if (nodes[s].next[c] < 0) {
          nodes[s].next[c] = (int)nodes.size();
          Node nn; nn.depth = nodes[s].depth + 1;
          nodes.push_back(nn);
        }
        s = nodes[s].next[c];
      }
      nodes[s].own.push_back({cls, (int)i});
    }
  }
  if (nodes.size() > MAX_STATES)
    return fail(DCRX_E_UNSUPPORTED, "merged automaton needs " + std::to_string(nodes.size()) + " states (limit 16383)");
  const uint32_t S = (uint32_t)nodes.size();
  H.n_states = S;

  // BFS: failure links, output links, full goto function
  std::vector<int> delta(S * 4, 0);
  {
    std::queue<int> q;
    for (int c = 0; c < 4; c++) {
      int t = nodes[0].next[c];
      if (t >= 0) { delta[c] = t; nodes[t].fail = 0; q.push(t); } else delta[c] = 0;
    }
    while (!q.empty()) {
      int s = q.front(); q.pop();
      int f = nodes[s].fail;
      nodes[s].out_link = !nodes[f].own.empty() ? f : nodes[f].out_link;
      for (int c = 0; c < 4; c++) {
        int t = nodes[s].next[c];
        if (t >= 0) { delta[s * 4 + c] = t; nodes[t].fail = delta[f * 4 + c]; q.push(t); }
        else delta[s * 4 + c] = delta[f * 4 + c];
      }
    }
  }

  // per-state output lists (own keywords, then the out_link chain: longest first)
  std::vector<uint32_t> kw_base(K_NCLASS + 1, 0);
  for (int c = 0; c < K_NCLASS; c++) kw_base[c + 1] = kw_base[c] + (uint32_t)kw_str[c].size();
  std::vector<uint32_t> st_out(S + 1, 0), outs, st_full(S, 0xFFFFFFFFu), st_flags(S, 0);
  for (uint32_t s = 0; s < S; s++) {
    st_out[s] = (uint32_t)outs.size();
    int cnt[K_NCLASS] = {0, 0, 0, 0, 0, 0};
    uint32_t vfull = 0xFFFF, jfull = 0xFFFF;
    for (int t = nodes[s].own.empty() ? nodes[s].out_link : (int)s; t >= 0; t = nodes[t].out_link) {
      // within one node order classes deterministically; only same-class order matters
      auto own = nodes[t].own;
      std::sort(own.begin(), own.end());
      for (auto &o : own) {
        outs.push_back(out_pack(o.first, nodes[t].depth, o.second, false));
        if (o.first == K_VFULL && cnt[K_VFULL] == 0) vfull = kw_idx[K_VFULL][o.second][0];  // v_seqs.index(tag), :282
        if (o.first == K_JFULL && cnt[K_JFULL] == 0) jfull = kw_idx[K_JFULL][o.second][0];  // j_seqs.index(tag), :406
        cnt[o.first]++;
      }
    }
    if (outs.size() > st_out[s]) outs.back() |= 0x80000000u;
    st_full[s] = vfull | (jfull << 16);
    uint32_t fl = 0;
    if (cnt[K_VFULL] >= 1) fl |= 1u << TE_VFULL_BIT;
    if (cnt[K_JFULL] >= 1) fl |= 1u << TE_JFULL_BIT;
    if (cnt[K_VH1]) fl |= 1u << TE_VH1_BIT;
    if (cnt[K_VH2]) fl |= 1u << TE_VH2_BIT;
    if (cnt[K_JH1]) fl |= 1u << TE_JH1_BIT;
    if (cnt[K_JH2]) fl |= 1u << TE_JH2_BIT;
    if (cnt[K_VFULL] >= 2) fl |= 1u << TE_VMULTI_BIT;
    if (cnt[K_JFULL] >= 2) fl |= 1u << TE_JMULTI_BIT;
    st_flags[s] = fl;
  }
  st_out[S] = (uint32_t)outs.size();
  for (auto &o : outs) {
    if (((o >> 9) & 0x3FFFFF) > 0xFFFF) return fail(DCRX_E_UNSUPPORTED, "keyword id overflow");
  }

  // Renumber: states without any output first (the root stays 0), output states last, so that
  // the per-output-state side tables are small and indexed by state - first_out.
  std::vector<uint32_t> new_id(S), old_of(S);
  uint32_t first_out = 0;
  {
    uint32_t nx = 0;
    for (uint32_t s = 0; s < S; s++) if (!st_flags[s]) { new_id[s] = nx; old_of[nx++] = s; }
    first_out = nx;
    for (uint32_t s = 0; s < S; s++) if (st_flags[s]) { new_id[s] = nx; old_of[nx++] = s; }
  }
  const uint32_t NO = S - first_out;
  std::vector<uint32_t> st_full_c(NO ? NO : 1, 0xFFFFFFFFu), st_out_c(NO + 1, 0), outs_c;
  for (uint32_t o = 0; o < NO; o++) {
    const uint32_t s = old_of[first_out + o];
    st_full_c[o] = st_full[s];
    st_out_c[o] = (uint32_t)outs_c.size();
    for (uint32_t x = st_out[s]; x < st_out[s + 1]; x++) outs_c.push_back(outs[x]);
  }
  st_out_c[NO] = (uint32_t)outs_c.size();

  std::vector<uint32_t> trans(S * 4);
  for (uint32_t s = 0; s < S; s++)
    for (int c = 0; c < 4; c++) {
      const uint32_t t = (uint32_t)delta[s * 4 + c];
      trans[new_id[s] * 4 + c] = (new_id[t] * 16u) | st_flags[t];
    }
  H.dfa_bytes = S * 16u;

  // keyword -> tags tables (global keyword id = kw_base[class] + kw)
  std::vector<uint32_t> kw_first, kw_begin, kw_tags;
  for (int c = 0; c < K_NCLASS; c++)
    for (size_t i = 0; i < kw_str[c].size(); i++) {
      kw_first.push_back(kw_idx[c][i][0]);
      kw_begin.push_back((uint32_t)kw_tags.size());
      for (uint32_t k : kw_idx[c][i]) kw_tags.push_back(k);
    }
  kw_begin.push_back((uint32_t)kw_tags.size());

  // ---- blob: the LDS image first (DFA + every side table the per-read code touches on its
  // common paths), then what only the general/slow paths read from global memory ----------------
  Blob B;
  DevTables &R = H.rel;
  R.n_states = S;
  R.first_out = first_out;
  R.row0 = 0;
  R.max_half_len = 0;
  uint32_t min_kw_len = 0xFFFFFFFFu;
  for (int g = 0; g < 2; g++) {
    for (const std::string &h : H.g[g].half1) { R.max_half_len = std::max<uint32_t>(R.max_half_len, (uint32_t)h.size()); min_kw_len = std::min<uint32_t>(min_kw_len, (uint32_t)h.size()); }
    for (const std::string &h : H.g[g].half2) { R.max_half_len = std::max<uint32_t>(R.max_half_len, (uint32_t)h.size()); min_kw_len = std::min<uint32_t>(min_kw_len, (uint32_t)h.size()); }
    for (const std::string &h : H.g[g].tags) min_kw_len = std::min<uint32_t>(min_kw_len, (uint32_t)h.size());
  }
  R.pair_rescue = 0;      // set below, once the pair table is known to exist
  R.dfa_bytes = H.dfa_bytes;
  R.image = as_off<uint8_t>(0);
  R.trans = as_off<uint32_t>(B.put(trans));
  R.st_full = as_off<uint32_t>(B.put(st_full_c));
  R.st_out = as_off<uint32_t>(B.put(st_out_c));
  R.outs = as_off<uint32_t>(B.put(outs_c));
  R.kw_base = as_off<uint32_t>(B.put(kw_base));
  R.kw_first = as_off<uint32_t>(B.put(kw_first));
  R.kw_begin = as_off<uint32_t>(B.put(kw_begin));
  R.kw_tags = as_off<uint32_t>(B.put(kw_tags));
  struct GeneArrays {
    std::vector<uint8_t> tag_len, tag_ascii, reg_bytes, reg_clean, w64_ok;
    std::vector<int32_t> jump;
    std::vector<uint32_t> reg_off, reg_len, reg_pk_off, reg_pk, reg_pk_rc;
    std::vector<uint64_t> w64_fwd, w64_rc, tag_pk_fwd, tag_pk_rc;
  } GA[2];
  for (int g = 0; g < 2; g++) {
    const GeneHost &G = H.g[g];
    GeneArrays &A = GA[g];
    A.tag_len.resize(G.n); A.tag_ascii.assign(G.n * 32, 0); A.reg_clean.resize(G.n); A.w64_ok.assign(G.n, 0);
    A.jump.resize(G.n); A.reg_off.resize(G.n); A.reg_len.resize(G.n); A.reg_pk_off.resize(G.n);
    A.w64_fwd.assign(G.n, 0); A.w64_rc.assign(G.n, 0); A.tag_pk_fwd.assign(G.n, 0); A.tag_pk_rc.assign(G.n, 0);
    for (uint32_t k = 0; k < G.n; k++) {
      A.tag_len[k] = (uint8_t)G.tags[k].size();
      std::memcpy(&A.tag_ascii[k * 32], G.tags[k].data(), G.tags[k].size());
      A.jump[k] = G.jumps[k];
      for (size_t s = 0; s < G.tags[k].size(); s++) {
        const uint64_t c = (uint64_t)base_code(G.tags[k][s]);
        A.tag_pk_fwd[k] |= c << (2 * s);
        A.tag_pk_rc[k] |= (c ^ 3) << (2 * (G.tags[k].size() - 1 - s));
      }
      A.reg_off[k] = (uint32_t)A.reg_bytes.size();
      A.reg_len[k] = (uint32_t)G.regions[k].size();
      A.reg_bytes.insert(A.reg_bytes.end(), G.regions[k].begin(), G.regions[k].end());
      bool clean = true;
      for (char c : G.regions[k]) if (base_code(c) < 0) clean = false;
      A.reg_clean[k] = clean ? 1 : 0;
      A.reg_pk_off[k] = (uint32_t)A.reg_pk.size();
      pack_region(G.regions[k], false, &A.reg_pk);
      pack_region(G.regions[k], true, &A.reg_pk_rc);
      // walk window: V = last 32 nt (get_v_deletions starts at the 3' end, :754), J = first 32 nt (:793)
      const std::string &r = G.regions[k];
      if (r.size() >= 32) {
        const size_t w0 = (g == 0) ? r.size() - 32 : 0;
        bool ok = true;
        uint64_t f = 0, rcw = 0;
        for (int s = 0; s < 32; s++) {
          int c = base_code(r[w0 + s]);
          if (c < 0) { ok = false; break; }
          f |= (uint64_t)c << (2 * s);
          rcw |= (uint64_t)(c ^ 3) << (2 * (31 - s));
        }
        if (ok) { A.w64_fwd[k] = f; A.w64_rc[k] = rcw; A.w64_ok[k] = 1; }
      }
    }
    GeneDevPtrs &P = R.g[g];
    P.n = G.n;
    P.split = G.split;
    // LDS-image part
    P.tag_len = as_off<uint8_t>(B.put(A.tag_len));
    P.jump = as_off<int32_t>(B.put(A.jump));
    P.tag_pk_fwd = as_off<uint64_t>(B.put(A.tag_pk_fwd));
    P.tag_pk_rc = as_off<uint64_t>(B.put(A.tag_pk_rc));
    P.w64_fwd = as_off<uint64_t>(B.put(A.w64_fwd));
    P.w64_rc = as_off<uint64_t>(B.put(A.w64_rc));
    P.w64_ok = as_off<uint8_t>(B.put(A.w64_ok));
    P.reg_len = as_off<uint32_t>(B.put(A.reg_len));
  }
  R.lds_image_bytes = (uint32_t)((B.bytes.size() + 15) & ~(size_t)15);
  // extended image: the packed germline regions behind the side tables — the event kernel stages them as well when
  // they fit (its exact walks read a region word per step)
  for (int g = 0; g < 2; g++) {
    GeneArrays &A = GA[g];
    GeneDevPtrs &P = R.g[g];
    P.reg_pk_off = as_off<uint32_t>(B.put(A.reg_pk_off));
    P.reg_clean = as_off<uint8_t>(B.put(A.reg_clean));
    P.reg_pk = as_off<uint32_t>(B.put(A.reg_pk));
    P.reg_pk_rc = as_off<uint32_t>(B.put(A.reg_pk_rc));
  }
  R.lds_image2_bytes = (uint32_t)((B.bytes.size() + 15) & ~(size_t)15);
  for (int g = 0; g < 2; g++) {
    GeneArrays &A = GA[g];
    GeneDevPtrs &P = R.g[g];
    P.tag_ascii = as_off<uint8_t>(B.put(A.tag_ascii));
    P.reg_off = as_off<uint32_t>(B.put(A.reg_off));
    P.reg_bytes = as_off<uint8_t>(B.put(A.reg_bytes));
  }
  // two-bases-per-step table for the fast kernel (new numbering throughout)
  R.trans16 = nullptr; R.dfa16_bytes = 0; R.row16_0 = 0;
  // The pair scan counts full-tag hits in a 6-bit field.  64 or more hits of one class in a
  // read of <= 320 nt need occurrences 4 or fewer bases apart, i.e. two tags of the class (or
  // one tag with itself) that overlap consistently at a shift of 1..4: such sets keep the
  // one-base scan, whose 9-bit count cannot wrap.
  bool dense_overlaps = false;
  for (int g = 0; g < 2 && !dense_overlaps; g++) {
    const auto &tags = H.g[g].tags;
    for (size_t x = 0; x < tags.size() && !dense_overlaps; x++)
      for (size_t y = 0; y < tags.size() && !dense_overlaps; y++)
        for (size_t dsh = 1; dsh <= 4; dsh++) {
          if (dsh >= tags[x].size()) break;
          const size_t ov = std::min(tags[x].size() - dsh, tags[y].size());
          if (tags[x].compare(dsh, ov, tags[y], 0, ov) == 0) { dense_overlaps = true; break; }
        }
  }
  if (S <= MAX_STATES16 && !dense_overlaps) {
    std::vector<uint32_t> trans16((size_t)S * 16);
    auto cnt_bits = [&](uint32_t fl, int full_bit, int multi_bit) { return ((fl >> full_bit) & 1u) + ((fl >> multi_bit) & 1u); };
    for (uint32_t s = 0; s < S; s++)        // s: OLD numbering
      for (int c1 = 0; c1 < 4; c1++)
        for (int c2 = 0; c2 < 4; c2++) {
          const uint32_t s1 = (uint32_t)delta[s * 4 + c1], s2 = (uint32_t)delta[s1 * 4 + c2];
          const uint32_t f1 = st_flags[s1], f2 = st_flags[s2];
          uint32_t e = new_id[s2] * 64u;
          e |= f1 & (0xFu << TE_VH1_BIT);                            // half-tag classes that end at the FIRST base
          const uint32_t nv = cnt_bits(f1, TE_VFULL_BIT, TE_VMULTI_BIT) + cnt_bits(f2, TE_VFULL_BIT, TE_VMULTI_BIT);
          const uint32_t nj = cnt_bits(f1, TE_JFULL_BIT, TE_JMULTI_BIT) + cnt_bits(f2, TE_JFULL_BIT, TE_JMULTI_BIT);
          if (nv >= 1) e |= 1u << TE_VFULL_BIT;
          if (nj >= 1) e |= 1u << TE_JFULL_BIT;
          if (nv >= 2) e |= 1u << TE_VMULTI_BIT;
          if (nj >= 2) e |= 1u << TE_JMULTI_BIT;
          if ((f2 >> TE_VFULL_BIT) & 1u) e |= 1u << TE16_V2_BIT;
          if ((f2 >> TE_JFULL_BIT) & 1u) e |= 1u << TE16_J2_BIT;
          e |= ((f2 >> TE_VH1_BIT) & 0xFu) << TE16_H2_SHIFT;         // ... and at the SECOND base
          trans16[(size_t)new_id[s] * 16 + c1 * 4 + c2] = e;
        }
    R.trans16 = as_off<uint32_t>(B.put(trans16));
    R.dfa16_bytes = S * 64u;
    R.pair_rescue = (R.max_half_len <= 15 && min_kw_len >= 2) ? 1u : 0u;   // the rescue kernel's window: 16 bases, one more than any half tag
  }
  // ---- v2 scan tables (dcrx_v2_device.h): per frame a filter automaton over the keywords as the
  // STORED read shows them (forward frame: the keywords; reverse frame: their reverse complements),
  // two bases per 16-bit entry, and per class a bucket table of the packed keywords ------------------
  R.v2_ok = 0;
  for (int c = 0; c < K_NCLASS; c++) R.kw_len[c] = kw_str[c].empty() ? 0u : (uint32_t)kw_str[c][0].size();
  for (int o = 0; o < 2; o++) { R.v2[o] = V2Ori{}; }
  if (H.equal_len_per_automaton) {
    bool ok = true;
    for (int o = 0; o < 2 && ok; o++) {
      auto shown = [&](const std::string &k) {        // the keyword as it appears in the stored read
        if (o == 0) return k;
        std::string r(k.rbegin(), k.rend());
        for (char &ch : r) ch = "TGCA"[base_code(ch)];
        return r;
      };
      struct N2 { int next[4] = {-1, -1, -1, -1}; int fail = 0; uint32_t fl = 0; };
      std::vector<N2> nd(1);
      const uint32_t group[K_NCLASS] = {V2_F_VF, V2_F_JF, V2_F_VH, V2_F_VH, V2_F_JH, V2_F_JH};
      for (int cls = 0; cls < K_NCLASS; cls++)
        for (const std::string &k0 : kw_str[cls]) {
          const std::string k = shown(k0);
          int s = 0;
          for (char ch : k) {
            const int c = base_code(ch);
            if (nd[s].next[c] < 0) { nd[s].next[c] = (int)nd.size(); nd.push_back(N2()); }
            s = nd[s].next[c];
          }
          nd[s].fl |= group[cls];
        }
      const uint32_t S2 = (uint32_t)nd.size();
      if (S2 > V2_MAX_STATES) { ok = false; break; }
      std::vector<int> d2(S2 * 4, 0);
      {
        std::queue<int> q;
        for (int c = 0; c < 4; c++) {
          const int t = nd[0].next[c];
          if (t >= 0) { d2[c] = t; nd[t].fail = 0; q.push(t); }
        }
        while (!q.empty()) {
          const int s = q.front(); q.pop();
          const int f = nd[s].fail;
          nd[s].fl |= nd[f].fl;                      // everything that ends at a suffix state ends here too (BFS: f is final)
          for (int c = 0; c < 4; c++) {
            const int t = nd[s].next[c];
            if (t >= 0) { d2[s * 4 + c] = t; nd[t].fail = d2[f * 4 + c]; q.push(t); }
            else d2[s * 4 + c] = d2[f * 4 + c];
          }
        }
      }
      std::vector<uint16_t> tr((size_t)S2 * 16);
      const bool narrow = S2 <= 2047;
      for (uint32_t s = 0; s < S2; s++)
        for (int x = 0; x < 16; x++) {               // x: raw nibble, first base in its low two bits
          const uint32_t s1 = (uint32_t)d2[s * 4 + (x & 3)], s2 = (uint32_t)d2[s1 * 4 + (x >> 2)];
          tr[(size_t)s * 16 + x] = (uint16_t)((s2 << (narrow ? 5 : 4)) | nd[s1].fl | nd[s2].fl);
        }
      // buckets
      Blob K;
      V2Ori &V = R.v2[o];
      for (int cls = 0; cls < K_NCLASS; cls++) {
        const size_t nk = kw_str[cls].size();
        std::vector<uint64_t> pk(nk);
        for (size_t i = 0; i < nk; i++) {
          const std::string k = shown(kw_str[cls][i]);
          uint64_t v = 0;
          for (size_t x = 0; x < k.size(); x++) v |= (uint64_t)base_code(k[x]) << (2 * x);
          pk[i] = v;
        }
        // two-choice placement (cuckoo): a table of at least twice the keywords, doubled until every keyword finds a slot
        uint32_t slots = 8;
        while (slots < 2 * nk) slots <<= 1;
        std::vector<int> owner;
        for (;; slots <<= 1) {
          if (slots > V2_PH_MAX_SLOTS) { ok = false; break; }      // (two keywords with one hash sum: the three-launch form serves the set)
          owner.assign(slots, -1);
          bool placed = true;
          for (size_t i = 0; i < nk && placed; i++) {
            int cur = (int)i;
            uint32_t s1, s2;
            v2_slots(pk[cur], slots - 1, s1, s2);
            uint32_t pos = s1;
            placed = false;
            for (int kick = 0; kick < 512; kick++) {
              if (owner[pos] < 0) { owner[pos] = cur; placed = true; break; }
              std::swap(cur, owner[pos]);
              v2_slots(pk[cur], slots - 1, s1, s2);
              pos = pos == s1 ? s2 : s1;
            }
          }
          if (placed) break;
        }
        if (!ok) break;
        std::vector<uint16_t> slot_kw(slots, (uint16_t)V2_PH_EMPTY), slot_tag(slots, (uint16_t)V2_PH_EMPTY);
        std::vector<uint64_t> slot_pk(slots, ~0ull);
        for (uint32_t at = 0; at < slots; at++)
          if (owner[at] >= 0) { const size_t i = (size_t)owner[at]; slot_kw[at] = (uint16_t)i; slot_pk[at] = pk[i]; slot_tag[at] = (uint16_t)kw_idx[cls][i][0]; }
        V.bk_mask[cls] = slots - 1;
        V.bk_kw_off[cls] = (uint32_t)K.put(slot_kw);
        V.bk_pk_off[cls] = (uint32_t)K.put(slot_pk);
        V.bk_tag_off[cls] = (uint32_t)K.put(slot_tag);
      }
      if (!ok) break;
      K.reserve(16);
      V.n_states = S2;
      V.narrow = narrow ? 1u : 0u;
      V.trans_bytes = S2 * 32u;
      V.trans = as_off<uint16_t>(B.put(tr));
      V.bk_bytes = (uint32_t)K.bytes.size();
      V.bk = as_off<uint8_t>(B.put(K.bytes));
    }
    if (ok) R.v2_ok = 1;
    else for (int o = 0; o < 2; o++) R.v2[o] = V2Ori{};
  }
  {
    // Bio.Seq complement table (ambiguous DNA, both cases, U like T); other bytes unchanged
    std::vector<uint8_t> comp(256);
    for (int c = 0; c < 256; c++) comp[c] = (uint8_t)c;
    const char *ck = "ACGTMRWSYKVHDBXNU", *cv = "TGCAKYWSRMBDHVXNA";
    for (int i = 0; ck[i]; i++) {
      comp[(uint8_t)ck[i]] = (uint8_t)cv[i];
      comp[(uint8_t)(ck[i] + 32)] = (uint8_t)(cv[i] + 32);
    }
    R.comp = as_off<uint8_t>(B.put(comp));
  }
  B.reserve(64);
  H.blob.swap(B.bytes);
  return DCRX_OK;
}

}  // namespace dcrx
