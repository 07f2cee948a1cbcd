// dcrx_synth_core.h — the synthetic-read function, compiled for host and device
// from this one definition (include/dcrx_synth.h describes the mixture).
#pragma once

#include <cstdint>

#include "dcrx_device.h"

#if defined(__HIPCC__)
#define DCRX_HD __host__ __device__
#else
#define DCRX_HD
#endif

namespace dcrx {

struct SynthParams {
  uint64_t seed;
  uint32_t read_len;
  uint32_t p_rearr_u16;  // of 65536
  uint32_t sub_u16;      // of 65536
  uint32_t n_u32;        // of 2^32
};

DCRX_HD inline uint64_t synth_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// k-th 64-bit draw of read `key`
DCRX_HD inline uint64_t synth_draw(uint64_t key, uint32_t k) {
  return synth_mix64(key + 0xD1B54A32D192ED03ull * (uint64_t)(k + 1));
}

DCRX_HD inline int synth_code(uint8_t c) {
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// Writes the packed stored (FASTQ-frame) read into words[0..nw) and returns the
// position of its 'N' (stored frame) or -1.
DCRX_HD inline int synth_read(const GeneDevPtrs &V, const GeneDevPtrs &J, const SynthParams &P, uint64_t idx,
                              uint32_t *words, uint32_t nw) {
  const uint64_t key = synth_mix64(P.seed ^ synth_mix64(idx));
  const int n = (int)P.read_len;
  const uint64_t d0 = synth_draw(key, 0);
  const bool rearr = (uint32_t)(d0 & 0xFFFFu) < P.p_rearr_u16;
  int prefix = 0, vstart = 0, vlen = 0, ins = 0, jdel = 0, jlen = 0;
  const uint8_t *vreg = nullptr, *jreg = nullptr;
  if (rearr) {
    const uint32_t v = (uint32_t)(synth_draw(key, 1) % V.n);
    const uint32_t j = (uint32_t)(synth_draw(key, 2) % J.n);
    const int vdel = (int)(synth_draw(key, 3) % 11u);
    jdel = (int)(synth_draw(key, 4) % 13u);
    ins = (int)(synth_draw(key, 5) % 26u);
    const int vts = 20 + (int)(synth_draw(key, 6) % 41u);
    const int Lv = (int)V.reg_len[v], Lj = (int)J.reg_len[j];
    vreg = V.reg_bytes + V.reg_off[v];
    jreg = J.reg_bytes + J.reg_off[j];
    int start = (Lv - V.jump[v]) - vts;
    if (start < 0) { prefix = -start; start = 0; }
    vstart = start;
    vlen = (Lv - vdel) - start; if (vlen < 0) vlen = 0;
    jlen = Lj - jdel; if (jlen < 0) jlen = 0;
  }
  uint64_t rnd = 0, sub = 0, alt = 0;
  int curw = n > 0 ? (n - 1) >> 4 : 0;
  uint32_t cur = 0;
  for (uint32_t k = (uint32_t)(n > 0 ? curw + 1 : 0); k < nw; k++) words[k] = 0;
  for (int i = 0; i < n; i++) {  // i: sense-frame position
    if ((i & 31) == 0) { rnd = synth_draw(key, 100u + (uint32_t)(i >> 5)); alt = synth_draw(key, 300u + (uint32_t)(i >> 5)); }
    if ((i & 3) == 0) sub = synth_draw(key, 200u + (uint32_t)(i >> 2));
    int c = (int)((rnd >> (2 * (i & 31))) & 3u);
    if (rearr) {
      int x = i - prefix;
      if (x >= 0 && x < vlen) { int g = synth_code(vreg[vstart + x]); if (g >= 0) c = g; }
      else if (x >= vlen + ins && x < vlen + ins + jlen) { int g = synth_code(jreg[jdel + (x - vlen - ins)]); if (g >= 0) c = g; }
    }
    if ((uint32_t)((sub >> (16 * (i & 3))) & 0xFFFFu) < P.sub_u16)
      c = (c + 1 + (int)(((alt >> (2 * (i & 31))) & 3u) % 3u)) & 3;
    // stored read = reverse complement of the sense frame; m runs downwards, so
    // each word is completed before the next one starts
    const int m = n - 1 - i;
    if ((m >> 4) != curw) { words[curw] = cur; cur = 0; curw = m >> 4; }
    cur |= (uint32_t)(c ^ 3) << (2 * (m & 15));
  }
  if (n > 0) words[curw] = cur;
  const uint64_t dn = synth_draw(key, 8);
  if ((uint32_t)(dn & 0xFFFFFFFFu) < P.n_u32 && n > 0) return (int)((dn >> 32) % (uint64_t)n);
  return -1;
}

}  // namespace dcrx
