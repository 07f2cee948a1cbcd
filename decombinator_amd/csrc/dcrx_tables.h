// dcrx_tables.h — host-side compiled form of one chain's tag tables, and the
// flat device image the kernels read.
//
// Replaces the module globals that the reference's import_tcr_info() builds
// (reference src/decombinator/decombine.py:593-746): v_seqs/j_seqs, the half
// tags, jump_to_end_v/jump_to_start_j, v_regions/j_regions and the six acora
// automata (:722-746), which are merged here into ONE goto-only DFA over
// {A,C,G,T} whose entries carry the output classes of their target state.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dcrx.h"
#include "dcrx_device.h"

namespace dcrx {

struct GeneHost {
  uint32_t n = 0;
  int split = 0;
  std::vector<std::string> tags, half1, half2, regions;  // regions upper-cased
  std::vector<int32_t> jumps;
};

struct HostTables {
  GeneHost g[2];  // 0 = V, 1 = J
  uint32_t n_states = 0;
  uint32_t n_keywords[K_NCLASS] = {0, 0, 0, 0, 0, 0};
  uint32_t max_tag_len = 0;
  bool equal_len_per_automaton = true;
  uint32_t dfa_bytes = 0;
  std::vector<uint8_t> blob;  // device image
  DevTables rel{};            // every pointer field holds its BYTE OFFSET inside blob

  // rel with `base` added to every pointer: what a kernel receives
  DevTables resolve(const uint8_t *base) const;
};

// Returns 0 or a negative dcrx_error; on failure `err` holds the message.
int compile_tables(const dcrx_tagset_t *ts, HostTables *out, std::string *err);

}  // namespace dcrx
