/*
 * dcrx_synth.h — seeded synthetic reads for benchmarks and at-scale parity
 * checks.  Not part of the reference's interface: the
 * reference has no generator; real use feeds reads from FASTQ through
 * dcrx_pack_reads.  Lives in the same shared library so that the device and
 * the host produce bit-identical reads from (seed, read index) alone: any shard
 * is reproducible on any rank without moving data.
 *
 * Mixture: p_rearranged of the reads are V(D)J amplicons — the tail of a V
 * region with 0..10 nt deleted, 0..25 random inserted nt, a J region with 0..12
 * nt deleted, random filler to read_len, V tag starting 20..60 nt into the read —
 * the rest are uniform random; every base is then substituted with probability
 * sub_rate; the read is stored reverse-complemented (the default `reverse`
 * orientation expects antisense reads, decombine.py:999-1001); with probability
 * n_rate one stored base becomes 'N' (an exception entry).
 */
#ifndef DCRX_SYNTH_H
#define DCRX_SYNTH_H

#include "dcrx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dcrx_synth_cfg {
  uint64_t seed;
  uint32_t read_len;
  float p_rearranged; /* 0.45 */
  float sub_rate;     /* 0.005 */
  float n_rate;       /* 0.0005 */
} dcrx_synth_cfg_t;

/* Packed reads [first_index, first_index+n) into host memory (n*stride bytes). */
int dcrx_synth_reads_host(const dcrx_tables_t *tables, const dcrx_synth_cfg_t *cfg, uint64_t first_index,
                          uint64_t n, uint32_t stride, uint8_t *packed);

/* Same reads generated on the current device into d_packed, asynchronously. */
int dcrx_synth_reads_device(dcrx_tables_t *tables, const dcrx_synth_cfg_t *cfg, uint64_t first_index,
                            uint64_t n, uint32_t stride, uint8_t *d_packed, void *hip_stream);

/* The exception list ('N' bases) of the same range, host side; exc_read is
 * relative to first_index.  Returns the count (may exceed cap) or an error. */
int64_t dcrx_synth_exceptions_host(const dcrx_tables_t *tables, const dcrx_synth_cfg_t *cfg,
                                   uint64_t first_index, uint64_t n, uint32_t *exc_read, uint16_t *exc_pos,
                                   uint8_t *exc_chr, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
