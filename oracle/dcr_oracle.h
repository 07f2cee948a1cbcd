/*
 * dcr_oracle.h — interface of the CPU oracle (test infrastructure only; see
 * the header of dcr_oracle.c).  Counter and status numbering is shared with
 * the product through include/dcrx_codes.h so that the two can be compared
 * field by field.
 */
#ifndef DCR_ORACLE_H
#define DCR_ORACLE_H

#include <stdint.h>
#include "../include/dcrx_codes.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dcro_tables dcro_tables;

/* What dcr() returns (decombine.py:572-581) plus the exit path and the frame.
 * insert = read[ins_start : ins_start + ins_len] in the frame dcr() saw. */
typedef struct dcro_result {
  int32_t status;    /* enum dcrx_status */
  int32_t frame;     /* 0 = reverse (dcr saw revcomp(read)), 1 = forward */
  int32_t v, j, vdel, jdel;
  int32_t ins_start, ins_len;
  int32_t v_start, j_end;
} dcro_result;

dcro_tables *dcro_tables_new(int nv, const char *const *v_tags, const int *v_jumps,
                             const char *const *v_regions, int nj, const char *const *j_tags,
                             const int *j_jumps, const char *const *j_regions,
                             int v_half_split, int j_half_split);
void dcro_tables_free(dcro_tables *t);

void dcro_revcomp(const char *in, int n, char *out); /* out needs n+1 bytes */

int dcro_dcr(const dcro_tables *t, const char *read, int n, int allow_ns, int lenthreshold,
             dcro_result *res, uint64_t *counts /* [DCRX_N_COUNTERS], accumulated */);

int dcro_decombine_read(const dcro_tables *t, const char *vdj, int n, int orientation,
                        int allow_ns, int lenthreshold, dcro_result *res, uint64_t *counts);

void dcro_decombine_batch(const dcro_tables *t, const char *ascii, const uint64_t *offsets,
                          uint64_t n_reads, int orientation, int allow_ns, int lenthreshold,
                          dcro_result *res, uint64_t *counts);

/* n_threads POSIX threads over contiguous slices; passes > 1 repeats the work (timing only). 0 or -1. */
int dcro_decombine_batch_mt(const dcro_tables *t, const char *ascii, const uint64_t *offsets,
                            uint64_t n_reads, int orientation, int allow_ns, int lenthreshold,
                            dcro_result *res, uint64_t *counts, int n_threads, int passes);

int dcro_findall(const dcro_tables *t, int gene, int which, const char *text, int n,
                 int *first_idx, int *start, int cap);

#ifdef __cplusplus
}
#endif
#endif
