/*
 * dcrx.h — C ABI of the MI355X-native `decombine` hot path.
 *
 * The reference (innate2adaptive/decombinator, pure Python) has no FFI or plugin
 * interface; its narrowest seam around this path is
 *
 *     decombine.import_tcr_info(inputargs)      src/decombinator/decombine.py:593-746
 *     decombine.dcr(read, inputargs) -> 7-list  src/decombinator/decombine.py:534-585
 *     the per-read driver loop                  src/decombinator/decombine.py:963-1050
 *
 * Each entry point below names the reference code it replaces.  A maintainer
 * binds this library with ctypes (INTEGRATION.md shows the stub); all pointers
 * are plain buffers owned by the caller, the only library-owned object is the
 * opaque dcrx_tables_t handle.
 *
 * Conventions
 *   - every function returns 0 on success or a negative dcrx_error; the text of
 *     the last failure on the calling thread is dcrx_last_error();
 *   - nothing throws, aborts or prints;
 *   - a dcrx_tables_t is not thread-safe: serialise calls that share one, and launch the
 *     asynchronous entry points that share one on ONE stream (the handle owns a workspace
 *     that consecutive launches reuse).  A launch may fork part of its work onto a stream the
 *     handle owns and joins it back before its last kernel: to the caller it is ordered on
 *     the stream it was given, like any other work there;
 *   - "device" pointers are HIP device memory on the current device
 *     (dcrx_set_device), "host" pointers ordinary process memory.
 */
#ifndef DCRX_H
#define DCRX_H

#include <stddef.h>
#include <stdint.h>

#include "dcrx_codes.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DCRX_ABI_VERSION 4

enum dcrx_error {
  DCRX_OK = 0,
  DCRX_E_INVALID = -1,      /* bad argument */
  DCRX_E_UNSUPPORTED = -2,  /* tag set outside what the device tables can express (see dcrx_tables_create) */
  DCRX_E_NOMEM = -3,
  DCRX_E_HIP = -4,          /* a HIP runtime call failed (no GPU, launch failure, ...) */
  DCRX_E_NOGPU = -5
};

/* ---- tables: replaces import_tcr_info's module globals (decombine.py:593-746) ---- */

/* What get_v_tags/get_j_tags (decombine.py:820-866) and SeqIO.parse (:683-696)
 * produce for one chain.  Strings are NUL-terminated; regions may be any case
 * (the library upper-cases them like `.seq.upper()`, :695). */
typedef struct dcrx_tagset {
  uint32_t n_v;
  const char *const *v_tags;    /* v_seqs */
  const int32_t *v_jumps;       /* jump_to_end_v */
  const char *const *v_regions; /* v_regions */
  uint32_t n_j;
  const char *const *j_tags;    /* j_seqs */
  const int32_t *j_jumps;       /* jump_to_start_j */
  const char *const *j_regions; /* j_regions */
  int32_t v_half_split;         /* decombine.py:657-661 */
  int32_t j_half_split;
} dcrx_tagset_t;

typedef struct dcrx_tables dcrx_tables_t;

typedef struct dcrx_tables_info {
  uint32_t n_v, n_j;
  uint32_t n_states;        /* states of the merged V/J/half-tag automaton */
  uint32_t dfa_bytes;       /* bytes of the LDS-resident transition table */
  uint32_t n_keywords[6];   /* distinct keywords: V, J, V half1, V half2, J half1, J half2 */
  uint32_t max_tag_len;
  uint32_t tables_in_lds;   /* 1 when the transition table fits the LDS budget */
  uint32_t equal_len_per_automaton; /* 1 when every automaton's keywords share one length (acora tie order then irrelevant) */
  uint32_t pair_scan_bytes;  /* bytes of the two-bases-per-step table of the fast kernel; 0 = not built (> 4095 states,
                                or tags that overlap themselves at shifts 1..4) */
  uint32_t v2_tables;        /* 1 when the v2 kernels serve this tag set (every keyword class of one length, each
                                frame's automaton within 4095 states); else the three-launch form runs */
  uint32_t v2_states[2];     /* states of the forward- / reverse-frame automaton of the v2 scan */
  uint32_t v2_scan_bytes[2]; /* bytes of its 16-bit two-bases-per-step table (what the scan kernel keeps in LDS) */
  uint32_t max_read_len;     /* longest read a batch may hold (DCRX_E_UNSUPPORTED beyond) */
} dcrx_tables_info_t;

/* Compiles the six Aho-Corasick automata of decombine.py:722-746 into one merged
 * DFA plus the per-tag side tables.  Pure host work: needs no GPU.
 * DCRX_E_UNSUPPORTED when a tag is empty, longer than 32 nt, has a character
 * outside ACGT, a half split outside (0, len), a jump outside [-32768, 32767],
 * or the automaton needs more than 16383 states. */
int dcrx_tables_create(const dcrx_tagset_t *tagset, dcrx_tables_t **out);
void dcrx_tables_destroy(dcrx_tables_t *tables);
int dcrx_tables_info(const dcrx_tables_t *tables, dcrx_tables_info_t *info);

/* ---- per-read results: replaces dcr()'s return value (decombine.py:572-581) ---- */

/* 16 bytes per read.  When status == DCRX_S_OK:
 *   [v, j, vdel, jdel, frame_read[ins_start : ins_start+ins_len], v_start, j_end]
 * is exactly dcr()'s 7-list, with frame_read = revcomp(read) when frame == 0 and
 * read itself when frame == 1 (decombine.py:1015-1020).  Otherwise every field
 * but status/frame is 0 and status names the exit path (dcrx_codes.h). */
typedef struct dcrx_record {
  uint16_t v, j;
  uint16_t v_start, j_end;
  uint16_t ins_start, ins_len;
  uint8_t vdel, jdel;
  uint8_t status; /* enum dcrx_status */
  uint8_t frame;  /* 0 reverse, 1 forward */
} dcrx_record_t;

/* inputargs keys read on the path: orientation (:999-1010), allowNs (:554),
 * lenthreshold (:557) */
typedef struct dcrx_cfg {
  int32_t orientation; /* enum dcrx_orientation */
  int32_t allow_ns;
  int32_t lenthreshold;
  uint32_t flags;      /* DCRX_F_* */
} dcrx_cfg_t;

#define DCRX_F_NONE 0u
/* Every other bit of `flags` is a test or profiling switch of the library's own build (declared in the private header
 * decombinator_amd/csrc/dcrx_debug_flags.h, used by this repository's tests and tools).  The switches that leave the
 * records unchanged (A/B launch shapes) are honoured; those that make a kernel stop half-way (records are then NOT
 * results) are refused with DCRX_E_INVALID unless the environment holds DCRX_DEBUG_FLAGS=1.  A caller passes 0. */

/* A batch of reads, 2-bit packed: base i of read r is bits [2(i%4), 2(i%4)+1] of
 * byte packed[r*stride + i/4]; A=0 C=1 G=2 T=3.  Bytes of the FASTQ sequence that
 * are not one of "ACGT" (N, IUPAC codes, lower case) are packed as 0 and listed
 * as exceptions sorted by (read, pos); the device treats them exactly as the
 * reference treats the original byte.  stride is a multiple of
 * 8 with 4*stride >= the longest read and stride <= 16384: reads of up to 65 535 nt — the 16-bit
 * lengths, exception positions and record offsets of this ABI (DCRX_E_UNSUPPORTED beyond;
 * dcrx_tables_info.max_read_len; the reference itself has no limit, decombine.py:228-265).  By stride:
 * <= 40 (150 nt, two reads per lane in registers), <= 80 (320 nt) and <= 128 (511 nt, one read per lane)
 * run on the v2 kernels' register shapes (where those do not apply — dcrx_tables_info.v2_tables == 0 —
 * batches with stride > 80 take a kernel that walks the packed words in memory); a batch with a larger
 * stride takes the long form, one read per lane from memory, far slower per read: a caller puts its
 * reads of 512 nt and more into batches of their own (decombinator_amd/decombine.py does). */
typedef struct dcrx_batch {
  uint64_t n_reads;         /* < 2^32 per call */
  const uint8_t *packed;
  uint32_t stride;
  uint32_t read_len;        /* length of every read when lens == NULL */
  const uint16_t *lens;     /* optional per-read lengths */
  uint64_t n_exc;
  const uint32_t *exc_read; /* read index, ascending */
  const uint16_t *exc_pos;  /* position in the read as stored (FASTQ frame), ascending within a read */
  const uint8_t *exc_chr;   /* the original byte; never one of "ACGT" */
} dcrx_batch_t;

/* ---- host-side packing (the reads come from readfq, decombine.py:228-265) ---- */

/* Packs n_reads ASCII sequences (concatenated in `ascii`, read r at
 * [offsets[r], offsets[r+1])) into `packed` (n_reads*stride bytes) and `lens`.
 * Exceptions go to exc_* (capacity exc_cap).  Returns the number of exceptions
 * found (which may exceed exc_cap: call again with larger buffers), or a
 * negative dcrx_error. */
int64_t dcrx_pack_reads(const char *ascii, const uint64_t *offsets, uint64_t n_reads,
                        uint32_t stride, uint8_t *packed, uint16_t *lens,
                        uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr,
                        uint64_t exc_cap);

/* Same, for reads that are not contiguous: read r is ascii[start[r] .. start[r]+len[r]).
 * This is the form the batch reader below hands out (and the R1 mode's vdj = seq[bclength:],
 * decombine.py:980). */
int64_t dcrx_pack_reads_span(const char *ascii, const uint64_t *start, const uint32_t *len,
                             uint64_t n_reads, uint32_t stride, uint8_t *packed, uint16_t *lens,
                             uint32_t *exc_read, uint16_t *exc_pos, uint8_t *exc_chr,
                             uint64_t exc_cap);

/* Inverse of dcrx_pack_reads: writes lens[r] (or read_len) ASCII bytes of read r
 * to ascii + offsets[r], exception bytes restored. */
int dcrx_unpack_reads(const dcrx_batch_t *host_batch, const uint64_t *offsets, char *ascii);

/* ---- FASTQ / FASTA batch reader (host only) ----
 * Replaces the generator readfq (decombine.py:228-265) over the reference's opener
 * (opener_check, :118-123; text mode = universal newlines), record for record: header up to
 * the first space, multi-line sequences and qualities, FASTA records (no quality), a
 * truncated last record, and the l[:-1] quirk of an unterminated last line.  A batch is a
 * set of offsets into one text buffer owned by the reader, valid until the next call on the
 * same reader. */
typedef struct dcrx_fastq dcrx_fastq_t;

#define DCRX_FASTQ_NO_QUAL 0xFFFFFFFFu /* qual_len of a record the reference yields with qual None */

typedef struct {
  uint64_t n_records;
  const char *text;
  uint64_t text_bytes;
  const uint64_t *name_off; const uint32_t *name_len;
  const uint64_t *seq_off;  const uint32_t *seq_len;
  const uint64_t *qual_off; const uint32_t *qual_len;
} dcrx_fastq_batch_t;

/* gzipped != 0: the file must be gzip (gzip.open); 0: read as it is (open). */
int dcrx_fastq_open(const char *path, int gzipped, dcrx_fastq_t **out);
/* One rank's part of a sharded stage (no counterpart in the single-process reference; its order contract is the read loop's,
 * decombine.py:951-1050): the bytes [begin, end) of a plain four-line FASTQ file, `begin` a record's first byte — the reader
 * touches no other byte of the file, and fails (DCRX_E_INVALID) on anything but plain four-line records. */
int dcrx_fastq_open_range(const char *path, uint64_t begin, uint64_t end, dcrx_fastq_t **out);
/* Newlines among the bytes [begin, end) of a file; nth >= 1: *nth_off = the offset just behind the nth of them (UINT64_MAX when
 * there are fewer; with n_lines == NULL the scan stops there); *has_cr: a carriage return occurs; *file_size.  With it the
 * ranks agree on record boundaries (record k of a four-line file starts at line 4 k) while each reads only its own byte range. */
int dcrx_fastq_lines(const char *path, uint64_t begin, uint64_t end, uint64_t nth, uint64_t *n_lines, uint64_t *nth_off, int *has_cr,
                     uint64_t *file_size);
void dcrx_fastq_close(dcrx_fastq_t *reader);
/* Up to max_records further records; n_records == 0 means the file is exhausted. */
int dcrx_fastq_next(dcrx_fastq_t *reader, uint64_t max_records, dcrx_fastq_batch_t *out);

/* How many of the n spans hold `byte` within their first `prefix` bytes: the reference's
 * `"N" in bc` tally over bc = seq[:bclength] (decombine.py:985-989). */
uint64_t dcrx_count_prefix_byte(const char *text, const uint64_t *start, const uint32_t *len,
                                uint64_t n, uint32_t prefix, int byte);

/* ---- bulk `.n12` row assembly (host only) ----
 * Replaces the row building of the read loop (decombine.py:1012-1039) for a whole batch:
 * for every record with status DCRX_S_OK, in read order, one line
 *   v j vdel jdel insert id inter-tag-seq inter-tag-qual barcode barcode-qual [v_tail]
 * with the string `field_sep` (", " gives the `.n12` text itself, io.py:507-509) between fields
 * and '\n' after the row.  vdj / qual / id / bc / bcq / tail
 * give read r's strings as spans (tail may be NULL: no sampling_analysis).  The reverse frame
 * is revcomp(vdj) and qual[::-1] (:1015-1017); slices clamp like Python's.
 * Returns the number of bytes the rows take (computed without touching the text); they are
 * written only when out != NULL and that fits out_cap (call with out = NULL to size the buffer).  *n_rows = rows.  DCRX_E_UNSUPPORTED when a field
 * itself contains field_sep (or '\n' cannot occur: lines are split there). */
typedef struct {
  const char *text;
  const uint64_t *start;
  const uint32_t *len;
} dcrx_spans_t;

int64_t dcrx_assemble_rows(const dcrx_record_t *records, uint64_t n_reads, const dcrx_spans_t *vdj,
                           const dcrx_spans_t *qual, const dcrx_spans_t *id, const dcrx_spans_t *bc,
                           const dcrx_spans_t *bcq, const dcrx_spans_t *tail, const char *field_sep,
                           char *out, uint64_t out_cap, uint64_t *n_rows);

/* ---- the hot path: replaces the body of the read loop, decombine.py:998-1013 ---- */

/* Host buffers in, host buffers out (H2D copy, kernels, D2H copy, synchronous).
 * records: n_reads entries; counters: DCRX_N_COUNTERS uint64, OVERWRITTEN with
 * this batch's tallies (the caller adds them into its Counter). */
int dcrx_decombine(dcrx_tables_t *tables, const dcrx_cfg_t *cfg, const dcrx_batch_t *host_batch,
                   dcrx_record_t *records, uint64_t *counters);

/* Device buffers in, device buffers out, asynchronous on `hip_stream`
 * (a hipStream_t, NULL = default stream).  Every pointer inside `device_batch`,
 * d_records and d_counters (DCRX_N_COUNTERS uint64, overwritten) are device
 * memory.  No allocation and no synchronisation happens inside once the tables
 * have been used on this device with a batch at least this large
 * (dcrx_reserve_device does that up front) — with ONE exception, and only where the caller has asked for it
 * (dcrx_set_tune_wait, below): the fourth call of a size class of 2^25 reads and more may then wait for the third
 * call's finishing launch, once per class.  Without that call nothing in here ever waits for the device, so the
 * stream may be captured or run arbitrarily far ahead of the host.
 * Calls on one handle are ordered on one stream (the workspace hangs off the handle); batches that should overlap —
 * a batch's finishing launch beside the next batch's scan, or the two chains of a two-chain library — take one handle
 * and one stream each ("Ownership, errors, threading"). */
int dcrx_decombine_device(dcrx_tables_t *tables, const dcrx_cfg_t *cfg,
                          const dcrx_batch_t *device_batch, dcrx_record_t *d_records,
                          uint64_t *d_counters, void *hip_stream);

/* Profiling aid: the following dcrx_decombine_device calls on `tables` record
 * start_event right before and stop_event right after the dominant kernel (the scan),
 * on the stream that kernel is launched on (hipEvent_t handles, e.g. from
 * dcrx_event_create).  NULL, NULL switches it off. */
int dcrx_set_timing_events(dcrx_tables_t *tables, void *start_event, void *stop_event);

/* The same around EVERY launch of a dcrx_decombine_device call (prologue, scan, finishing kernels):
 * the time the whole hot path takes on the device for one batch. */
int dcrx_set_step_events(dcrx_tables_t *tables, void *start_event, void *stop_event);

/* Uploads the tables to the current device and sizes the per-launch workspace
 * for batches of up to max_reads reads.  Footprint (device memory, owned by the
 * handle until dcrx_tables_destroy): the entry lists between the kernels of a
 * call, ~109 bytes per read at stride <= 40 where the scan kernel takes the tail itself (pair tables of up to 64 KB:
 * 1.09 GB for 10 M reads of 150 nt, 10.9 GB for 100 M) and ~163 where the tail is a role of the finishing launch (the tail
 * list is allocated by the first call that needs it), ~310 bytes per read at stride <= 80 (none beyond stride 128: the
 * long form keeps no lists), plus max_reads / 8
 * bytes of exception bitmap, 8 bytes per read of hand-over queues and a few MB
 * of tables.  A later, larger batch grows it (synchronising the device). */
int dcrx_reserve_device(dcrx_tables_t *tables, uint64_t max_reads);

/* Compacts the status==OK records of a batch (the DCR tuples that are gathered
 * across GPUs): d_hits gets the records in input order, d_hit_index their read
 * indices (first_index + position), d_n_hits (one uint64) the count. */
int dcrx_compact_hits_device(const dcrx_record_t *d_records, uint64_t n_reads, uint64_t first_index,
                             dcrx_record_t *d_hits, uint64_t *d_hit_index, uint64_t *d_n_hits,
                             void *hip_stream);

/* Same compaction with the read indices as a bitmap instead of a list: bit (i & 63) of
 * d_ok_bitmap[i >> 6] is set when read i decombined ((n_reads + 63) / 64 words; the k-th record
 * of d_hits belongs to the k-th set bit).  A third fewer bytes to gather than 8-byte indices. */
int dcrx_compact_hits_bitmap_device(const dcrx_record_t *d_records, uint64_t n_reads,
                                    dcrx_record_t *d_hits, uint64_t *d_ok_bitmap, uint64_t *d_n_hits,
                                    void *hip_stream);

/* The same with each decombined record squeezed into 12 bytes (three little-endian uint32; offsets of nine bits: batches of
 * reads of up to 511 nt only):
 *   word 0: v (bits 0-11) | j (12-23) | vdel (24-31)
 *   word 1: v_start (0-8) | j_end (9-17) | ins_start (18-26)
 *   word 2: ins_len (0-8) | jdel (9-16) | frame (17)
 * status is DCRX_S_OK by construction.  Requires < 4096 V and J tags (any real tag set) and reads of up to 511 nt (positions
 * < 512).  d_tuples12: 12 bytes per record. */
int dcrx_compact_hits_packed_device(const dcrx_record_t *d_records, uint64_t n_reads, void *d_tuples12,
                                    uint64_t *d_ok_bitmap, uint64_t *d_n_hits, void *hip_stream);

/* The same in 8 bytes per decombined record (two little-endian uint32), what a sharded run gathers on rank 0:
 *   word 0: v (bits 0-10) | j (11-19) | vdel (20-27) | jdel low 4 bits (28-31)
 *   word 1: jdel high 4 bits (0-3) | v_start (4-12) | j_end (13-21) | ins_len (22-30) | frame (31)
 * ins_start is not sent: it is the base after the end of V (decombine.py:547, :577), v_start + jump_to_end_v[v] - vdel, which
 * the receiver has from the tag file.  Requires < 2048 V tags and < 512 J tags (any real tag set; dcrx_tables_info gives
 * the counts); positions are < 512 by the read-length limit. */
int dcrx_compact_hits_packed8_device(const dcrx_record_t *d_records, uint64_t n_reads, void *d_tuples8,
                                     uint64_t *d_ok_bitmap, uint64_t *d_n_hits, void *hip_stream);

/* The narrow tuple: what a sharded run gathers when the receiver holds the same tag tables.  Field widths come from the
 * tables (dcrx_tuple_layout), fields are packed least significant first:
 *   v | j | vdel | jdel | v_start | j_end | short_end | frame
 * w_v / w_j: bits of n_v - 1 / n_j - 1; w_vdel: bits of max(jump_to_end_v[k] - len(v_seqs[k])) and w_jdel: bits of
 * max(jump_to_start_j[k]) — a decombined read has passed the filter of decombine.py:560-564, so its deletions fit; w_pos
 * (v_start and j_end): bits of max_read_len.  Neither ins_start nor ins_len is sent:
 *   ins_start = v_start + jump_to_end_v[v] - vdel                       (the base after the end of V, :283-285, :547)
 *   ins_len   = j_end - L - jump_to_start_j[j] + jdel - ins_start       (start of J, :407-409, :447, :506-509, :806)
 * with L = len(j_seqs[j]), or 2 * j_half_split when short_end is set: the J half1 rescue sets j_seq_end to the half's
 * start + len(half1) + j_half_split (:450-454), which is not the tag's end when the split is not the tag's middle.
 * bits = the sum (+ 2); bytes = max(4, ceil(bits / 8)) <= 8.  Human beta, original tags, 150 nt: 39 bits, 5 bytes.
 * DCRX_E_UNSUPPORTED when bits > 64 (the caller gathers 12-byte tuples instead).
 * Contract: every read of a batch whose tuples are packed with a layout is at most 2^w_pos - 1 bases long (max_read_len
 * fits by construction).  dcrx_decombine_device refuses a batch that can break it while a tuple sink is set (DCRX_E_INVALID:
 * read_len, or 4 * stride for a batch with lens); dcrx_compact_hits_narrow_device sees records only and trusts its caller. */
typedef struct dcrx_tuple_layout {
  uint8_t w_v, w_j, w_vdel, w_jdel, w_pos;
  uint8_t bits, bytes, reserved;
  uint32_t max_read_len;
} dcrx_tuple_layout_t;
int dcrx_tuple_layout(const dcrx_tables_t *tables, uint32_t max_read_len, dcrx_tuple_layout_t *layout);

/* Bytes of the message of a batch over n_reads read slots with n_hits decombined ones (dcrx_compact_hits_narrow_device):
 *   [ (n_reads + 63) / 64 uint64: bit (i & 63) of word i >> 6 = read i decombined ]
 *   [ n_hits uint32: the tuples' low 32 bits, in read order ]
 *   [ n_hits x (bytes - 4) bytes: their high bytes, least significant first ]
 * The third part starts where the second ends: a sender ships the first dcrx_tuple_message_bytes(...) bytes. */
uint64_t dcrx_tuple_message_bytes(const dcrx_tuple_layout_t *layout, uint64_t n_reads, uint64_t n_hits);

/* Compacts the status==OK records of a batch into that message (d_message: room for n_hits == n_reads); d_n_hits (one
 * uint64) gets the count.  n_slots >= n_reads: the read slots the bitmap spans — a sharded run sizes every step's message
 * for its largest batch, a short last batch leaves the bits beyond its reads zero.  Needs the tables on the current
 * device (the J tags' lengths and jumps decide short_end). */
int dcrx_compact_hits_narrow_device(dcrx_tables_t *tables, const dcrx_tuple_layout_t *layout,
                                    const dcrx_record_t *d_records, uint64_t n_reads, uint64_t n_slots, void *d_message,
                                    uint64_t *d_n_hits, void *hip_stream);

/* The tuple sink of a handle: while one is set, every dcrx_decombine_device call on the handle ALSO leaves the batch's
 * message — exactly what dcrx_compact_hits_narrow_device would make of its records, n_slots >= the batch's reads — in
 * d_message and the count in d_n_hits, on the call's stream.  For the shipped kernels and tuples of up to 40 bits the
 * kernels that write the records leave the tuples behind as they go and one short launch behind them puts the message in
 * read order (no pass over the records: the gather of a sharded run costs the step ~2 %, not 12 %); any other launch
 * shape compacts the records behind the call.  The buffers may change from call to call (a sharded run alternates two
 * messages); layout == NULL turns the sink off.  The handle's workspace grows by ~32 bytes per read of its largest batch. */
int dcrx_set_tuple_sink(dcrx_tables_t *tables, const dcrx_tuple_layout_t *layout, void *d_message, uint64_t n_slots,
                        uint64_t *d_n_hits);

/* ---- the first consumer of the rows: the front half of `collapse` (host, threaded) -------------------------------
 * What read_in_data (src/decombinator/collapse.py:482-565) does to every `.n12` row before it starts grouping rows:
 * get_barcode_positions :367-479 (spacer searches :192-236), set_barcode :278-326, check_umi_quality :343-353 and the
 * inter-tag length filter :553-556, with the reference's counter keys.  `text`: rows as dcrx_assemble_rows writes them
 * (fields separated by cfg->field_sep, one row per line).  Per row a dcrx_collapse_row_t; counters are ADDED to
 * counters[DCRX_CF_N_COUNTERS].  All three spacer searches of the reference are decided here: the spacer verbatim, with up
 * to two substitutions ("{1s<=2}"), and the indel form ("{2i+2d+1s<=2}", :198-201: the spacer with one base inserted or one
 * deleted, the leftmost start first, the insertion where both hold).  DCRX_CF_DEFER (nothing counted) is left for rows
 * that are not ten fields of ASCII text: the caller's own error path.  rows == NULL: returns the number of rows.
 * row_offsets (optional, n + 1 entries): where each row starts in `text`.  Returns the number of rows or an error. */
enum dcrx_collapse_status { DCRX_CF_OK = 0, DCRX_CF_NO_BCLOCS = 1, DCRX_CF_LOW_QUALITY = 2, DCRX_CF_OVERLONG = 3, DCRX_CF_DEFER = 255 };
enum dcrx_collapse_counter {
  DCRX_CF_C_INPUT_DCRS = 0,        /* readdata_input_dcrs */
  DCRX_CF_C_FAIL_N = 1,            /* getbarcode_fail_N */
  DCRX_CF_C_FAIL_NOSPACER = 2,     /* getbarcode_fail_nospacerfound */
  DCRX_CF_C_FAIL_NOT2SPACERS = 3,  /* getbarcode_fail_not2spacersfound */
  DCRX_CF_C_FAIL_N1SHORT = 4,      /* getbarcode_fail_n1tooshort */
  DCRX_CF_C_FAIL_N1LONG = 5,       /* getbarcode_fail_n1toolong */
  DCRX_CF_C_FAIL_N2PASTEND = 6,    /* getbarcode_fail_n2pastend */
  DCRX_CF_C_PASS_EXACT = 7,        /* getbarcode_pass_exactmatch */
  DCRX_CF_C_PASS_REGEX = 8,        /* getbarcode_pass_regexmatch */
  DCRX_CF_C_PASS_FUZZY_RIGHTLEN = 9, /* getbarcode_pass_fuzzymatch_rightlen */
  DCRX_CF_C_PASS_FUZZY_SHORT = 10, /* getbarcode_pass_fuzzymatch_short */
  DCRX_CF_C_PASS_FUZZY_LONG = 11,  /* getbarcode_pass_fuzzymatch_long */
  DCRX_CF_C_PASS_OTHER = 12,       /* getbarcode_pass_other */
  DCRX_CF_C_FAIL_NO_BCLOCS = 13,   /* readdata_fail_no_bclocs */
  DCRX_CF_C_SHORT_BARCODE = 14,    /* readdata_short_barcode */
  DCRX_CF_C_LONG_BARCODE = 15,     /* readdata_long_barcode */
  DCRX_CF_C_FAIL_LOW_QUALITY = 16, /* readdata_fail_low_barcode_quality */
  DCRX_CF_C_FAIL_OVERLONG = 17,    /* readdata_fail_overlong_intertag_seq */
  DCRX_CF_C_SUCCESS = 18,          /* readdata_success */
  DCRX_CF_N_COUNTERS = 19
};
typedef struct dcrx_collapse_cfg {
  int32_t oligo;            /* 0 m13, 1 i8, 2 i8_single, 3 nebio, 4 takara (collapse.py:174-189) */
  int32_t allow_ns;         /* inputargs["allowNs"] (:390) */
  int32_t lenthreshold;     /* inputargs["lenthreshold"] (:553) */
  double min_bc_q, bc_q_below_min, avg_q_threshold; /* barcode_quality_parameters (:343-353) */
  char field_sep[8];        /* NUL-terminated; ", " for `.n12` text */
} dcrx_collapse_cfg_t;
typedef struct dcrx_collapse_row {
  int16_t b1start, b1end, b2start, b2end; /* bc_locs (b2start = b2end = -1 for nebio / takara), -1 when none */
  uint8_t status;                          /* enum dcrx_collapse_status */
  uint8_t barcode_len, barcode_qual_len;
  uint8_t pad;
  char barcode[24], barcode_qual[24];      /* set_barcode's two strings */
} dcrx_collapse_row_t;
/* spacerSearch (collapse.py:204-212) on its own: every non-overlapping match of `spacer` in seq[0, n), left to right, found by
 * the first of the three searches that finds any — regex.findall(spacer, seq), "(spacer){1s<=2}", "(spacer){2i+2d+1s<=2}".
 * starts[k] / lens[k] (up to `cap` of them) receive the matches; *kind: 0 verbatim, 1 substitutions, 2 indel.  Returns the
 * number of matches (it may exceed cap) or an error. */
int32_t dcrx_spacer_search(const char *seq, int32_t n, const char *spacer, int32_t m, int32_t *starts, int32_t *lens, int32_t cap,
                           int32_t *kind);
int64_t dcrx_collapse_front(const char *text, uint64_t n_bytes, const dcrx_collapse_cfg_t *cfg, dcrx_collapse_row_t *rows,
                            uint64_t rows_cap, uint64_t *row_offsets, uint64_t *counters, int n_threads);

/* ---- the intermediate files' gzip step (reference io.py:497-506: the text re-read and written through gzip.open) ----
 * A multi-member gzip file whose pieces (4 MB of text each) are deflated by n_threads threads (0: the host's); any gzip
 * reader sees one stream with exactly the bytes written.  level 1..9 (the reference's gzip.open uses 9).
 * Errors (a path that cannot be opened, a short write) are DCRX_E_INVALID with the text in dcrx_last_error(). */
int dcrx_gzip_open(const char *path, int level, int n_threads, void **writer);
int dcrx_gzip_write(void *writer, const void *data, uint64_t n_bytes);
int dcrx_gzip_close(void *writer);

/* ---- translate.get_cdr3 for a batch of DCRs (host only) ----
 * Replaces the per-row body of the reference's CDR3 step (src/decombinator/translate.py:257-357, called per unique DCR from
 * cdr3translator :388-533): from the gene tables of import_gene_information (:163-254) and the five fields of each DCR,
 *   sequence = V region without its last vdel bases + insert + J region from base jdel on (:296-305), its translation
 *   (standard table; codons with IUPAC ambiguity codes as Bio.Seq.translate resolves them), the in-frame, stop-codon,
 *   conserved-C and conserved-F calls (:312-347) and, for productive rows, where junction_aa / junction lie (:350-355).
 * Every index and slice behaves as Python's.  Strings travel as one text and offsets: gene k's region is
 * v_regions[v_region_off[k] .. v_region_off[k + 1]) (upper case, as the reference stores them), likewise the V genes'
 * conserved residues (v_translate_residue: compared as a whole string with ONE residue of the translation, :327) and the J
 * genes' motifs (j_translate_residue: searched with re.findall in four residues, :341-343 — literal characters, '.',
 * character classes and escaped literals are served here; a motif with any other regular-expression syntax is parsed only when a
 * row uses its gene, as the reference compiles only the motif it searches with, and such a row comes back DCRX_CDR3_MOTIF_LEFT:
 * text, in_frame, stop, conserved_c, start_cdr3 are set, productive holds the calls so far, junction_aa_off / junction_aa_len
 * say which residues of sequence_aa the search looks at, and the caller finishes the row with its own engine — conserved_f,
 * end_cdr3 = len(sequence_aa[start_cdr3:]) + j_pos + start_cdr3 + 1 on a match, the junctions; decombinator_amd/translate.py does).
 * The DCRs: v / j / vdel / jdel as integers (int(dcr[k]), :283-286), the insert of row r = ins[ins_off[r] .. ins_off[r + 1])
 * (the caller has stripped the blank the `translate` command's rows carry, :287-290).
 * Returns the bytes the rows' text takes — per row its sequence, then its sequence_aa — written into `text` when that fits
 * text_cap (call with text = NULL to size the buffer); rows[r] says where they lie.  A row the reference would raise on keeps
 * status != 0 and nothing else: DCRX_CDR3_INDEX_ERROR (a gene index outside its table, or a translation shorter than the V
 * gene's residue position: IndexError), DCRX_CDR3_BAD_CODON (a letter that is no nucleotide code: Biopython's "Codon '...'
 * is invalid", bad_codon_at = the codon's first base; a codon of three gaps '---' is the gap '-', as Seq.translate has it). */
enum dcrx_cdr3_status { DCRX_CDR3_OK = 0, DCRX_CDR3_INDEX_ERROR = 1, DCRX_CDR3_BAD_CODON = 2, DCRX_CDR3_MOTIF_LEFT = 3 };
typedef struct dcrx_cdr3_genes {
  uint32_t n_v, n_j;
  const char *v_regions; const uint64_t *v_region_off;      /* n_v + 1 offsets */
  const char *j_regions; const uint64_t *j_region_off;      /* n_j + 1 */
  const int32_t *v_pos; const char *v_res; const uint32_t *v_res_off;        /* v_translate_position, v_translate_residue */
  const int32_t *j_pos; const char *j_motif; const uint32_t *j_motif_off;    /* j_translate_position, j_translate_residue */
} dcrx_cdr3_genes_t;
typedef struct dcrx_cdr3_row {
  uint64_t seq_off, aa_off;              /* into the text: sequence, sequence_aa */
  uint32_t seq_len, aa_len;
  uint32_t junction_off, junction_len;          /* inside the row's sequence (productive rows; else 0, 0) */
  uint32_t junction_aa_off, junction_aa_len;    /* inside the row's sequence_aa */
  int32_t start_cdr3, end_cdr3;          /* as the reference computes them (residues) */
  uint32_t bad_codon_at;
  uint8_t status;                        /* enum dcrx_cdr3_status */
  uint8_t productive, in_frame, stop, conserved_c, conserved_f;      /* 1 = "T" */
  uint8_t pad[2];
  uint32_t reserved;                     /* (64 bytes) */
} dcrx_cdr3_row_t;
int64_t dcrx_cdr3_batch(const dcrx_cdr3_genes_t *genes, uint64_t n, const int32_t *v, const int32_t *j, const int32_t *vdel,
                        const int32_t *jdel, const char *ins, const uint64_t *ins_off, dcrx_cdr3_row_t *rows, char *text, uint64_t text_cap);

/* What a handle has settled for its own launches (no counterpart in the reference).  Where the scan kernel takes the tail
 * itself, a handle times the finishing launches of its first calls of a batch-size class (batches of 2^k .. 2^(k+1) - 1 reads,
 * k >= 20) on two settings and keeps the faster for the class: 4096 or 3072 rescue waves for batches below 2^25 reads, 8192 or
 * 4096 from there (`candidates`: the first setting | the second << 16).  rescue_waves = the choice once settled, 0 before (a
 * launch then runs on the first setting or is one of the two samples); launches = calls seen in the class; us_first / us_second =
 * what the samples took (0 before).  The samples are read without waiting, so a caller that queues launches ahead of the device
 * keeps the first setting until they are complete — except, where the caller has allowed it with dcrx_set_tune_wait(tables, 1),
 * for batches of 2^25 reads and more: the fourth call of such a class then waits for the third's finishing launch, once
 * (milliseconds; such batches gain up to 6 % from the choice).  The default is never to wait.
 * orientation: DCRX_ORIENT_REVERSE or DCRX_ORIENT_FORWARD, as in dcrx_cfg_t (the frame whose launches are meant).
 * DCRX_E_INVALID for a null argument.
 * Round 6: where the scan kernel takes the tail, a handle also settles per size class whether list E's entries (one gene to rescue) are
 * finished inside the scan kernel as well (launch_form 4) or by the finishing launch (3): where list E held at most a quarter of the
 * reads of the class's first launch, three launches of each form are timed (events on the dispatches, read without waiting) from
 * the class's ninth launch on, and the faster form stays. */
typedef struct dcrx_tune_state {
  uint32_t rescue_waves, launches;
  float us_first, us_second;
  uint32_t launch_form;      /* the frame's last call, whatever its size: 0 none yet, 1 the three-launch form (tag sets or shapes the
                                v2 kernels do not serve), 2 the v2 kernels with the tail a role of the finishing launch, 3 the v2
                                kernels with the tail inside the scan kernel, 4 the same with list E's entries finished inside the scan kernel as
                                well (chosen per size class from the share of list-E entries in the class's first launch) */
  uint32_t candidates;
} dcrx_tune_state_t;
int dcrx_tune_state(const dcrx_tables_t *tables, int orientation, uint64_t n_reads, dcrx_tune_state_t *out);
/* allow != 0: dcrx_decombine_device may wait once per big-batch size class as described above (0, the default: never). */
int dcrx_set_tune_wait(dcrx_tables_t *tables, int allow);

/* The persistent kernels of dcrx_decombine_device normally fill every compute unit; n_cus of
 * them are left free from the next call on (for a collective running on another stream). */
int dcrx_set_reserved_cus(dcrx_tables_t *tables, uint32_t n_cus);

/* ---- multi-GPU: RCCL over xGMI, bound directly (SURVEY.md 8(b) dcrx_decombine_sharded, 8(e)) -----------------------
 * The reference has no counterpart (one process, one Counter: decombine.py:598, README.md:370-376 "submit many jobs").
 * Reads are sharded over the ranks in contiguous ranges (rank r of W owns reads [r N / W, (r + 1) N / W): concatenating the
 * ranks' outputs in rank order is the reference's input order, outdata.append :1039), every rank runs the hot path on its own
 * GPU, and ONE exchange follows: the ranks' counts of decombined reads (all-gather), their tuple messages — the
 * dcrx_set_tuple_sink message: a bitmap of the shard's reads and the narrow tuples of the decombined ones — to rank 0 in
 * exact sizes (grouped send / receive), the uint64[DCRX_N_COUNTERS] summed over the ranks (all-reduce).
 * librccl is opened on the first call of this section (by soname: a process that already holds an RCCL shares it); a caller
 * that never comes here never loads it.  DCRX_E_UNSUPPORTED when it cannot be opened.
 *
 * A communicator: one process per GPU — rank 0 draws an id (dcrx_comm_unique_id), carries its DCRX_COMM_ID_BYTES bytes to the
 * other ranks by whatever means it has (a file, a socket, an environment variable of the launcher), and every rank calls
 * dcrx_comm_create on its own device (dcrx_set_device first) — or one process for all GPUs: dcrx_comm_create_all fills
 * out[0 .. world) (ncclCommInitAll; devices == NULL: 0 .. world - 1). */
#define DCRX_COMM_ID_BYTES 128
enum dcrx_comm_op { DCRX_COMM_SUM = 0, DCRX_COMM_MAX = 1 };
typedef struct dcrx_comm dcrx_comm_t;
int dcrx_comm_available(void);                        /* 1 when librccl could be opened */
int dcrx_comm_unique_id(uint8_t *id /* DCRX_COMM_ID_BYTES */);
int dcrx_comm_create(const uint8_t *id, int world, int rank, dcrx_comm_t **out);
int dcrx_comm_create_all(int world, const int *devices, dcrx_comm_t **out /* [world] */);
void dcrx_comm_destroy(dcrx_comm_t *comm);
int dcrx_comm_info(const dcrx_comm_t *comm, int *world, int *rank, int *device);
/* The pieces, asynchronous on `hip_stream`, device memory throughout — for a caller that pipelines steps (bench.py:
 * the transfers of step k beside the scan of step k + 1).  gather_v: every rank but `root` sends send_bytes from d_send;
 * the root receives recv_bytes[r] into d_recv[r] from every r != root, all in one group (its own message stays where it
 * is; d_recv and recv_bytes are read on the root only). */
int dcrx_comm_allreduce_u64(dcrx_comm_t *comm, const uint64_t *d_in, uint64_t *d_out, uint64_t n, int op, void *hip_stream);
int dcrx_comm_allreduce_f64(dcrx_comm_t *comm, const double *d_in, double *d_out, uint64_t n, int op, void *hip_stream);
int dcrx_comm_allgather(dcrx_comm_t *comm, const void *d_in, void *d_out /* world x bytes_per_rank */, uint64_t bytes_per_rank, void *hip_stream);
int dcrx_comm_gather_v(dcrx_comm_t *comm, const void *d_send, uint64_t send_bytes, void *const *d_recv, const uint64_t *recv_bytes,
                       int root, void *hip_stream);
int dcrx_comm_barrier(dcrx_comm_t *comm, void *hip_stream);      /* an all-reduce of one word, then the stream is waited for */
/* Host memory in and out, synchronous (staged through device memory): sizes, error flags, a Counter's keys. */
int dcrx_comm_allgather_host(dcrx_comm_t *comm, const void *h_in, void *h_out /* world x bytes_per_rank */, uint64_t bytes_per_rank);
int dcrx_comm_allreduce_host_u64(dcrx_comm_t *comm, uint64_t *h_inout, uint64_t n, int op);
/* One step of a sharded run in one call, on every rank: dcrx_decombine_device on this rank's shard (device memory, as
 * there) with the tuple sink on d_message (room for dcrx_tuple_message_bytes(layout, n_slots, n_slots); n_slots >= the
 * shard's reads); then the exchange above.  On return the stream holds — not yet waited for — the transfers and the
 * all-reduce: d_counters will be the JOB's counters on every rank, and on rank 0 d_gathered[r] rank r's message for r > 0
 * (d_gathered[0] is not touched: rank 0's own message is d_message; d_gathered is read on rank 0 only).  n_hits_by_rank
 * (host, [world]) is filled before the call returns: the one wait inside (the counts decide the transfers' sizes).
 * Unpacking a message: bitmap of n_slots bits, then the tuples in read order (dcrx_tuple_layout_t above). */
int dcrx_decombine_sharded(dcrx_tables_t *tables, dcrx_comm_t *comm, const dcrx_cfg_t *cfg, const dcrx_batch_t *device_shard,
                           dcrx_record_t *d_records, uint64_t *d_counters, const dcrx_tuple_layout_t *layout, void *d_message,
                           uint64_t n_slots, void *const *d_gathered, uint64_t *n_hits_by_rank, void *hip_stream);

/* ---- device plumbing for callers without a HIP binding of their own ---- */
int dcrx_device_count(void);
int dcrx_set_device(int device);
int dcrx_device_name(char *buf, size_t cap);
int dcrx_malloc_device(void **ptr, size_t bytes);
int dcrx_free_device(void *ptr);
/* Page-locked host memory: dcrx_decombine copies straight from a `packed` buffer and into a `records` buffer that live in
 * such memory (its own or any other the HIP runtime has pinned), without the staging copies pageable buffers need. */
int dcrx_malloc_host(void **ptr, size_t bytes);
int dcrx_free_host(void *ptr);
int dcrx_memcpy_h2d(void *dst_device, const void *src_host, size_t bytes);
int dcrx_memcpy_d2h(void *dst_host, const void *src_device, size_t bytes);
int dcrx_memset_device(void *dst_device, int value, size_t bytes);
int dcrx_synchronize(void);
/* HIP events on a stream, for timing the launches above where they run */
int dcrx_event_create(void **event);
int dcrx_event_destroy(void *event);
int dcrx_event_record(void *event, void *hip_stream);
int dcrx_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
int dcrx_event_synchronize(void *event);
int dcrx_event_create_ordering(void **event);      /* an event without timestamps (hipEventDisableTiming): for ordering streams, cheaper to record */
/* streams of the caller's own, and copies on them (a sharded caller's side stream for the exchange) */
int dcrx_stream_create(void **hip_stream);
int dcrx_stream_destroy(void *hip_stream);
int dcrx_stream_synchronize(void *hip_stream);
int dcrx_stream_wait_event(void *hip_stream, void *event);
int dcrx_memcpy_d2h_async(void *dst_host, const void *src_device, size_t bytes, void *hip_stream);
int dcrx_memcpy_d2d_async(void *dst_device, const void *src_device, size_t bytes, void *hip_stream);
int dcrx_memset_device_async(void *dst_device, int value, size_t bytes, void *hip_stream);

int dcrx_abi_version(void);
const char *dcrx_last_error(void);
/* 1 when the HIP kernels are compiled in (always, in this library), and the
 * gfx target they were built for */
const char *dcrx_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* DCRX_H */
