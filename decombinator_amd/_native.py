"""ctypes binding of libdcrx.so (include/dcrx.h, include/dcrx_synth.h).

The HIP library IS the product: there is no Python or CPU implementation of the
hot path behind this module.  Importing it fails loudly when the shared object
has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C decombinator_amd/csrc`), and every decombine call raises DcrxError
when no GPU is present.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCRX_LIB_PATH: developer override for A/B builds of the same library (never a different backend)
LIB_PATH = os.environ.get("DCRX_LIB_PATH") or os.path.join(_HERE, "csrc", "libdcrx.so")

N_COUNTERS = 32
ABI_VERSION = 4

# enum dcrx_counter order; the strings are the reference's Counter keys
# (reference decombine.py:598 and the increments cited in include/dcrx_codes.h)
COUNTER_NAMES = [
    "multiple_v_matches", "verr2", "foundv1notv2", "verr1", "foundv2notv1",
    "no_vtags_found", "multiple_j_matches", "jerr2", "foundj1notj2", "jerr1",
    "no_j_assigned", "dcrfilter_intertagN", "dcrfilter_toolong_intertag",
    "dcrfilter_imposs_deletion", "dcrfilter_tag_overlap", "VJ_assignment_failed",
    "v_del_failed_tag_at_end", "v_del_failed", "j_del_failed", "vj_count",
    "read_count", "foundj2notj1", "frame_forward",
]

DEVICE_ERRORS = 31      # DCRX_C_DEVICE_ERRORS (not a reference counter)

STATUS_NAMES = [
    "OK", "V_MULTI", "V_WALK_FAIL_AT_END", "V_WALK_FAIL", "V_HALF1_EXHAUSTED", "V_HALF2_EXHAUSTED",
    "V_NONE", "J_MULTI", "J_WALK_FAIL", "J_HALF1_EXHAUSTED", "J_HALF2_EXHAUSTED", "J_NONE",
    "F_INTERTAG_N", "F_TOOLONG", "F_IMPOSS_DEL", "F_OVERLAP",
]

ORIENTATIONS = {"reverse": 0, "forward": 1, "both": 2}
F_FORCE_SLOW_READER = 1
F_PROFILE_SCAN_ONLY = 2
F_ONE_BASE_SCAN = 4
F_V1_KERNELS = 64        # the three-launch form even where the v2 kernel applies
F_V2_NO_LEAN_RESCUE = 8192   # A/B and tests: event entries go to the general form at once (no lean rescue kernel)
F_V2_LEAN_SERIAL = 32768     # A/B: the finishing roles as launches of their own, one after the other (default: roles of one launch)
F_V2_NO_FUSE = 131072       # A/B: no tail inside the scan kernel
F_V2_SIDE_STREAMS = 65536    # A/B: round 3's shape, the tail kernel and the X pass on side streams beside the rescue kernel


def F_V2_SHAPE(k):
    """v2 kernel launch shape (A/B): 1 = one read per lane, two blocks per CU; 2 = two reads per lane; 3 = one read, one block."""
    return int(k) << 8

F_PROFILE_LIST_SCAN_ONLY = 8
F_LIST_RESCUE = 16      # rescue queue through the list kernel even when the pair form applies

RECORD_DTYPE = np.dtype([
    ("v", "<u2"), ("j", "<u2"), ("v_start", "<u2"), ("j_end", "<u2"),
    ("ins_start", "<u2"), ("ins_len", "<u2"), ("vdel", "u1"), ("jdel", "u1"),
    ("status", "u1"), ("frame", "u1"),
])
assert RECORD_DTYPE.itemsize == 16


class DcrxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dcrx error {code}: {msg}")
        self.code = code


class TagSetC(C.Structure):
    _fields_ = [
        ("n_v", C.c_uint32), ("v_tags", C.POINTER(C.c_char_p)), ("v_jumps", C.POINTER(C.c_int32)),
        ("v_regions", C.POINTER(C.c_char_p)),
        ("n_j", C.c_uint32), ("j_tags", C.POINTER(C.c_char_p)), ("j_jumps", C.POINTER(C.c_int32)),
        ("j_regions", C.POINTER(C.c_char_p)),
        ("v_half_split", C.c_int32), ("j_half_split", C.c_int32),
    ]


class TablesInfoC(C.Structure):
    _fields_ = [
        ("n_v", C.c_uint32), ("n_j", C.c_uint32), ("n_states", C.c_uint32), ("dfa_bytes", C.c_uint32),
        ("n_keywords", C.c_uint32 * 6), ("max_tag_len", C.c_uint32), ("tables_in_lds", C.c_uint32),
        ("equal_len_per_automaton", C.c_uint32), ("pair_scan_bytes", C.c_uint32),
        ("v2_tables", C.c_uint32), ("v2_states", C.c_uint32 * 2), ("v2_scan_bytes", C.c_uint32 * 2), ("max_read_len", C.c_uint32),
    ]


class TupleLayoutC(C.Structure):
    _fields_ = [("w_v", C.c_uint8), ("w_j", C.c_uint8), ("w_vdel", C.c_uint8), ("w_jdel", C.c_uint8), ("w_pos", C.c_uint8),
                ("bits", C.c_uint8), ("bytes", C.c_uint8), ("reserved", C.c_uint8), ("max_read_len", C.c_uint32)]


class CfgC(C.Structure):
    _fields_ = [("orientation", C.c_int32), ("allow_ns", C.c_int32), ("lenthreshold", C.c_int32),
                ("flags", C.c_uint32)]


class BatchC(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("packed", C.c_void_p), ("stride", C.c_uint32), ("read_len", C.c_uint32),
        ("lens", C.c_void_p), ("n_exc", C.c_uint64), ("exc_read", C.c_void_p), ("exc_pos", C.c_void_p),
        ("exc_chr", C.c_void_p),
    ]


class FastqBatchC(C.Structure):
    _fields_ = [
        ("n_records", C.c_uint64), ("text", C.c_void_p), ("text_bytes", C.c_uint64),
        ("name_off", C.c_void_p), ("name_len", C.c_void_p), ("seq_off", C.c_void_p), ("seq_len", C.c_void_p),
        ("qual_off", C.c_void_p), ("qual_len", C.c_void_p),
    ]


class SpansC(C.Structure):
    _fields_ = [("text", C.c_void_p), ("start", C.c_void_p), ("len", C.c_void_p)]


_collapse_tls = threading.local()      # per thread: the row buffer of its last collapse_front call


class CollapseCfgC(C.Structure):
    _fields_ = [("oligo", C.c_int32), ("allow_ns", C.c_int32), ("lenthreshold", C.c_int32), ("min_bc_q", C.c_double),
                ("bc_q_below_min", C.c_double), ("avg_q_threshold", C.c_double), ("field_sep", C.c_char * 8)]


COLLAPSE_ROW_DTYPE = np.dtype([("b1start", "<i2"), ("b1end", "<i2"), ("b2start", "<i2"), ("b2end", "<i2"), ("status", "u1"),
                               ("barcode_len", "u1"), ("barcode_qual_len", "u1"), ("pad", "u1"), ("barcode", "S24"), ("barcode_qual", "S24")])
assert COLLAPSE_ROW_DTYPE.itemsize == 60
COLLAPSE_OLIGOS = {"m13": 0, "i8": 1, "i8_single": 2, "nebio": 3, "takara": 4}
COLLAPSE_COUNTERS = [
    "readdata_input_dcrs", "getbarcode_fail_N", "getbarcode_fail_nospacerfound", "getbarcode_fail_not2spacersfound",
    "getbarcode_fail_n1tooshort", "getbarcode_fail_n1toolong", "getbarcode_fail_n2pastend", "getbarcode_pass_exactmatch",
    "getbarcode_pass_regexmatch", "getbarcode_pass_fuzzymatch_rightlen", "getbarcode_pass_fuzzymatch_short",
    "getbarcode_pass_fuzzymatch_long", "getbarcode_pass_other", "readdata_fail_no_bclocs", "readdata_short_barcode",
    "readdata_long_barcode", "readdata_fail_low_barcode_quality", "readdata_fail_overlong_intertag_seq", "readdata_success",
]
CF_OK, CF_NO_BCLOCS, CF_LOW_QUALITY, CF_OVERLONG, CF_DEFER = 0, 1, 2, 3, 255


def spacer_search(seq: str, spacer: str):
    """dcrx_spacer_search: ([(start, length), ...] of every match, kind) — kind 0 verbatim, 1 substitutions, 2 indel."""
    sb, pb = seq.encode("latin-1", "replace"), spacer.encode("latin-1", "replace")
    cap = len(sb) + 1
    starts = np.zeros(cap, dtype=np.int32)
    lens = np.zeros(cap, dtype=np.int32)
    kind = C.c_int32(0)
    n = check(lib().dcrx_spacer_search(sb, len(sb), pb, len(pb), starts.ctypes.data, lens.ctypes.data, cap, C.byref(kind)))
    return [(int(starts[k]), int(lens[k])) for k in range(min(n, cap))], int(kind.value)


def collapse_front(text: bytes, oligo: str, allow_ns: bool, lenthreshold: int, quality_parameters, field_sep: str = ", ", n_threads: int = 0):
    """dcrx_collapse_front over `.n12` text: (rows as COLLAPSE_ROW_DTYPE, row offsets (n + 1), counters uint64[len(COLLAPSE_COUNTERS)])."""
    cfg = CollapseCfgC()
    cfg.oligo = COLLAPSE_OLIGOS[oligo.lower()]
    cfg.allow_ns, cfg.lenthreshold = int(bool(allow_ns)), int(lenthreshold)
    cfg.min_bc_q, cfg.bc_q_below_min, cfg.avg_q_threshold = (float(x) for x in quality_parameters)
    cfg.field_sep = field_sep.encode("ascii")
    buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, dtype=np.uint8)
    scratch = np.zeros(len(COLLAPSE_COUNTERS), dtype=np.uint64)
    n = check(lib().dcrx_collapse_front(buf.ctypes.data, len(text), C.byref(cfg), None, 0, None, scratch.ctypes.data, 0))      # sizing call
    # (the per-row records of a call are written once and the pages behind a fresh allocation cost more than the rows themselves:
    # the buffer of the last call is kept and handed out again when the caller has let go of it)
    # (per thread: two threads never share a buffer, and a thread hands its own out again only when nothing else refers to it)
    held = getattr(_collapse_tls, "buf", None)
    if held is None or len(held) < n or sys.getrefcount(held) > 3:
        held = np.empty(max(n, 1), dtype=COLLAPSE_ROW_DTYPE)
        _collapse_tls.buf = held
    rows = held
    del held
    offs = np.empty(n + 1, dtype=np.uint64)
    cnt = np.zeros(len(COLLAPSE_COUNTERS), dtype=np.uint64)
    check(lib().dcrx_collapse_front(buf.ctypes.data, len(text), C.byref(cfg), rows.ctypes.data, n, offs.ctypes.data, cnt.ctypes.data, int(n_threads)))
    return rows[:n], offs, cnt


class Cdr3GenesC(C.Structure):
    _fields_ = [("n_v", C.c_uint32), ("n_j", C.c_uint32),
                ("v_regions", C.c_void_p), ("v_region_off", C.c_void_p), ("j_regions", C.c_void_p), ("j_region_off", C.c_void_p),
                ("v_pos", C.c_void_p), ("v_res", C.c_void_p), ("v_res_off", C.c_void_p),
                ("j_pos", C.c_void_p), ("j_motif", C.c_void_p), ("j_motif_off", C.c_void_p)]


CDR3_ROW_DTYPE = np.dtype([("seq_off", "<u8"), ("aa_off", "<u8"), ("seq_len", "<u4"), ("aa_len", "<u4"), ("junction_off", "<u4"),
                           ("junction_len", "<u4"), ("junction_aa_off", "<u4"), ("junction_aa_len", "<u4"), ("start_cdr3", "<i4"),
                           ("end_cdr3", "<i4"), ("bad_codon_at", "<u4"), ("status", "u1"), ("productive", "u1"), ("in_frame", "u1"),
                           ("stop", "u1"), ("conserved_c", "u1"), ("conserved_f", "u1"), ("pad", "u1", (2,)), ("reserved", "<u4")])
CDR3_OK, CDR3_INDEX_ERROR, CDR3_BAD_CODON, CDR3_MOTIF_LEFT = 0, 1, 2, 3


class Cdr3Genes:
    """The gene tables of dcrx_cdr3_batch (the reference's import_gene_information, translate.py:163-254) as texts and offsets."""

    def __init__(self, v_regions, j_regions, v_pos, v_res, j_pos, j_motif):
        def blob(strings, off_dtype):
            bs = [str(x).encode("latin-1") for x in strings]
            off = np.zeros(len(bs) + 1, dtype=off_dtype)
            off[1:] = np.cumsum([len(b) for b in bs])
            return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy(), off
        self.n_v, self.n_j = len(v_regions), len(j_regions)
        self._keep = [blob(v_regions, np.uint64), blob(j_regions, np.uint64), blob(v_res, np.uint32), blob(j_motif, np.uint32),
                      np.ascontiguousarray(v_pos, dtype=np.int32), np.ascontiguousarray(j_pos, dtype=np.int32)]
        (vr, vo), (jr, jo), (vs, vso), (jm, jmo), vp_, jp_ = self._keep
        self.c = Cdr3GenesC(self.n_v, self.n_j, vr.ctypes.data, vo.ctypes.data, jr.ctypes.data, jo.ctypes.data,
                            vp_.ctypes.data, vs.ctypes.data, vso.ctypes.data, jp_.ctypes.data, jm.ctypes.data, jmo.ctypes.data)


def cdr3_batch(genes: Cdr3Genes, v, j, vdel, jdel, inserts):
    """dcrx_cdr3_batch: (rows[CDR3_ROW_DTYPE], text bytes) for the DCRs (v, j, vdel, jdel: integers; inserts: strings)."""
    n = len(v)
    v, j, vdel, jdel = (np.ascontiguousarray(x, dtype=np.int32) for x in (v, j, vdel, jdel))
    ib = [str(x).encode("latin-1") for x in inserts]
    ioff = np.zeros(n + 1, dtype=np.uint64)
    ioff[1:] = np.cumsum([len(b) for b in ib])
    itext = np.frombuffer(b"".join(ib) + b"\0", dtype=np.uint8).copy()
    rows = np.zeros(n, dtype=CDR3_ROW_DTYPE)
    assert CDR3_ROW_DTYPE.itemsize == 64
    args = (C.byref(genes.c), n, v.ctypes.data, j.ctypes.data, vdel.ctypes.data, jdel.ctypes.data, itext.ctypes.data, ioff.ctypes.data, rows.ctypes.data)
    need = int(lib().dcrx_cdr3_batch(*args, None, 0))
    if need < 0:
        check(need)
    text = np.zeros(max(need, 1), dtype=np.uint8)
    got = int(lib().dcrx_cdr3_batch(*args, text.ctypes.data, need))
    if got < 0:
        check(got)
    return rows, text.tobytes()


class TuneStateC(C.Structure):
    _fields_ = [("rescue_waves", C.c_uint32), ("launches", C.c_uint32), ("us_first", C.c_float), ("us_second", C.c_float),
                ("launch_form", C.c_uint32), ("candidates", C.c_uint32)]


class SynthCfgC(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("read_len", C.c_uint32), ("p_rearranged", C.c_float),
                ("sub_rate", C.c_float), ("n_rate", C.c_float)]


# every symbol include/dcrx.h and include/dcrx_synth.h declare
EXPORTS = [
    "dcrx_tables_create", "dcrx_tables_destroy", "dcrx_tables_info", "dcrx_pack_reads", "dcrx_pack_reads_span",
    "dcrx_unpack_reads", "dcrx_fastq_open", "dcrx_fastq_open_range", "dcrx_fastq_lines", "dcrx_fastq_close", "dcrx_fastq_next", "dcrx_count_prefix_byte", "dcrx_assemble_rows",
    "dcrx_decombine", "dcrx_decombine_device", "dcrx_set_timing_events", "dcrx_set_step_events", "dcrx_reserve_device", "dcrx_compact_hits_device",
    "dcrx_compact_hits_bitmap_device", "dcrx_compact_hits_packed_device", "dcrx_set_reserved_cus", "dcrx_tune_state", "dcrx_cdr3_batch",
    "dcrx_device_count", "dcrx_set_device", "dcrx_device_name", "dcrx_malloc_device", "dcrx_free_device",
    "dcrx_malloc_host", "dcrx_free_host", "dcrx_memcpy_h2d", "dcrx_memcpy_d2h", "dcrx_memset_device", "dcrx_synchronize", "dcrx_event_create",
    "dcrx_event_destroy", "dcrx_event_record", "dcrx_event_elapsed_ms", "dcrx_abi_version", "dcrx_last_error",
    "dcrx_compact_hits_packed8_device", "dcrx_tuple_layout", "dcrx_tuple_message_bytes", "dcrx_compact_hits_narrow_device", "dcrx_set_tuple_sink", "dcrx_collapse_front", "dcrx_spacer_search", "dcrx_gzip_open", "dcrx_gzip_write", "dcrx_gzip_close", "dcrx_build_info", "dcrx_synth_reads_host", "dcrx_synth_reads_device", "dcrx_synth_exceptions_host",
    "dcrx_set_tune_wait", "dcrx_comm_available", "dcrx_comm_unique_id", "dcrx_comm_create", "dcrx_comm_create_all", "dcrx_comm_destroy", "dcrx_comm_info",
    "dcrx_comm_allreduce_u64", "dcrx_comm_allreduce_f64", "dcrx_comm_allgather", "dcrx_comm_gather_v", "dcrx_comm_barrier", "dcrx_comm_allgather_host",
    "dcrx_comm_allreduce_host_u64", "dcrx_decombine_sharded", "dcrx_event_synchronize", "dcrx_event_create_ordering", "dcrx_stream_create", "dcrx_stream_destroy", "dcrx_stream_synchronize",
    "dcrx_stream_wait_event", "dcrx_memcpy_d2h_async", "dcrx_memcpy_d2d_async", "dcrx_memset_device_async",
]

_lib = None


def lib():
    """The loaded library; raises ImportError with build instructions when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the decombine hot path.")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "dcrx_tables_create": (i32, [C.POINTER(TagSetC), C.POINTER(vp)]),
        "dcrx_tables_destroy": (None, [vp]),
        "dcrx_tables_info": (i32, [vp, C.POINTER(TablesInfoC)]),
        "dcrx_pack_reads": (C.c_int64, [vp, vp, u64, u32, vp, vp, vp, vp, vp, u64]),
        "dcrx_pack_reads_span": (C.c_int64, [vp, vp, vp, u64, u32, vp, vp, vp, vp, vp, u64]),
        "dcrx_unpack_reads": (i32, [C.POINTER(BatchC), vp, vp]),
        "dcrx_fastq_open": (i32, [C.c_char_p, i32, C.POINTER(vp)]),
        "dcrx_fastq_close": (None, [vp]),
        "dcrx_fastq_open_range": (i32, [C.c_char_p, u64, u64, C.POINTER(vp)]),
        "dcrx_fastq_lines": (i32, [C.c_char_p, u64, u64, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32), C.POINTER(u64)]),
        "dcrx_fastq_next": (i32, [vp, u64, C.POINTER(FastqBatchC)]),
        "dcrx_count_prefix_byte": (u64, [vp, vp, vp, u64, u32, i32]),
        "dcrx_assemble_rows": (C.c_int64, [vp, u64, C.POINTER(SpansC), C.POINTER(SpansC), C.POINTER(SpansC),
                                           C.POINTER(SpansC), C.POINTER(SpansC), C.POINTER(SpansC), C.c_char_p, vp, u64,
                                           C.POINTER(u64)]),
        "dcrx_decombine": (i32, [vp, C.POINTER(CfgC), C.POINTER(BatchC), vp, vp]),
        "dcrx_decombine_device": (i32, [vp, C.POINTER(CfgC), C.POINTER(BatchC), vp, vp, vp]),
        "dcrx_set_timing_events": (i32, [vp, vp, vp]),
        "dcrx_set_step_events": (i32, [vp, vp, vp]),
        "dcrx_reserve_device": (i32, [vp, u64]),
        "dcrx_compact_hits_device": (i32, [vp, u64, u64, vp, vp, vp, vp]),
        "dcrx_compact_hits_bitmap_device": (i32, [vp, u64, vp, vp, vp, vp]),
        "dcrx_compact_hits_packed_device": (i32, [vp, u64, vp, vp, vp, vp]),
        "dcrx_compact_hits_packed8_device": (i32, [vp, u64, vp, vp, vp, vp]),
        "dcrx_tuple_layout": (i32, [vp, u32, C.POINTER(TupleLayoutC)]),
        "dcrx_tuple_message_bytes": (u64, [C.POINTER(TupleLayoutC), u64, u64]),
        "dcrx_compact_hits_narrow_device": (i32, [vp, C.POINTER(TupleLayoutC), vp, u64, u64, vp, vp, vp]),
        "dcrx_set_tuple_sink": (i32, [vp, C.POINTER(TupleLayoutC), vp, u64, vp]),
        "dcrx_collapse_front": (C.c_int64, [vp, u64, C.POINTER(CollapseCfgC), vp, u64, vp, vp, i32]),
        "dcrx_spacer_search": (i32, [C.c_char_p, i32, C.c_char_p, i32, vp, vp, i32, C.POINTER(C.c_int32)]),
        "dcrx_set_reserved_cus": (i32, [vp, u32]),
        "dcrx_tune_state": (i32, [vp, i32, u64, C.POINTER(TuneStateC)]),
        "dcrx_cdr3_batch": (C.c_int64, [C.POINTER(Cdr3GenesC), u64, vp, vp, vp, vp, vp, vp, vp, vp, u64]),
        "dcrx_gzip_open": (i32, [C.c_char_p, i32, i32, C.POINTER(vp)]),
        "dcrx_gzip_write": (i32, [vp, vp, u64]),
        "dcrx_gzip_close": (i32, [vp]),
        "dcrx_device_count": (i32, []),
        "dcrx_set_device": (i32, [i32]),
        "dcrx_device_name": (i32, [C.c_char_p, C.c_size_t]),
        "dcrx_malloc_device": (i32, [C.POINTER(vp), C.c_size_t]),
        "dcrx_free_device": (i32, [vp]),
        "dcrx_malloc_host": (i32, [C.POINTER(vp), C.c_size_t]),
        "dcrx_free_host": (i32, [vp]),
        "dcrx_memcpy_h2d": (i32, [vp, vp, C.c_size_t]),
        "dcrx_memcpy_d2h": (i32, [vp, vp, C.c_size_t]),
        "dcrx_memset_device": (i32, [vp, i32, C.c_size_t]),
        "dcrx_synchronize": (i32, []),
        "dcrx_event_create": (i32, [C.POINTER(vp)]),
        "dcrx_event_destroy": (i32, [vp]),
        "dcrx_event_record": (i32, [vp, vp]),
        "dcrx_event_elapsed_ms": (i32, [vp, vp, C.POINTER(C.c_float)]),
        "dcrx_abi_version": (i32, []),
        "dcrx_last_error": (C.c_char_p, []),
        "dcrx_build_info": (C.c_char_p, []),
        "dcrx_synth_reads_host": (i32, [vp, C.POINTER(SynthCfgC), u64, u64, u32, vp]),
        "dcrx_synth_reads_device": (i32, [vp, C.POINTER(SynthCfgC), u64, u64, u32, vp, vp]),
        "dcrx_synth_exceptions_host": (C.c_int64, [vp, C.POINTER(SynthCfgC), u64, u64, vp, vp, vp, u64]),
        "dcrx_set_tune_wait": (i32, [vp, i32]),
        "dcrx_comm_available": (i32, []),
        "dcrx_comm_unique_id": (i32, [vp]),
        "dcrx_comm_create": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "dcrx_comm_create_all": (i32, [i32, vp, C.POINTER(vp)]),
        "dcrx_comm_destroy": (None, [vp]),
        "dcrx_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "dcrx_comm_allreduce_u64": (i32, [vp, vp, vp, u64, i32, vp]),
        "dcrx_comm_allreduce_f64": (i32, [vp, vp, vp, u64, i32, vp]),
        "dcrx_comm_allgather": (i32, [vp, vp, vp, u64, vp]),
        "dcrx_comm_gather_v": (i32, [vp, vp, u64, vp, vp, i32, vp]),
        "dcrx_comm_barrier": (i32, [vp, vp]),
        "dcrx_comm_allgather_host": (i32, [vp, vp, vp, u64]),
        "dcrx_comm_allreduce_host_u64": (i32, [vp, vp, u64, i32]),
        "dcrx_decombine_sharded": (i32, [vp, vp, C.POINTER(CfgC), C.POINTER(BatchC), vp, vp, C.POINTER(TupleLayoutC), vp, u64, vp, vp, vp]),
        "dcrx_event_synchronize": (i32, [vp]),
        "dcrx_event_create_ordering": (i32, [C.POINTER(vp)]),
        "dcrx_stream_create": (i32, [C.POINTER(vp)]),
        "dcrx_stream_destroy": (i32, [vp]),
        "dcrx_stream_synchronize": (i32, [vp]),
        "dcrx_stream_wait_event": (i32, [vp, vp]),
        "dcrx_memcpy_d2h_async": (i32, [vp, vp, C.c_size_t, vp]),
        "dcrx_memcpy_d2d_async": (i32, [vp, vp, C.c_size_t, vp]),
        "dcrx_memset_device_async": (i32, [vp, i32, C.c_size_t, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the .so does not match include/dcrx.h
        fn.restype = res
        fn.argtypes = args
    if L.dcrx_abi_version() != ABI_VERSION:
        raise ImportError(f"libdcrx.so ABI {L.dcrx_abi_version()} != binding {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise DcrxError(rc, lib().dcrx_last_error().decode("utf-8", "replace"))
    return rc


def _strs(xs):
    arr = (C.c_char_p * max(1, len(xs)))()
    for i, x in enumerate(xs):
        arr[i] = x.encode("ascii") if isinstance(x, str) else bytes(x)
    return arr


class Tables:
    """Opaque dcrx_tables_t: replaces import_tcr_info's module globals
    (reference decombine.py:593-746) for one chain."""

    def __init__(self, v_tags, v_jumps, v_regions, j_tags, j_jumps, j_regions,
                 v_half_split: int, j_half_split: int):
        if not (len(v_tags) == len(v_jumps) == len(v_regions)):
            raise ValueError("V tags, jumps and regions differ in length")
        if not (len(j_tags) == len(j_jumps) == len(j_regions)):
            raise ValueError("J tags, jumps and regions differ in length")
        self._keep = (_strs(v_tags), (C.c_int32 * max(1, len(v_jumps)))(*[int(x) for x in v_jumps]),
                      _strs(v_regions), _strs(j_tags),
                      (C.c_int32 * max(1, len(j_jumps)))(*[int(x) for x in j_jumps]), _strs(j_regions))
        ts = TagSetC(len(v_tags), self._keep[0], self._keep[1], self._keep[2],
                     len(j_tags), self._keep[3], self._keep[4], self._keep[5],
                     int(v_half_split), int(j_half_split))
        h = C.c_void_p()
        self._h = None
        check(lib().dcrx_tables_create(C.byref(ts), C.byref(h)))
        self._h = h
        self.n_v, self.n_j = len(v_tags), len(j_tags)
        # what a receiver of narrow tuples re-derives positions from (TupleCodec)
        self.v_jumps = [int(x) for x in v_jumps]
        self.j_jumps = [int(x) for x in j_jumps]
        self.j_lens = [len(x) for x in j_tags]
        self.j_half_split = int(j_half_split)

    @property
    def handle(self):
        return self._h

    def info(self) -> dict:
        inf = TablesInfoC()
        check(lib().dcrx_tables_info(self._h, C.byref(inf)))
        return {"n_v": inf.n_v, "n_j": inf.n_j, "n_states": inf.n_states, "dfa_bytes": inf.dfa_bytes,
                "n_keywords": list(inf.n_keywords), "max_tag_len": inf.max_tag_len,
                "tables_in_lds": bool(inf.tables_in_lds),
                "equal_len_per_automaton": bool(inf.equal_len_per_automaton), "pair_scan_bytes": inf.pair_scan_bytes,
                "v2_tables": bool(inf.v2_tables), "v2_states": list(inf.v2_states), "v2_scan_bytes": max(inf.v2_scan_bytes),
                "max_read_len": inf.max_read_len}

    def tune_state(self, n_reads: int, orientation: str = "reverse") -> dict:
        """dcrx_tune_state: what the handle has settled for the finishing launches of batches of n_reads' size class."""
        st = TuneStateC()
        check(lib().dcrx_tune_state(self._h, ORIENTATIONS[orientation], int(n_reads), C.byref(st)))
        first, second = int(st.candidates) & 0xFFFF, int(st.candidates) >> 16
        return {"rescue_waves": int(st.rescue_waves), "launches": int(st.launches), f"us_{first}": round(float(st.us_first), 2),
                f"us_{second}": round(float(st.us_second), 2),
                "launch_form": {0: "none yet", 1: "three-launch form", 2: "v2, tail as a role", 3: "v2, tail inside the scan", 4: "v2, tail and list E inside the scan"}.get(int(st.launch_form), "?")}

    def close(self):
        if self._h is not None:
            lib().dcrx_tables_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stride_for(max_len: int) -> int:
    """Smallest legal stride (multiple of 8 bytes) for reads of up to max_len nt."""
    return max(8, ((max_len + 3) // 4 + 7) // 8 * 8)


class PackedBatch:
    """Host-side packed reads (what dcrx_pack_reads produces)."""

    def __init__(self, packed, stride, read_len, lens, exc_read, exc_pos, exc_chr):
        self.packed = packed
        self.stride = int(stride)
        self.read_len = int(read_len)
        self.lens = lens
        self.exc_read, self.exc_pos, self.exc_chr = exc_read, exc_pos, exc_chr

    @property
    def n_reads(self) -> int:
        return self.packed.shape[0]

    def as_c(self) -> BatchC:
        b = BatchC()
        b.n_reads = self.n_reads
        b.packed = self.packed.ctypes.data
        b.stride = self.stride
        b.read_len = self.read_len
        b.lens = self.lens.ctypes.data if self.lens is not None else None
        b.n_exc = len(self.exc_read)
        b.exc_read = self.exc_read.ctypes.data if len(self.exc_read) else None
        b.exc_pos = self.exc_pos.ctypes.data if len(self.exc_read) else None
        b.exc_chr = self.exc_chr.ctypes.data if len(self.exc_read) else None
        return b


def pack_reads(reads, stride: int | None = None) -> PackedBatch:
    """Packs a list of str/bytes reads (or a (buffer, offsets) pair) 2 bits per base."""
    if isinstance(reads, tuple):
        buf, offsets = reads
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    else:
        bs = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in reads]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
    n = len(offsets) - 1
    lens64 = np.diff(offsets.astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
    max_len = int(lens64.max()) if n else 0
    if stride is None:
        stride = stride_for(max_len)
    packed = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    cap = 1024
    while True:
        er = np.zeros(cap, dtype=np.uint32)
        ep = np.zeros(cap, dtype=np.uint16)
        ec = np.zeros(cap, dtype=np.uint8)
        got = lib().dcrx_pack_reads(buf.ctypes.data if len(buf) else None, offsets.ctypes.data, n, stride,
                                    packed.ctypes.data if n else None, lens.ctypes.data, er.ctypes.data,
                                    ep.ctypes.data, ec.ctypes.data, cap)
        check(int(got))
        if got <= cap:
            break
        cap = int(got)
    uniform = n > 0 and bool((lens64 == lens64[0]).all())
    return PackedBatch(packed, stride, int(lens64[0]) if uniform else 0, None if uniform else lens,
                       er[:got].copy(), ep[:got].copy(), ec[:got].copy())


def pack_reads_span(text, start, length, stride: int | None = None) -> PackedBatch:
    """Packs the reads text[start[r] : start[r]+length[r]] (bytes / uint8 buffer, uint64, uint32)."""
    buf = text if isinstance(text, np.ndarray) else np.frombuffer(text, dtype=np.uint8)
    start = np.ascontiguousarray(start, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    n = len(start)
    max_len = int(length.max()) if n else 0
    if stride is None:
        stride = stride_for(max_len)
    packed = np.empty((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint16)
    cap = 1024 + n // 64
    while True:
        er = np.zeros(cap, dtype=np.uint32)
        ep = np.zeros(cap, dtype=np.uint16)
        ec = np.zeros(cap, dtype=np.uint8)
        got = lib().dcrx_pack_reads_span(buf.ctypes.data if len(buf) else None, start.ctypes.data, length.ctypes.data, n,
                                         stride, packed.ctypes.data if n else None, lens.ctypes.data, er.ctypes.data,
                                         ep.ctypes.data, ec.ctypes.data, cap)
        check(int(got))
        if got <= cap:
            break
        cap = int(got)
    uniform = n > 0 and bool((length == length[0]).all())
    return PackedBatch(packed, stride, int(length[0]) if uniform else 0, None if uniform else lens,
                       er[:got].copy(), ep[:got].copy(), ec[:got].copy())


NO_QUAL = 0xFFFFFFFF
FAST_MAX_READ_LEN = 511      # reads up to this length run on the register shapes; longer ones (to dcrx_tables_info.max_read_len) in batches of their own


class FastqBatch:
    """One batch of records from FastqReader: `text` (bytes) plus offset / length arrays
    (qual_len == NO_QUAL where the reference's readfq yields qual None)."""

    __slots__ = ("n", "text", "name_off", "name_len", "seq_off", "seq_len", "qual_off", "qual_len")

    def __init__(self, c: FastqBatchC, copy: bool = True):
        n = self.n = int(c.n_records)
        # zero-copy window on the reader's buffer: valid until the next call of next() on that reader
        self.text = (C.c_char * int(c.text_bytes)).from_address(c.text) if c.text_bytes else b""

        def arr(ptr, ct, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
            # copy=False: the offset and length arrays are windows on the reader's store as well (36 MB per million records not
            # copied) — valid, like the text, until the next call of next() on that reader
            return a.astype(dt, copy=True) if copy else a
        self.name_off, self.name_len = arr(c.name_off, C.c_uint64, np.uint64), arr(c.name_len, C.c_uint32, np.uint32)
        self.seq_off, self.seq_len = arr(c.seq_off, C.c_uint64, np.uint64), arr(c.seq_len, C.c_uint32, np.uint32)
        self.qual_off, self.qual_len = arr(c.qual_off, C.c_uint64, np.uint64), arr(c.qual_len, C.c_uint32, np.uint32)

    def truncate(self, n: int) -> None:
        self.n = n
        for a in ("name_off", "name_len", "seq_off", "seq_len", "qual_off", "qual_len"):
            setattr(self, a, getattr(self, a)[:n])

    def name(self, k: int) -> str:
        o = int(self.name_off[k])
        return self.text[o:o + int(self.name_len[k])].decode("utf-8", "replace")

    def seq(self, k: int) -> str:
        o = int(self.seq_off[k])
        return self.text[o:o + int(self.seq_len[k])].decode("latin-1")

    def qual(self, k: int):
        if int(self.qual_len[k]) == NO_QUAL:
            return None
        o = int(self.qual_off[k])
        return self.text[o:o + int(self.qual_len[k])].decode("latin-1")

    def records(self):
        """(name, seq, qual) tuples, as readfq yields them."""
        return [(self.name(k), self.seq(k), self.qual(k)) for k in range(self.n)]


class FastqReader:
    """Batch FASTQ / FASTA reader in libdcrx (dcrx_fastq_*): same records as readfq()."""

    def __init__(self, path: str, gzipped: bool | None = None, byte_range=None):
        """byte_range = (begin, end): the records of that byte range of a plain four-line FASTQ file only (a shard)."""
        if gzipped is None:
            gzipped = str(path).endswith(".gz")
        self._h = C.c_void_p()
        if byte_range is not None:
            check(lib().dcrx_fastq_open_range(os.fsencode(path), int(byte_range[0]), int(byte_range[1]), C.byref(self._h)))
        else:
            check(lib().dcrx_fastq_open(os.fsencode(path), int(bool(gzipped)), C.byref(self._h)))

    def next(self, max_records: int, copy: bool = True) -> FastqBatch:
        c = FastqBatchC()
        check(lib().dcrx_fastq_next(self._h, int(max_records), C.byref(c)))
        return FastqBatch(c, copy)

    def close(self) -> None:
        if self._h:
            lib().dcrx_fastq_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fastq_lines(path: str, begin: int, end: int, nth: int = 0, count: bool = True):
    """(newlines in the bytes [begin, end) of the file, offset just behind the nth of them or None, a carriage return occurs,
    file size) — dcrx_fastq_lines; count=False: only the offset (the scan stops at the nth newline)."""
    n, off, cr, size = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_uint64(0)
    check(lib().dcrx_fastq_lines(os.fsencode(path), int(begin), int(end), int(nth), C.byref(n) if count else None, C.byref(off), C.byref(cr),
                                 C.byref(size)))
    return int(n.value), (None if off.value == 0xFFFFFFFFFFFFFFFF else int(off.value)), bool(cr.value), int(size.value)


def count_prefix_byte(text: bytes, start, length, prefix: int, byte: str) -> int:
    start = np.ascontiguousarray(start, dtype=np.uint64)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    if len(start) == 0 or prefix <= 0:
        return 0
    buf = np.frombuffer(text, dtype=np.uint8)
    return int(lib().dcrx_count_prefix_byte(buf.ctypes.data, start.ctypes.data, length.ctypes.data, len(start),
                                            int(prefix), ord(byte)))


class SeparatorClash(RuntimeError):
    """A row field contains the field separator byte: the caller assembles that batch row by row."""


def _uninitialised_bytes(n: int) -> bytes:
    """A bytes object of n bytes that nobody has written yet (CPython: PyBytes_FromStringAndSize(NULL, n)); the caller fills
    it through _bytes_address() before anyone else sees it."""
    f = C.pythonapi.PyBytes_FromStringAndSize
    f.restype, f.argtypes = C.py_object, [C.c_char_p, C.c_ssize_t]
    return f(None, n)


def _bytes_address(b: bytes) -> int:
    f = C.pythonapi.PyBytes_AsString
    f.restype, f.argtypes = C.c_void_p, [C.py_object]
    return f(b)


def assemble_rows_blob(records, vdj, qual, ident, bc, bcq, tail=None, field_sep: str = ", "):
    """dcrx_assemble_rows: each argument after `records` is (text bytes, uint64 start[], uint32 len[]).
    Returns (bytes blob of '\n'-terminated rows, number of rows)."""
    keep = []

    def spans(t):
        if t is None:
            return None
        text, start, length = t
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(1, dtype=np.uint8)
        start = np.ascontiguousarray(start, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        keep.extend((buf, start, length))
        return C.byref(SpansC(buf.ctypes.data, start.ctypes.data, length.ctypes.data))
    records = np.ascontiguousarray(records)
    args = [spans(t) for t in (vdj, qual, ident, bc, bcq, tail)]
    nrows = C.c_uint64(0)
    sep = field_sep.encode("latin-1")
    need = int(lib().dcrx_assemble_rows(records.ctypes.data, len(records), *args, sep, None, 0, C.byref(nrows)))
    check(need)
    # the text as a bytes object whose buffer the library fills: bytearray(need) zeroes its half a gigabyte on one thread first
    # (0.24 s of a 0.37 s call for 4 M reads on the build container), a fresh bytes object's pages are first touched by the
    # library's sixteen writers
    out = _uninitialised_bytes(need)
    if need:
        got = int(lib().dcrx_assemble_rows(records.ctypes.data, len(records), *args, sep, _bytes_address(out), need,
                                           C.byref(nrows)))
        if got == -2:                   # DCRX_E_UNSUPPORTED
            raise SeparatorClash(lib().dcrx_last_error().decode("utf-8", "replace"))
        check(got)
        if got != need:
            raise DcrxError(-1, f"dcrx_assemble_rows wrote {got} of {need} bytes")
    return out, int(nrows.value)


def unpack_reads_raw(batch: PackedBatch):
    """Inverse of pack_reads into one ASCII buffer: (uint8 buffer with a trailing NUL, uint64 offsets[n+1])."""
    n = batch.n_reads
    lens = batch.lens.astype(np.uint64) if batch.lens is not None else np.full(n, batch.read_len, dtype=np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    out = np.zeros(int(offsets[-1]) + 1, dtype=np.uint8)
    b = batch.as_c()
    check(lib().dcrx_unpack_reads(C.byref(b), offsets.ctypes.data, out.ctypes.data))
    return out, offsets


def unpack_reads(batch: PackedBatch):
    """Inverse of pack_reads: list of str."""
    n = batch.n_reads
    lens = batch.lens.astype(np.uint64) if batch.lens is not None else np.full(n, batch.read_len, dtype=np.uint64)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    out = np.zeros(int(offsets[-1]) + 1, dtype=np.uint8)
    b = batch.as_c()
    check(lib().dcrx_unpack_reads(C.byref(b), offsets.ctypes.data, out.ctypes.data))
    raw = out.tobytes()
    return [raw[int(offsets[i]):int(offsets[i + 1])].decode("latin-1") for i in range(n)]


def make_cfg(orientation="reverse", allow_ns=False, lenthreshold=130, flags=0) -> CfgC:
    o = ORIENTATIONS[orientation] if isinstance(orientation, str) else int(orientation)
    return CfgC(o, int(bool(allow_ns)), int(lenthreshold), int(flags))


class GzipWriter:
    """dcrx_gzip_open / write / close: a gzip file written by several threads (multi-member; any gzip reader reads it as one
    stream).  Stands for the reference's `gzip.open(name, "wt")` + writelines (io.py:497-506)."""

    def __init__(self, path: str, level: int = 6, n_threads: int = 0):
        self._h = C.c_void_p()
        check(lib().dcrx_gzip_open(os.fsencode(path), int(level), int(n_threads), C.byref(self._h)))

    def write(self, data) -> None:
        mv = memoryview(data).cast("B")
        if mv.nbytes:
            arr = np.frombuffer(mv, dtype=np.uint8)
            check(lib().dcrx_gzip_write(self._h, arr.ctypes.data, mv.nbytes))

    def close(self) -> None:
        if self._h:
            h, self._h = self._h, C.c_void_p()
            check(lib().dcrx_gzip_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """Page-locked host memory from dcrx_malloc_host, freed with the last array that views it."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        check(lib().dcrx_malloc_host(C.byref(self.ptr), max(1, int(nbytes))))
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().dcrx_free_host(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """An uninitialised numpy array in page-locked host memory: dcrx_decombine copies from / into such buffers directly."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    blk = _PinnedBlock(n)
    buf = (C.c_uint8 * max(1, n)).from_address(blk.ptr.value)
    buf._dcrx_block = blk           # the ctypes object, which the array keeps alive, keeps the block alive
    return np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)


def decombine(tables: Tables, batch: PackedBatch, orientation="reverse", allow_ns=False,
              lenthreshold=130, flags=0, out: np.ndarray | None = None):
    """dcrx_decombine on host buffers.  Returns (records[RECORD_DTYPE], counters uint64[32]).  `out`: a records array to
    fill (e.g. pinned_empty(n, RECORD_DTYPE), reused from call to call)."""
    cfg = make_cfg(orientation, allow_ns, lenthreshold, flags)
    if out is not None:
        assert out.dtype == RECORD_DTYPE and out.shape == (batch.n_reads,) and out.flags.c_contiguous
    rec = out if out is not None else np.zeros(batch.n_reads, dtype=RECORD_DTYPE)
    cnt = np.zeros(N_COUNTERS, dtype=np.uint64)
    b = batch.as_c()
    check(lib().dcrx_decombine(tables.handle, C.byref(cfg), C.byref(b),
                               rec.ctypes.data if batch.n_reads else None, cnt.ctypes.data))
    if int(cnt[DEVICE_ERRORS]):      # (include/dcrx_codes.h: a wave gave up waiting for another — the records are not complete)
        raise RuntimeError(f"dcrx_decombine: {int(cnt[DEVICE_ERRORS])} device-side wait(s) timed out; the records of this call are incomplete")
    return rec, cnt


def synth_cfg(seed: int, read_len: int = 150, p_rearranged: float = 0.45, sub_rate: float = 0.005,
              n_rate: float = 0.0005) -> SynthCfgC:
    return SynthCfgC(int(seed), int(read_len), float(p_rearranged), float(sub_rate), float(n_rate))


def synth_reads_host(tables: Tables, cfg: SynthCfgC, first: int, n: int, stride: int | None = None, pinned: bool = False) -> PackedBatch:
    """Seeded synthetic reads [first, first+n) generated by the library on the host."""
    if stride is None:
        stride = stride_for(cfg.read_len)
    packed = pinned_empty((n, stride), np.uint8) if pinned else np.zeros((n, stride), dtype=np.uint8)
    check(lib().dcrx_synth_reads_host(tables.handle, C.byref(cfg), first, n, stride,
                                      packed.ctypes.data if n else None))
    er, ep, ec = synth_exceptions_host(tables, cfg, first, n)
    return PackedBatch(packed, stride, cfg.read_len, None, er, ep, ec)


def synth_exceptions_host(tables: Tables, cfg: SynthCfgC, first: int, n: int):
    cap = max(16, int(n * cfg.n_rate * 2) + 64)
    while True:
        er = np.zeros(cap, dtype=np.uint32)
        ep = np.zeros(cap, dtype=np.uint16)
        ec = np.zeros(cap, dtype=np.uint8)
        got = int(lib().dcrx_synth_exceptions_host(tables.handle, C.byref(cfg), first, n, er.ctypes.data,
                                                   ep.ctypes.data, ec.ctypes.data, cap))
        check(got)
        if got <= cap:
            return er[:got].copy(), ep[:got].copy(), ec[:got].copy()
        cap = got


def device_count() -> int:
    return int(lib().dcrx_device_count())


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().dcrx_device_name(buf, 256))
    return buf.value.decode()


# ---- device-resident path (no torch needed: the library owns the HIP calls) ----

class DeviceBuffer:
    """A hipMalloc'd buffer owned through the C ABI."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().dcrx_malloc_device(C.byref(p), int(nbytes)))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    @classmethod
    def from_host(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(max(arr.nbytes, 16))
        if arr.nbytes:
            check(lib().dcrx_memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def to_host(self, dtype, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        if out.nbytes:
            check(lib().dcrx_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().dcrx_free_device(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBatch:
    """Packed reads resident in HBM (the buffers dcrx_decombine_device reads)."""

    def __init__(self, n_reads: int, stride: int, read_len: int, packed: DeviceBuffer,
                 lens: DeviceBuffer | None = None, exc=None):
        self.n_reads, self.stride, self.read_len = int(n_reads), int(stride), int(read_len)
        self.packed, self.lens = packed, lens
        self.exc = exc  # (n_exc, DeviceBuffer read, DeviceBuffer pos, DeviceBuffer chr) or None

    @classmethod
    def from_host(cls, b: PackedBatch) -> "DeviceBatch":
        exc = None
        if len(b.exc_read):
            exc = (len(b.exc_read), DeviceBuffer.from_host(b.exc_read), DeviceBuffer.from_host(b.exc_pos),
                   DeviceBuffer.from_host(b.exc_chr))
        # 16 spare bytes after the last read (word-pair loads of the final read)
        packed = DeviceBuffer(b.packed.nbytes + 16)
        if b.packed.nbytes:
            check(lib().dcrx_memcpy_h2d(packed.ptr, b.packed.ctypes.data, b.packed.nbytes))
        return cls(b.n_reads, b.stride, b.read_len, packed,
                   DeviceBuffer.from_host(b.lens) if b.lens is not None else None, exc)

    def as_c(self) -> BatchC:
        b = BatchC()
        b.n_reads, b.packed, b.stride, b.read_len = self.n_reads, self.packed.ptr, self.stride, self.read_len
        b.lens = self.lens.ptr if self.lens is not None else None
        if self.exc:
            b.n_exc, b.exc_read, b.exc_pos, b.exc_chr = self.exc[0], self.exc[1].ptr, self.exc[2].ptr, self.exc[3].ptr
        else:
            b.n_exc, b.exc_read, b.exc_pos, b.exc_chr = 0, None, None, None
        return b


def synth_reads_device(tables: Tables, cfg: SynthCfgC, first: int, n: int, stride: int | None = None,
                       stream=None) -> DeviceBatch:
    """Same reads as synth_reads_host, generated in HBM by the library's kernel."""
    if stride is None:
        stride = stride_for(cfg.read_len)
    packed = DeviceBuffer(n * stride + 16)
    check(lib().dcrx_synth_reads_device(tables.handle, C.byref(cfg), first, n, stride, packed.ptr, stream))
    er, ep, ec = synth_exceptions_host(tables, cfg, first, n)
    exc = None
    if len(er):
        exc = (len(er), DeviceBuffer.from_host(er), DeviceBuffer.from_host(ep), DeviceBuffer.from_host(ec))
    return DeviceBatch(n, stride, cfg.read_len, packed, None, exc)


def decombine_device(tables: Tables, batch: DeviceBatch, d_records: DeviceBuffer, d_counters: DeviceBuffer,
                     orientation="reverse", allow_ns=False, lenthreshold=130, flags=0, stream=None):
    """Asynchronous launch on `stream` (None = default); results stay in HBM."""
    cfg = make_cfg(orientation, allow_ns, lenthreshold, flags)
    b = batch.as_c()
    check(lib().dcrx_decombine_device(tables.handle, C.byref(cfg), C.byref(b), d_records.ptr, d_counters.ptr, stream))


def synchronize():
    check(lib().dcrx_synchronize())


class Event:
    def __init__(self, timing: bool = True):
        p = C.c_void_p()
        check((lib().dcrx_event_create if timing else lib().dcrx_event_create_ordering)(C.byref(p)))      # (timing=False: for ordering streams only, cheaper to record)
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().dcrx_event_record(self.ptr, stream))

    def synchronize(self):
        check(lib().dcrx_event_synchronize(self.ptr))

    def elapsed_ms(self, later: "Event") -> float:
        ms = C.c_float()
        check(lib().dcrx_event_elapsed_ms(self.ptr, later.ptr, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.ptr:
                lib().dcrx_event_destroy(self.ptr)
        except Exception:
            pass


class Stream:
    """A HIP stream of the caller's own (dcrx_stream_create): a sharded run's side stream for the exchange."""

    def __init__(self):
        p = C.c_void_p()
        check(lib().dcrx_stream_create(C.byref(p)))
        self.ptr = p.value

    def synchronize(self):
        check(lib().dcrx_stream_synchronize(self.ptr))

    def wait_event(self, ev: "Event"):
        check(lib().dcrx_stream_wait_event(self.ptr, ev.ptr))

    def __del__(self):
        try:
            if self.ptr:
                lib().dcrx_stream_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


# ---- multi-GPU: RCCL bound by the library itself (include/dcrx.h, "multi-GPU") ----
COMM_ID_BYTES = 128
COMM_SUM, COMM_MAX = 0, 1


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL greets on the C library's stdout when a communicator is made (version, host, library path): a process whose stdout
    is a result — bench.py prints ONE JSON line there — sends that to stderr instead (the descriptor itself, around the call)."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class Comm:
    """An RCCL communicator through the C ABI: one process per GPU (`Comm(uid, world, rank)` after dcrx_set_device), the id drawn
    by rank 0 (`Comm.unique_id()`) and carried to the others by the caller — `comm_from_env()` does that for ranks started on one
    node by torch.distributed.run or by bench.py's own spawn.  No torch anywhere."""

    def __init__(self, uid: bytes, world: int, rank: int):
        assert len(uid) == COMM_ID_BYTES
        self._uid = np.frombuffer(uid, dtype=np.uint8).copy()
        h = C.c_void_p()
        with _c_stdout_to_stderr():
            rc = lib().dcrx_comm_create(self._uid.ctypes.data, int(world), int(rank), C.byref(h))
        check(rc)
        self.handle, self.world, self.rank = h.value, int(world), int(rank)

    @staticmethod
    def available() -> bool:
        return bool(lib().dcrx_comm_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = np.zeros(COMM_ID_BYTES, dtype=np.uint8)
        check(lib().dcrx_comm_unique_id(buf.ctypes.data))
        return buf.tobytes()

    # --- device memory, asynchronous on `stream` (a pointer or None) ---
    def allreduce_u64(self, d_in: int, d_out: int, n: int, op: int = COMM_SUM, stream=None):
        check(lib().dcrx_comm_allreduce_u64(self.handle, d_in, d_out, int(n), op, stream))

    def allreduce_f64(self, d_in: int, d_out: int, n: int, op: int = COMM_MAX, stream=None):
        check(lib().dcrx_comm_allreduce_f64(self.handle, d_in, d_out, int(n), op, stream))

    def allgather(self, d_in: int, d_out: int, bytes_per_rank: int, stream=None):
        check(lib().dcrx_comm_allgather(self.handle, d_in, d_out, int(bytes_per_rank), stream))

    def gather_v(self, d_send: int, send_bytes: int, d_recv=None, recv_bytes=None, root: int = 0, stream=None):
        """Every rank but `root` sends send_bytes from d_send; the root receives recv_bytes[r] into d_recv[r] (one group)."""
        ptrs = sizes = None
        if self.rank == root and self.world > 1:
            ptrs = (C.c_void_p * self.world)(*[int(p) if p else None for p in d_recv])
            sizes = (C.c_uint64 * self.world)(*[int(b) for b in recv_bytes])
        check(lib().dcrx_comm_gather_v(self.handle, d_send, int(send_bytes), ptrs, sizes, int(root), stream))

    def barrier(self, stream=None):
        check(lib().dcrx_comm_barrier(self.handle, stream))

    # --- host memory, synchronous: the control plane of a sharded run ---
    def allgather_host(self, arr: np.ndarray) -> np.ndarray:
        """Every rank's array (same shape and dtype on all ranks), stacked in rank order."""
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + a.shape, dtype=a.dtype)
        check(lib().dcrx_comm_allgather_host(self.handle, a.ctypes.data if a.nbytes else None, out.ctypes.data if out.nbytes else None, a.nbytes))
        return out

    def allreduce_host_u64(self, values, op: int = COMM_SUM) -> np.ndarray:
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        check(lib().dcrx_comm_allreduce_host_u64(self.handle, a.ctypes.data if a.size else None, a.size, op))
        return a

    def allgather_bytes(self, blob: bytes) -> list:
        """Every rank's bytes, in rank order (sizes first, then one padded all-gather)."""
        sizes = self.allgather_host(np.array([len(blob)], dtype=np.uint64)).reshape(-1)
        kmax = int(sizes.max()) if sizes.size else 0
        if kmax == 0:
            return [b"" for _ in range(self.world)]
        pad = np.zeros(kmax, dtype=np.uint8)
        pad[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        got = self.allgather_host(pad)
        return [got[r, :int(sizes[r])].tobytes() for r in range(self.world)]

    def allgather_object(self, obj) -> list:
        import pickle
        return [pickle.loads(b) for b in self.allgather_bytes(pickle.dumps(obj))]

    def gather_bytes(self, blob: bytes, dst: int = 0):
        """Every rank's bytes on `dst`, in rank order (a list there, None elsewhere): the sizes (an all-gather), then one exact-size
        transfer per peer from device memory into device memory (dcrx_comm_gather_v), waited for before this returns."""
        sizes = [int(x) for x in self.allgather_host(np.array([len(blob)], dtype=np.uint64)).reshape(-1)]
        if self.world == 1:
            return [blob]
        mine = DeviceBuffer.from_host(np.frombuffer(blob, dtype=np.uint8)) if len(blob) else DeviceBuffer(16)
        recv = None
        if self.rank == dst:
            recv = [DeviceBuffer(max(n, 16)) if r != dst else None for r, n in enumerate(sizes)]
        self.gather_v(mine.ptr, len(blob), [b.ptr if b is not None else None for b in recv] if recv else None, sizes if recv else None, dst, None)
        synchronize()
        if self.rank != dst:
            return None
        return [blob if r == dst else recv[r].to_host(np.uint8, sizes[r]).tobytes() for r in range(self.world)]

    def close(self):
        if getattr(self, "handle", None):
            with _c_stdout_to_stderr():
                lib().dcrx_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_from_env(timeout_s: float = 300.0) -> Comm:
    """The communicator of a rank started with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT in its environment (one node:
    torch.distributed.run, bench.py's spawn).  Rank 0 draws the id and leaves it in a file of the node's temporary directory —
    named for the launcher's port and the launcher's process, which every rank of one launch shares and no other launch does —
    written under another name and renamed, so that a reader never sees half of it; the others wait for the file.  Rank 0
    removes it once every rank has joined (the communicator exists).  DCRX_COMM_ID_FILE overrides the path (a shared file
    system carries the id between nodes the same way)."""
    import tempfile
    import time
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    path = os.environ.get("DCRX_COMM_ID_FILE")
    if not path:
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        path = os.path.join(tempfile.gettempdir(), f"dcrx_comm_id_{key}")
    if rank == 0:
        uid = Comm.unique_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(uid)
        os.replace(tmp, path)
    else:
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as fh:
                    uid = fh.read()
                if len(uid) == COMM_ID_BYTES:
                    break
            except OSError:
                pass
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout_s:.0f} s (did rank 0 start?)")
            time.sleep(0.01)
    comm = Comm(uid, world, rank)
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return comm


def compact_hits_bitmap_device(d_records: DeviceBuffer, n_reads: int, d_hits: DeviceBuffer, d_ok_bitmap: DeviceBuffer,
                               d_n_hits: DeviceBuffer, stream=None):
    """Decombined records in input order + a bitmap of which reads they belong to ((n_reads+63)//64 uint64)."""
    check(lib().dcrx_compact_hits_bitmap_device(d_records.ptr, n_reads, d_hits.ptr, d_ok_bitmap.ptr, d_n_hits.ptr, stream))


def pack_tuples8(rec) -> np.ndarray:
    """Host-side twin of dcrx_compact_hits_packed8_device: the status-OK records of `rec` as (k, 2) uint32 tuples in read order."""
    r = rec[rec["status"] == 0]
    w = np.zeros((len(r), 2), dtype=np.uint32)
    jd = r["jdel"].astype(np.uint32)
    w[:, 0] = (r["v"].astype(np.uint32) & 0x7FF) | ((r["j"].astype(np.uint32) & 0x1FF) << 11) | (r["vdel"].astype(np.uint32) << 20) | ((jd & 0xF) << 28)
    w[:, 1] = (jd >> 4) | ((r["v_start"].astype(np.uint32) & 0x1FF) << 4) | ((r["j_end"].astype(np.uint32) & 0x1FF) << 13) | \
              ((r["ins_len"].astype(np.uint32) & 0x1FF) << 22) | ((r["frame"].astype(np.uint32) & 1) << 31)
    return w


def unpack_tuples8(words: np.ndarray, v_jumps) -> np.ndarray:
    """(k, 2) uint32 tuples of dcrx_compact_hits_packed8_device -> RECORD_DTYPE records (status OK).  ins_start, which the
    tuple leaves out, is the base after the end of V: v_start + jump_to_end_v[v] - vdel (decombine.py:283-285, :547, :577)."""
    w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 2)
    rec = np.zeros(len(w), dtype=RECORD_DTYPE)
    rec["v"] = w[:, 0] & 0x7FF
    rec["j"] = (w[:, 0] >> 11) & 0x1FF
    rec["vdel"] = (w[:, 0] >> 20) & 0xFF
    rec["jdel"] = ((w[:, 0] >> 28) & 0xF) | ((w[:, 1] & 0xF) << 4)
    rec["v_start"] = (w[:, 1] >> 4) & 0x1FF
    rec["j_end"] = (w[:, 1] >> 13) & 0x1FF
    rec["ins_len"] = (w[:, 1] >> 22) & 0x1FF
    rec["frame"] = (w[:, 1] >> 31) & 1
    jumps = np.asarray(v_jumps, dtype=np.int64)
    rec["ins_start"] = rec["v_start"].astype(np.int64) + jumps[rec["v"].astype(np.int64)] - rec["vdel"].astype(np.int64)
    return rec


class TupleCodec:
    """The narrow tuple of include/dcrx.h (dcrx_tuple_layout) on the host: the layout of a tag set for reads of up to
    max_read_len nt, the host-side twin of dcrx_compact_hits_narrow_device (`pack`) and the receiver's side (`unpack`)."""

    def __init__(self, tables: "Tables", max_read_len: int):
        self.layout = TupleLayoutC()
        check(lib().dcrx_tuple_layout(tables.handle, int(max_read_len), C.byref(self.layout)))
        L = self.layout
        self.bytes, self.bits = int(L.bytes), int(L.bits)
        self.widths = [int(L.w_v), int(L.w_j), int(L.w_vdel), int(L.w_jdel), int(L.w_pos), int(L.w_pos), 1, 1]
        self.v_jumps = np.asarray(tables.v_jumps, dtype=np.int64)
        self.j_jumps = np.asarray(tables.j_jumps, dtype=np.int64)
        self.j_lens = np.asarray(tables.j_lens, dtype=np.int64)
        self.j_short = 2 * int(tables.j_half_split)

    def message_bytes(self, n_reads: int, n_hits: int) -> int:
        return ((n_reads + 63) // 64) * 8 + n_hits * self.bytes

    def pack(self, rec, n_slots: int = None) -> np.ndarray:
        """The message of a batch's records (uint8): bitmap (over n_slots >= len(rec) read slots), low words, high bytes."""
        n_slots = len(rec) if n_slots is None else n_slots
        ok = rec["status"] == 0
        r = rec[ok]
        j = r["j"].astype(np.int64)
        tagpos = r["ins_start"].astype(np.int64) + r["ins_len"].astype(np.int64) - r["jdel"].astype(np.int64) + self.j_jumps[j]
        short_end = ((r["j_end"].astype(np.int64) - tagpos) != self.j_lens[j]).astype(np.uint64)
        fields = [r["v"], r["j"], r["vdel"], r["jdel"], r["v_start"], r["j_end"], short_end, r["frame"] & 1]
        t = np.zeros(len(r), dtype=np.uint64)
        sh = 0
        for f, w in zip(fields, self.widths):
            t |= f.astype(np.uint64) << np.uint64(sh)
            sh += w
        bits = np.zeros(((n_slots + 63) // 64) * 64, dtype=np.uint8)
        bits[:len(rec)] = ok
        bitmap = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
        lo = (t & np.uint64(0xFFFFFFFF)).astype("<u4").view(np.uint8)
        hi = (t >> np.uint64(32)).astype("<u8").view(np.uint8).reshape(-1, 8)[:, :self.bytes - 4].reshape(-1)
        return np.concatenate([bitmap, lo, hi])

    def unpack(self, message, n_reads: int, n_hits: int):
        """(records (status OK), read indices) of a message of n_hits tuples over n_reads reads."""
        m = np.ascontiguousarray(message, dtype=np.uint8)
        bm = ((n_reads + 63) // 64) * 8
        idx = np.nonzero(np.unpackbits(m[:bm], bitorder="little")[:n_reads])[0]
        t = m[bm:bm + 4 * n_hits].view("<u4").astype(np.uint64)
        hb = self.bytes - 4
        if hb:
            hi = np.zeros((n_hits, 8), dtype=np.uint8)
            hi[:, :hb] = m[bm + 4 * n_hits:bm + self.bytes * n_hits].reshape(n_hits, hb)
            t |= hi.view("<u8").reshape(-1) << np.uint64(32)
        vals = []
        sh = 0
        for w in self.widths:
            vals.append(((t >> np.uint64(sh)) & np.uint64((1 << w) - 1)).astype(np.int64))
            sh += w
        v, j, vdel, jdel, v_start, j_end, short_end, frame = vals
        rec = np.zeros(n_hits, dtype=RECORD_DTYPE)
        for name, x in (("v", v), ("j", j), ("vdel", vdel), ("jdel", jdel), ("v_start", v_start), ("j_end", j_end), ("frame", frame)):
            rec[name] = x
        ins_start = v_start + self.v_jumps[v] - vdel
        start_j = j_end - np.where(short_end == 1, self.j_short, self.j_lens[j]) - self.j_jumps[j] + jdel
        rec["ins_start"] = ins_start
        rec["ins_len"] = start_j - ins_start
        return rec, idx


def compact_hits_narrow_device(tables: "Tables", codec: TupleCodec, d_records_ptr: int, n_reads: int, d_message_ptr: int,
                               d_n_hits_ptr: int, stream=None, n_slots: int = None):
    check(lib().dcrx_compact_hits_narrow_device(tables.handle, C.byref(codec.layout), d_records_ptr, n_reads,
                                                n_reads if n_slots is None else n_slots, d_message_ptr, d_n_hits_ptr, stream))


def set_tuple_sink(tables: "Tables", codec, d_message_ptr: int = 0, n_slots: int = 0, d_n_hits_ptr: int = 0):
    """dcrx_set_tuple_sink: the next dcrx_decombine_device calls on `tables` also leave the batch's message; codec None: off."""
    if codec is None:
        check(lib().dcrx_set_tuple_sink(tables.handle, None, None, 0, None))
    else:
        check(lib().dcrx_set_tuple_sink(tables.handle, C.byref(codec.layout), d_message_ptr, n_slots, d_n_hits_ptr))


def unpack_tuples12(words: np.ndarray) -> np.ndarray:
    """(k, 3) uint32 tuples of dcrx_compact_hits_packed_device -> RECORD_DTYPE records (status OK)."""
    w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 3)
    rec = np.zeros(len(w), dtype=RECORD_DTYPE)
    rec["v"] = w[:, 0] & 0xFFF
    rec["j"] = (w[:, 0] >> 12) & 0xFFF
    rec["vdel"] = (w[:, 0] >> 24) & 0xFF
    rec["v_start"] = w[:, 1] & 0x1FF
    rec["j_end"] = (w[:, 1] >> 9) & 0x1FF
    rec["ins_start"] = (w[:, 1] >> 18) & 0x1FF
    rec["ins_len"] = w[:, 2] & 0x1FF
    rec["jdel"] = (w[:, 2] >> 9) & 0xFF
    rec["frame"] = (w[:, 2] >> 17) & 1
    return rec


def compact_hits_device(d_records: DeviceBuffer, n_reads: int, first_index: int, d_hits: DeviceBuffer,
                        d_hit_index: DeviceBuffer, d_n_hits: DeviceBuffer, stream=None):
    check(lib().dcrx_compact_hits_device(d_records.ptr, n_reads, first_index, d_hits.ptr, d_hit_index.ptr,
                                         d_n_hits.ptr, stream))
