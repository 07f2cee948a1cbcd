"""ctypes front end of oracle/dcr_oracle.c (TEST INFRASTRUCTURE ONLY).

Builds oracle/build/libdcr_oracle.so with gcc on first use when it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libdcr_oracle.so")

N_COUNTERS = 32

# names in enum dcrx_counter order (include/dcrx_codes.h); the strings are the
# reference's Counter keys (decombine.py:598 ff.)
COUNTER_NAMES = [
    "multiple_v_matches", "verr2", "foundv1notv2", "verr1", "foundv2notv1",
    "no_vtags_found", "multiple_j_matches", "jerr2", "foundj1notj2", "jerr1",
    "no_j_assigned", "dcrfilter_intertagN", "dcrfilter_toolong_intertag",
    "dcrfilter_imposs_deletion", "dcrfilter_tag_overlap", "VJ_assignment_failed",
    "v_del_failed_tag_at_end", "v_del_failed", "j_del_failed", "vj_count",
    "read_count", "foundj2notj1", "frame_forward",
]


class Result(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "status", "frame", "v", "j", "vdel", "jdel", "ins_start", "ins_len", "v_start", "j_end")]


RESULT_DTYPE = np.dtype([(n, "<i4") for n, _ in Result._fields_])


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dcr_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "build/libdcr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.dcro_tables_new.restype = C.c_void_p
        L.dcro_tables_new.argtypes = [
            C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
            C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
            C.c_int, C.c_int]
        L.dcro_tables_free.argtypes = [C.c_void_p]
        L.dcro_revcomp.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.dcro_dcr.restype = C.c_int
        L.dcro_dcr.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                               C.POINTER(Result), C.POINTER(C.c_uint64)]
        L.dcro_decombine_read.restype = C.c_int
        L.dcro_decombine_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(Result), C.POINTER(C.c_uint64)]
        L.dcro_decombine_batch.restype = None
        L.dcro_decombine_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.dcro_decombine_batch_mt.restype = C.c_int
        L.dcro_decombine_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.dcro_findall.restype = C.c_int
        L.dcro_findall.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        _lib = L
    return _lib


def _strs(xs):
    arr = (C.c_char_p * max(1, len(xs)))()
    for i, x in enumerate(xs):
        arr[i] = x.encode("ascii") if isinstance(x, str) else bytes(x)
    return arr


def _ints(xs):
    return (C.c_int * max(1, len(xs)))(*[int(x) for x in xs])


class OracleTables:
    """The per-chain tables of import_tcr_info (decombine.py:593-746)."""

    def __init__(self, v_tags, v_jumps, v_regions, j_tags, j_jumps, j_regions,
                 v_half_split: int, j_half_split: int):
        self._h = lib().dcro_tables_new(
            len(v_tags), _strs(v_tags), _ints(v_jumps), _strs(v_regions),
            len(j_tags), _strs(j_tags), _ints(j_jumps), _strs(j_regions),
            int(v_half_split), int(j_half_split))
        if not self._h:
            raise MemoryError("dcro_tables_new failed")

    def close(self):
        if self._h:
            lib().dcro_tables_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dcr(self, read: str, allow_ns: bool = False, lenthreshold: int = 130, counts=None):
        """dcr(read) on the frame as given (decombine.py:534).  Returns (list-or-None, Result)."""
        if counts is None:
            counts = np.zeros(N_COUNTERS, dtype=np.uint64)
        b = read.encode("latin-1")
        res = Result()
        ok = lib().dcro_dcr(self._h, b, len(b), int(allow_ns), int(lenthreshold), C.byref(res),
                            counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        if ok:
            out = [res.v, res.j, res.vdel, res.jdel,
                   read[res.ins_start:res.ins_start + res.ins_len], res.v_start, res.j_end]
        else:
            out = None
        return out, res

    def decombine_read(self, vdj: str, orientation: int = 0, allow_ns: bool = False,
                       lenthreshold: int = 130, counts=None):
        if counts is None:
            counts = np.zeros(N_COUNTERS, dtype=np.uint64)
        b = vdj.encode("latin-1")
        res = Result()
        ok = lib().dcro_decombine_read(self._h, b, len(b), int(orientation), int(allow_ns),
                                       int(lenthreshold), C.byref(res),
                                       counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        return bool(ok), res

    def decombine_batch(self, ascii_buf: np.ndarray, offsets: np.ndarray, orientation: int = 0,
                        allow_ns: bool = False, lenthreshold: int = 130):
        """ascii_buf: uint8 concatenation of the reads; offsets: uint64[n+1].
        Returns (results structured array, counters uint64[32])."""
        ascii_buf = np.ascontiguousarray(ascii_buf, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        counts = np.zeros(N_COUNTERS, dtype=np.uint64)
        lib().dcro_decombine_batch(self._h, ascii_buf.ctypes.data, offsets.ctypes.data, n,
                                   int(orientation), int(allow_ns), int(lenthreshold),
                                   res.ctypes.data, counts.ctypes.data)
        return res, counts

    def decombine_batch_mt(self, ascii_buf: np.ndarray, offsets: np.ndarray, orientation: int = 0,
                           allow_ns: bool = False, lenthreshold: int = 130, n_threads: int = 0, passes: int = 1):
        """decombine_batch over POSIX threads inside the C library (n_threads 0 = every host core);
        passes > 1 repeats the work for timing."""
        import os
        ascii_buf = np.ascontiguousarray(ascii_buf, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        counts = np.zeros(N_COUNTERS, dtype=np.uint64)
        if n_threads <= 0:
            n_threads = os.cpu_count() or 1
        rc = lib().dcro_decombine_batch_mt(self._h, ascii_buf.ctypes.data, offsets.ctypes.data, n,
                                           int(orientation), int(allow_ns), int(lenthreshold),
                                           res.ctypes.data, counts.ctypes.data, int(n_threads), int(passes))
        if rc:
            raise MemoryError("dcro_decombine_batch_mt")
        return res, counts

    def findall(self, gene: int, which: int, text: str, cap: int = 4096):
        b = text.encode("latin-1")
        idx = (C.c_int * cap)()
        st = (C.c_int * cap)()
        n = lib().dcro_findall(self._h, gene, which, b, len(b), idx, st, cap)
        return [(idx[i], st[i]) for i in range(min(n, cap))]


def revcomp(s: str) -> str:
    b = s.encode("latin-1")
    out = C.create_string_buffer(len(b) + 1)
    lib().dcro_revcomp(b, len(b), out)
    return out.raw[:len(b)].decode("latin-1")
