"""Shared machinery of the parity tests.

Two backends run the SAME per-read device code:
  * "hip"  — libdcrx.so on the GPU through the C ABI (the product; `-m gpu` tests)
  * "emul" — tests/host_emul: a test-only host compile of the device functions
             (CPU; debugging and sanitizers; never shipped)
Both are compared with the golden vectors (captured from the reference) and with
the CPU oracle (oracle/dcr_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from collections import Counter, defaultdict

import numpy as np

from decombinator_amd import _native as nat
from oracle import oracle as orc
from tests import golden_util as gu

_EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul")
_emul = None


def emul_lib(asan: bool = False):
    global _emul
    name = "libdcrx_emul_asan.so" if asan else "libdcrx_emul.so"
    path = os.path.join(_EMUL_DIR, "build", name)
    subprocess.check_call(["make", "-C", _EMUL_DIR, f"build/{name}"], stdout=subprocess.DEVNULL)
    if asan:
        return path
    if _emul is None:
        L = C.CDLL(path)
        L.emul_decombine.restype = C.c_int
        L.emul_decombine.argtypes = [C.POINTER(nat.TagSetC), C.POINTER(nat.CfgC), C.POINTER(nat.BatchC),
                                     C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
        L.emul_v2_lean.restype = C.c_uint64
        L.emul_v2_reads.restype = C.c_uint64
        _emul = L
    return _emul


def tagset_c(ts: dict):
    keep = (nat._strs(ts["v_tags"]), (C.c_int32 * len(ts["v_jumps"]))(*ts["v_jumps"]), nat._strs(ts["v_regions"]),
            nat._strs(ts["j_tags"]), (C.c_int32 * len(ts["j_jumps"]))(*ts["j_jumps"]), nat._strs(ts["j_regions"]))
    t = nat.TagSetC(len(ts["v_tags"]), keep[0], keep[1], keep[2], len(ts["j_tags"]), keep[3], keep[4], keep[5],
                    ts["v_half_split"], ts["j_half_split"])
    return t, keep


def native_tables(ts: dict) -> nat.Tables:
    return nat.Tables(ts["v_tags"], ts["v_jumps"], ts["v_regions"], ts["j_tags"], ts["j_jumps"], ts["j_regions"],
                      ts["v_half_split"], ts["j_half_split"])


class Backend:
    """Runs batches for one tag set on "hip" or "emul"."""

    def __init__(self, kind: str, ts: dict):
        self.kind = kind
        self.ts = ts
        if kind == "hip":
            self.tables = native_tables(ts)
        else:
            self._tsc, self._keep = tagset_c(ts)
            emul_lib()

    def run(self, batch: nat.PackedBatch, orientation="reverse", allow_ns=False, lenthreshold=130, flags=0):
        if self.kind == "hip":
            return nat.decombine(self.tables, batch, orientation, allow_ns, lenthreshold, flags)
        cfg = nat.make_cfg(orientation, allow_ns, lenthreshold, flags)
        rec = np.zeros(batch.n_reads, dtype=nat.RECORD_DTYPE)
        cnt = np.zeros(nat.N_COUNTERS, dtype=np.uint64)
        b = batch.as_c()
        err = C.create_string_buffer(512)
        rc = emul_lib().emul_decombine(C.byref(self._tsc), C.byref(cfg), C.byref(b), rec.ctypes.data,
                                       cnt.ctypes.data, err, 512)
        if rc:
            raise nat.DcrxError(rc, err.value.decode())
        return rec, cnt


def record_to_expect(read_fastq: str, rec) -> list | None:
    """The reference's dcr() 7-list from a device record (None when not decombined)."""
    if int(rec["status"]) != 0:
        return None
    frame_read = read_fastq if int(rec["frame"]) == 1 else orc.revcomp(read_fastq)
    s, l = int(rec["ins_start"]), int(rec["ins_len"])
    return [int(rec["v"]), int(rec["j"]), int(rec["vdel"]), int(rec["jdel"]), frame_read[s:s + l],
            int(rec["v_start"]), int(rec["j_end"])]


def oracle_records(ot: orc.OracleTables, reads, orientation, allow_ns, lenthreshold):
    """Oracle results in the device record layout + counters."""
    bs = [r.encode("latin-1") for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)
    o = nat.ORIENTATIONS[orientation] if isinstance(orientation, str) else orientation
    res, cnt = ot.decombine_batch(buf, offsets, o, allow_ns, lenthreshold)
    return oracle_to_records(res), cnt


def oracle_to_records(res):
    rec = np.zeros(len(res), dtype=nat.RECORD_DTYPE)
    for f in ("v", "j", "v_start", "j_end", "ins_start", "ins_len", "vdel", "jdel", "status", "frame"):
        rec[f] = res[f]
    return rec


def assert_records_equal(got, want, reads=None, what=""):
    if got.tobytes() == want.tobytes():
        return
    bad = np.nonzero(got != want)[0]
    i = int(bad[0])
    msg = f"{what}: {len(bad)} of {len(got)} records differ; first at {i}: got {got[i]} want {want[i]}"
    if reads is not None:
        msg += f" read={reads[i]}"
    raise AssertionError(msg)


def assert_counters_equal(got, want, what=""):
    g = {n: int(got[i]) for i, n in enumerate(nat.COUNTER_NAMES) if int(got[i])}
    w = {n: int(want[i]) for i, n in enumerate(nat.COUNTER_NAMES) if int(want[i])}
    assert g == w, f"{what}: counters differ: got-want = " \
                   f"{ {k: (g.get(k, 0), w.get(k, 0)) for k in set(g) | set(w) if g.get(k, 0) != w.get(k, 0)} }"


def check_fixture(kind: str, path: str, flags: int = 0):
    """Golden fixture through a backend: per-case 7-list + frame, per-group counters
    against the reference's own numbers, and full records against the oracle."""
    fx = gu.load(path)
    ts = fx["tagset"]
    be = Backend(kind, ts)
    ot = gu.oracle_tables(ts)
    groups = defaultdict(list)
    for i, cs in enumerate(fx["cases"]):
        assert len(cs["read"]) <= 511, "a golden case beyond dcrx_tables_info.max_read_len: pin it as refused, do not skip it"
        groups[(cs["orientation"], cs["allowNs"], cs["lenthreshold"])].append(i)
    assert groups
    n_checked = 0
    for (orientation, allow_ns, lenthr), idxs in groups.items():
        reads = [fx["cases"][i]["read"] for i in idxs]
        batch = nat.pack_reads(reads)
        rec, cnt = be.run(batch, orientation, allow_ns, lenthr, flags)
        want_counts = Counter()
        for k, i in enumerate(idxs):
            cs = fx["cases"][i]
            got = record_to_expect(cs["read"], rec[k])
            assert got == cs["expect"], (kind, i, cs["label"], cs["read"], got, cs["expect"], rec[k])
            if cs["expect"] is not None:
                assert ("forward" if rec[k]["frame"] else "reverse") == cs["frame"], (i, cs["label"])
            want_counts.update(cs["counts"])
            n_checked += 1
        got_counts = {n: int(cnt[j]) for j, n in enumerate(nat.COUNTER_NAMES) if int(cnt[j]) and n != "frame_forward"}
        assert got_counts == dict(want_counts), (kind, orientation, allow_ns, lenthr,
                                                 {k: (got_counts.get(k, 0), want_counts.get(k, 0))
                                                  for k in set(got_counts) | set(want_counts)
                                                  if got_counts.get(k, 0) != want_counts.get(k, 0)})
        orec, ocnt = oracle_records(ot, reads, orientation, allow_ns, lenthr)
        assert_records_equal(rec, orec, reads, f"{kind} vs oracle {os.path.basename(path)}")
        assert_counters_equal(cnt, ocnt, f"{kind} vs oracle {os.path.basename(path)}")
    return n_checked
