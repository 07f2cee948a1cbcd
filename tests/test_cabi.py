"""The C-ABI shared library: loads, exports every symbol include/*.h declares,
compiles tables, packs reads and generates synthetic reads on the host, and
refuses to decombine without a GPU (no CPU fallback).  No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from decombinator_amd import _native as nat
from decombinator_amd import synth
from oracle import oracle as orc
from tests import golden_util as gu
from tests import parity_util as pu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("dcrx.h", "dcrx_synth.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(dcrx_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    L = C.CDLL(nat.LIB_PATH)
    decl = _declared()
    assert decl == set(nat.EXPORTS), decl ^ set(nat.EXPORTS)
    for name in decl:
        assert hasattr(L, name), name
    assert nat.lib().dcrx_abi_version() == nat.ABI_VERSION
    assert b"gfx950" in nat.lib().dcrx_build_info()


def test_record_layout_is_16_bytes():
    assert nat.RECORD_DTYPE.itemsize == 16
    assert [nat.RECORD_DTYPE.fields[n][1] for n in ("v", "j", "v_start", "j_end", "ins_start", "ins_len",
                                                    "vdel", "jdel", "status", "frame")] == \
        [0, 2, 4, 6, 8, 10, 12, 13, 14, 15]


def test_counter_names_match_oracle_numbering():
    assert nat.COUNTER_NAMES == orc.COUNTER_NAMES
    hdr = open(os.path.join(ROOT, "include", "dcrx_codes.h")).read()
    for i, n in enumerate(nat.COUNTER_NAMES):
        m = re.search(r"DCRX_C_%s = (\d+)" % n.upper(), hdr)
        assert m and int(m.group(1)) == i, n
    for i, n in enumerate(nat.STATUS_NAMES):
        m = re.search(r"DCRX_S_%s = (\d+)" % n, hdr)
        assert m and int(m.group(1)) == i, n


def test_tables_compile_and_info():
    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    inf = t.info()
    assert inf["n_v"] == 60 and inf["n_j"] == 13
    assert inf["n_keywords"][0] == 60 and inf["n_keywords"][1] == 13
    assert inf["n_keywords"][2] < 60  # shared half tags collapse
    assert inf["tables_in_lds"] and inf["dfa_bytes"] == inf["n_states"] * 16 < 64 * 1024
    assert inf["equal_len_per_automaton"]
    t.close()


@pytest.mark.parametrize("mutate,msg", [
    (lambda d: d["v_tags"].__setitem__(0, "ACGTNACGTACGTACGTACG"), "outside ACGT"),
    (lambda d: d["v_tags"].__setitem__(0, "ACGT" * 9), "length"),
    (lambda d: d["j_tags"].__setitem__(0, "ACGTA"), "half"),
    (lambda d: d["v_jumps"].__setitem__(0, 40000), "jump"),
    (lambda d: d["j_jumps"].__setitem__(0, 300), "J jump"),
])
def test_unsupported_tag_sets_are_refused(mutate, msg):
    fx = gu.load(gu.golden_files()[0])["tagset"]
    mutate(fx)
    with pytest.raises(nat.DcrxError) as e:
        pu.native_tables(fx)
    assert e.value.code == -2 and msg in str(e.value)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(5)
    reads = []
    for L in (0, 1, 3, 4, 5, 16, 17, 31, 32, 33, 150, 150, 151, 300, 320):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L))
        reads.append(s)
    reads[5] = reads[5][:3] + "N" + reads[5][4:]
    reads[10] = "n" + reads[10][1:75] + "RYK" + reads[10][78:]
    b = nat.pack_reads(reads)
    assert b.stride == 80 and b.lens is not None and len(b.exc_read) == 5
    assert list(b.exc_read) == [5, 10, 10, 10, 10] and list(b.exc_pos) == [3, 0, 75, 76, 77]
    assert nat.unpack_reads(b) == reads
    # packing convention: base i in bits 2(i%4) of byte i/4, A0 C1 G2 T3
    b2 = nat.pack_reads(["ACGTTGCA"])
    assert b2.read_len == 8 and b2.lens is None
    assert list(b2.packed[0, :2]) == [0b11100100, 0b00011011]
    with pytest.raises(nat.DcrxError):
        nat.pack_reads(["A" * 400], stride=80)


def test_synth_host_is_deterministic_and_rearranged_reads_decombine():
    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    cfg = nat.synth_cfg(seed=2, n_rate=0.01)
    a = nat.synth_reads_host(t, cfg, 1000, 4000)
    b = nat.synth_reads_host(t, cfg, 3000, 1000)
    assert (a.packed[2000:3000] == b.packed).all()  # any shard reproducible from (seed, index)
    assert 10 < len(a.exc_read) < 100
    reads = nat.unpack_reads(a)
    assert all(len(r) == 150 for r in reads)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    rec, cnt = pu.oracle_records(ot, reads, "reverse", False, 130)
    frac = int(cnt[nat.COUNTER_NAMES.index("vj_count")]) / len(reads)
    assert 0.35 < frac < 0.46, frac           # 45 % rearranged, a few lost to substitutions
    assert int(cnt[nat.COUNTER_NAMES.index("verr1")]) + int(cnt[nat.COUNTER_NAMES.index("verr2")]) > 20


def test_decombine_without_gpu_fails_loudly():
    if nat.device_count() > 0:
        pytest.skip("a GPU is present")
    fx = gu.load(gu.golden_files()[0])["tagset"]
    t = pu.native_tables(fx)
    with pytest.raises(nat.DcrxError) as e:
        nat.decombine(t, nat.pack_reads(["ACGT" * 30]))
    assert e.value.code in (-4, -5)


def test_bad_arguments():
    fx = gu.load(gu.golden_files()[0])["tagset"]
    t = pu.native_tables(fx)
    b = nat.pack_reads(["ACGT" * 30])
    b.stride = 12
    with pytest.raises(nat.DcrxError) as e:
        nat.decombine(t, b)
    assert e.value.code == -1


def test_pair_scan_table_is_built_only_when_safe():
    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    inf = t.info()
    assert inf["pair_scan_bytes"] == inf["n_states"] * 64
    # a tag that overlaps itself at shift 2 could occur 64+ times in a read: the 6-bit hit count of
    # the pair scan could wrap, so the set must keep the one-base scan
    vt = list(ts.v_tags)
    vt[0] = "ACACACACACACACACACAC"
    t2 = nat.Tables(vt, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    assert t2.info()["pair_scan_bytes"] == 0


def test_multi_gpu_entries_say_no_without_a_gpu_and_never_abort():
    """The RCCL section of the ABI on a box without a GPU: the library opens librccl only when a communicator is asked for, a
    failing call comes back as an error code with RCCL's text, bad arguments are refused before RCCL is touched — nothing aborts the
    process, nothing falls back to anything.  (With a GPU: tests/test_gpu_parity.py runs one rank through RCCL proper.)"""
    if nat.device_count() > 0:
        pytest.skip("a GPU is present: covered by the GPU tests")
    assert nat.lib().dcrx_comm_available() in (0, 1)
    with pytest.raises(nat.DcrxError):
        nat.Comm(nat.Comm.unique_id(), 1, 0)
    h = C.c_void_p()
    uid = np.zeros(nat.COMM_ID_BYTES, dtype=np.uint8)
    assert nat.lib().dcrx_comm_create(uid.ctypes.data, 2, 2, C.byref(h)) == -1 and not h.value       # rank outside the world
    assert nat.lib().dcrx_comm_create(None, 1, 0, C.byref(h)) == -1
    assert nat.lib().dcrx_comm_allreduce_u64(None, None, None, 1, 0, None) == -1
    assert nat.lib().dcrx_comm_gather_v(None, None, 0, None, None, 0, None) == -1
    assert nat.lib().dcrx_decombine_sharded(None, None, None, None, None, None, None, None, 0, None, None, None) == -1
    nat.lib().dcrx_comm_destroy(None)                                                                 # a no-op


def test_comm_from_env_carries_the_id_through_a_file(tmp_path, monkeypatch):
    """comm_from_env's rendezvous without RCCL: rank 0 leaves the id in a file (written under another name and renamed), the
    others read it — here with Comm replaced by a recorder, three 'ranks' in turn."""
    made = []

    class FakeComm:
        def __init__(self, uid, world, rank):
            made.append((bytes(uid), world, rank))

        @staticmethod
        def unique_id():
            return bytes(range(128))
    monkeypatch.setattr(nat, "Comm", FakeComm)
    path = tmp_path / "id"
    monkeypatch.setenv("DCRX_COMM_ID_FILE", str(path))
    monkeypatch.setenv("WORLD_SIZE", "3")
    for rank in (1, 2):                 # the others first: they must wait for the file, not read half of it
        monkeypatch.setenv("RANK", str(rank))
        with pytest.raises(TimeoutError):
            nat.comm_from_env(timeout_s=0.05)
    path.write_bytes(b"short")          # (a half-written file is not an id)
    monkeypatch.setenv("RANK", "1")
    with pytest.raises(TimeoutError):
        nat.comm_from_env(timeout_s=0.05)
    path.unlink()
    # rank 0 writes, then — in a real run once every rank has joined — removes the file: here the removal is rank 0's last act,
    # so the others are served from a copy made in between
    monkeypatch.setenv("RANK", "0")
    import shutil
    orig_remove = os.remove
    monkeypatch.setattr(os, "remove", lambda p: (shutil.copy(p, str(path) + ".kept"), orig_remove(p)))
    nat.comm_from_env()
    monkeypatch.setattr(os, "remove", orig_remove)
    assert not path.exists()
    os.replace(str(path) + ".kept", str(path))
    for rank in (1, 2):
        monkeypatch.setenv("RANK", str(rank))
        nat.comm_from_env(timeout_s=1.0)
    assert made == [(bytes(range(128)), 3, 0), (bytes(range(128)), 3, 1), (bytes(range(128)), 3, 2)]
