"""The narrow tuple of a sharded run's gather (include/dcrx.h, dcrx_tuple_layout) on the host: the layout the library derives
from a tag set, and TupleCodec.pack / unpack against the oracle's records — every field of a decombined record comes back,
ins_start and ins_len included (they do not travel: decombine.py:283-285, :407-409, :450-454, :506-509)."""
import numpy as np
import pytest

from decombinator_amd import _native as nat, synth
from tests import golden_util as gu, parity_util as pu


def _tsd(ts):
    return dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
                j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])


def _bits(x):
    return max(1, int(x).bit_length())


@pytest.mark.parametrize("which", ["config2", "config5", "extended_alpha"])
def test_layout_follows_the_tables(which):
    ts = {"config2": lambda: synth.config_tagset(2), "config5": lambda: synth.config5_tagsets()[0],
          "extended_alpha": lambda: synth.config3_tagsets()[0]}[which]()
    t = pu.native_tables(_tsd(ts))
    c = nat.TupleCodec(t, 150)
    L = c.layout
    assert L.w_v == _bits(len(ts.v_tags) - 1) and L.w_j == _bits(len(ts.j_tags) - 1)
    assert L.w_vdel == _bits(max(j - len(tag) for j, tag in zip(ts.v_jumps, ts.v_tags)))
    assert L.w_jdel == _bits(max(ts.j_jumps)) and L.w_pos == 8
    assert L.bits == L.w_v + L.w_j + L.w_vdel + L.w_jdel + 2 * L.w_pos + 2 and L.bytes == max(4, (L.bits + 7) // 8)
    if which == "config2":
        assert (L.bits, L.bytes) == (39, 5)
    assert nat.lib().dcrx_tuple_message_bytes(L, 1000, 10) == 16 * 8 + 10 * L.bytes == c.message_bytes(1000, 10)
    assert nat.TupleCodec(t, 65535).layout.w_pos == 16


@pytest.mark.parametrize("sub_rate", [0.005, 0.04])
@pytest.mark.parametrize("config", [2, 5])
def test_round_trip_on_oracle_records(config, sub_rate):
    ts = synth.config_tagset(2) if config == 2 else synth.config5_tagsets()[1]
    tsd = _tsd(ts)
    t = pu.native_tables(tsd)
    ot = gu.oracle_tables(tsd)
    n = 40_000
    reads = nat.unpack_reads(nat.synth_reads_host(t, nat.synth_cfg(seed=11, sub_rate=sub_rate), 0, n))
    rec, _ = pu.oracle_records(ot, reads, "reverse", False, 130)
    ok = np.nonzero(rec["status"] == 0)[0]
    assert len(ok) > n // 5
    c = nat.TupleCodec(t, 150)
    msg = c.pack(rec)
    assert len(msg) == c.message_bytes(n, len(ok))
    back, idx = c.unpack(msg, n, len(ok))
    assert (idx == ok).all()
    assert back.tobytes() == rec[ok].tobytes()
    # the half1 rescue of J ends the read's J part at start + 2 * split, not at the tag's end: some tuples carry the bit
    if ts.half_splits[1] * 2 != len(ts.j_tags[0]):
        sh = sum(c.widths[:6])
        lo = msg[((n + 63) // 64) * 8:][:4 * len(ok)].view("<u4").astype(np.uint64)
        hi = msg[((n + 63) // 64) * 8 + 4 * len(ok):].reshape(-1, c.bytes - 4)[:, 0].astype(np.uint64)
        tup = lo | (hi << np.uint64(32))
        assert ((tup >> np.uint64(sh)) & np.uint64(1)).sum() > 0


def test_empty_and_all_failed_batches():
    ts = synth.config_tagset(2)
    t = pu.native_tables(_tsd(ts))
    c = nat.TupleCodec(t, 150)
    rec = np.zeros(130, dtype=nat.RECORD_DTYPE)
    rec["status"] = 3
    msg = c.pack(rec)
    assert len(msg) == 3 * 8 and not msg.any()
    back, idx = c.unpack(msg, 130, 0)
    assert len(back) == 0 and len(idx) == 0
    assert len(c.pack(rec[:0])) == 0


def test_tuple_sink_arguments_are_checked_without_a_gpu():
    """dcrx_set_tuple_sink validates before anything touches a device: a layout that is not the tables' own, null buffers;
    layout NULL turns the sink off."""
    import ctypes as C
    ts = synth.config_tagset(2)
    t = pu.native_tables(_tsd(ts))
    c = nat.TupleCodec(t, 150)
    lib = nat.lib()
    assert lib.dcrx_set_tuple_sink(t.handle, None, None, 0, None) == 0                    # off: always fine
    assert lib.dcrx_set_tuple_sink(t.handle, C.byref(c.layout), None, 100, None) != 0      # no buffers
    other = nat.TupleCodec(pu.native_tables(_tsd(synth.config5_tagsets()[0])), 150)
    assert (other.layout.w_v, other.layout.w_j) != (c.layout.w_v, c.layout.w_j)
    assert lib.dcrx_set_tuple_sink(t.handle, C.byref(other.layout), 8, 100, 8) != 0         # another tag set's layout
    assert b"layout" in lib.dcrx_last_error()
    bad = nat.TupleLayoutC.from_buffer_copy(bytes(c.layout))
    bad.bytes = 8
    assert lib.dcrx_set_tuple_sink(t.handle, C.byref(bad), 8, 100, 8) != 0
    assert lib.dcrx_set_tuple_sink(None, C.byref(c.layout), 8, 100, 8) != 0
