"""The tuple sink (include/dcrx.h, dcrx_set_tuple_sink): while one is set, a dcrx_decombine_device call also leaves the batch's
message — bitmap | low words | high bytes of the narrow tuples, in read order.  Whatever path a read takes inside the call
(lean tail in the scan kernel's ring or as a role, lean rescue, general form over reads with exception bytes, left list,
hand-overs to the list kernel) and whatever the launch shape (the kernels' own items + the place kernel, or the compaction
of the records behind the call), the message must equal TupleCodec.pack of the call's records."""
import numpy as np
import pytest

from decombinator_amd import _native as nat, synth
from oracle import oracle as orc
from tests import parity_util as pu

pytestmark = pytest.mark.gpu


def _tables(ts):
    tsd = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
               j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
    return pu.native_tables(tsd)


def _run(t, db, n, orientation="reverse", flags=0, n_slots=None, codec=None, repeats=1, want_cnt=None):
    """One call with the sink set: (records, message bytes, count)."""
    n_slots = n if n_slots is None else n_slots
    codec = codec or nat.TupleCodec(t, db.read_len)
    d_rec = nat.DeviceBuffer(max(n, 1) * 16)
    d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)
    d_msg = nat.DeviceBuffer(codec.message_bytes(n_slots, n_slots) + 64)
    d_n = nat.DeviceBuffer(8)
    nat.set_tuple_sink(t, codec, d_msg.ptr, n_slots, d_n.ptr)
    try:
        for _ in range(repeats):
            nat.check(nat.lib().dcrx_memset_device(d_msg.ptr, 0xEE, codec.message_bytes(n_slots, n_slots)))
            nat.decombine_device(t, db, d_rec, d_cnt, orientation=orientation, flags=flags)
            nat.synchronize()
            rec = d_rec.to_host(nat.RECORD_DTYPE, n)
            k = int(d_n.to_host(np.uint64, 1)[0])
            assert k == int((rec["status"] == 0).sum())
            msg = d_msg.to_host(np.uint8, codec.message_bytes(n_slots, k))
            want = codec.pack(rec, n_slots=n_slots)
            assert msg.tobytes() == want.tobytes(), _first_difference(msg, want, n_slots, k, codec)
    finally:
        nat.set_tuple_sink(t, None)
    return rec, msg, k


def _first_difference(msg, want, n_slots, k, codec):
    bm = ((n_slots + 63) // 64) * 8
    d = np.nonzero(msg != want)[0]
    where = "bitmap" if d[0] < bm else ("low words" if d[0] < bm + 4 * k else "high bytes")
    return f"{len(d)} bytes differ, first at {int(d[0])} ({where}; bitmap {bm} bytes, {k} tuples of {codec.bytes} bytes)"


@pytest.mark.parametrize("flags", [0, nat.F_V2_NO_FUSE, nat.F_V2_SIDE_STREAMS, nat.F_V1_KERNELS],
                         ids=["fused-tail", "tail-as-a-role", "side-streams(compaction)", "three-launch(compaction)"])
def test_message_equals_the_records_config2(flags):
    ts = synth.config_tagset(2)
    t = _tables(ts)
    n = 1_500_000
    db = nat.synth_reads_device(t, nat.synth_cfg(seed=31, n_rate=0.002), 0, n)
    rec, msg, k = _run(t, db, n, flags=flags, repeats=3)
    assert k > n // 3
    # ... and what a receiver makes of it is the decombined records, read for read
    codec = nat.TupleCodec(t, 150)
    back, idx = codec.unpack(msg, n, k)
    ok = np.nonzero(rec["status"] == 0)[0]
    assert (idx == ok).all() and back.tobytes() == rec[ok].tobytes()


@pytest.mark.parametrize("which", [0, 1], ids=["alpha-extended", "beta-extended"])
def test_extended_sets_tail_as_a_role(which):
    ts = synth.config3_tagsets()[which]
    t = _tables(ts)
    n = 700_000
    db = nat.synth_reads_device(t, nat.synth_cfg(seed=5 + which, sub_rate=0.01), 0, n)
    _, _, k = _run(t, db, n, repeats=2)
    assert k > n // 5


def test_short_batches_more_slots_than_reads_and_reuse_of_the_handle():
    ts = synth.config_tagset(2)
    t = _tables(ts)
    big = nat.synth_reads_device(t, nat.synth_cfg(seed=9), 0, 300_000)
    _run(t, big, 300_000)
    for n in (100_000, 4097, 513, 64, 1):
        db = nat.synth_reads_device(t, nat.synth_cfg(seed=10 + n % 7, p_rearranged=0.8), 0, n)
        _run(t, db, n, n_slots=300_000)
    _run(t, big, 300_000)
    # sink off again: a call leaves the old buffers alone (and the records are what they were)
    d_rec = nat.DeviceBuffer(300_000 * 16)
    d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)
    nat.decombine_device(t, big, d_rec, d_cnt)
    nat.synchronize()


def test_exception_bytes_left_list_and_hand_overs():
    """Reads with clustered Ns (list X, general form), ragged short reads (the lean forms leave some to the left list) and reads
    with twelve tandem half tags (handed to the list kernel): every late path of the sink."""
    from tests import test_emul_parity as tep
    ts = synth.config_tagset(2)
    t = _tables(ts)
    _, reads = tep._n_clustered_reads(ts, 30_000, 23)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=3, p_rearranged=0.9, sub_rate=0.01, n_rate=0.001), 0, 120_000)
    more = nat.unpack_reads(hb)
    h = orc.revcomp(ts.j_tags[0][:6])
    more = [r[:20] + (h + "AC") * 12 + r[116:] if i % 3 == 0 else r for i, r in enumerate(more)]
    rng = np.random.default_rng(4)
    cut = rng.integers(40, 151, size=len(more))
    more = [r[:c] if i % 5 == 1 else r for i, (r, c) in enumerate(zip(more, cut))]
    reads = reads + more
    b = nat.pack_reads(reads, stride=40)
    db = nat.DeviceBatch.from_host(b)
    codec = nat.TupleCodec(t, 150)
    for orientation in ("reverse", "forward", "both"):
        rec, _, k = _run(t, db, len(reads), orientation=orientation, codec=codec, repeats=2)
        if orientation == "reverse":
            assert k > 30_000


def test_empty_batch_leaves_an_empty_message():
    ts = synth.config_tagset(2)
    t = _tables(ts)
    db = nat.synth_reads_device(t, nat.synth_cfg(seed=9), 0, 0)
    codec = nat.TupleCodec(t, 150)
    d_rec = nat.DeviceBuffer(16)
    d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)
    d_msg = nat.DeviceBuffer(codec.message_bytes(1000, 1000))
    d_n = nat.DeviceBuffer(8)
    nat.check(nat.lib().dcrx_memset_device(d_msg.ptr, 0xEE, codec.message_bytes(1000, 1000)))
    nat.set_tuple_sink(t, codec, d_msg.ptr, 1000, d_n.ptr)
    nat.decombine_device(t, db, d_rec, d_cnt)
    nat.synchronize()
    nat.set_tuple_sink(t, None)
    assert int(d_n.to_host(np.uint64, 1)[0]) == 0
    assert not d_msg.to_host(np.uint8, 16 * 8).any()


def test_regions_whose_tuples_do_not_fit_the_place_kernels_lds():
    """The place kernel stages a region's tuples in LDS for whole-line stores; a region with more tuples than fit (launches of
    100 M reads) writes each tuple to its place.  Forced here through DCRX_DEBUG_PLACE_HCAP (read once: a process of its own)."""
    import os
    import subprocess
    import sys
    code = r'''
from decombinator_amd import _native as nat, synth
from tests.test_gpu_sink import _tables, _run
ts = synth.config_tagset(2)
t = _tables(ts)
n = 2_000_000
db = nat.synth_reads_device(t, nat.synth_cfg(seed=61, p_rearranged=0.9), 0, n)
rec, msg, k = _run(t, db, n, repeats=2)
assert k > n // 2
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DCRX_DEBUG_FLAGS="1", DCRX_DEBUG_PLACE_HCAP="1000", PYTHONPATH=root), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])
