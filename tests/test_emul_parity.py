"""CPU run of the device per-read code (tests/host_emul, a test-only host compile
of decombinator_amd/csrc/dcrx_dcr_device.h) against the golden vectors and the
oracle.  Debug/sanitizer aid; the parity tests proper are tests/test_gpu_parity.py."""
import pytest

from decombinator_amd import _native as nat
from tests import golden_util as gu
from tests import parity_util as pu


@pytest.mark.parametrize("path", gu.golden_files(), ids=lambda p: p.split("/")[-1])
@pytest.mark.parametrize("flags", [0, nat.F_ONE_BASE_SCAN, nat.F_FORCE_SLOW_READER, nat.F_LIST_RESCUE, nat.F_V2_NO_LEAN_RESCUE],
                         ids=["pairscan", "onebase", "slowreader", "listrescue", "general-form-only"])
def test_emul_matches_golden_and_oracle(path, flags):
    assert pu.check_fixture("emul", path, flags) > 500


def test_emul_synthetic_orientations_and_ragged_lengths():
    from tests import emul_extended
    assert emul_extended.run(60000)


@pytest.mark.parametrize("flags", [0, nat.F_LIST_RESCUE], ids=["rescue-form", "list-form"])
def test_emul_ragged_lengths_forward_and_reverse_with_many_tag_errors(flags):
    """Odd and even lengths in both frames with 3 % substitutions: the pair form of the rescue
    (left-over last base forward, lone first base reverse) against the oracle."""
    import numpy as np
    from decombinator_amd import synth
    from oracle import oracle as orc
    ts = synth.config_tagset(2)
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
    t = pu.native_tables(d)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], *ts.half_splits)
    be = pu.Backend("emul", d)
    rng = np.random.default_rng(13)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=19, read_len=160, sub_rate=0.03, n_rate=0.002), 0, 30_000, stride=40)
    reads = [orc.revcomp(r) for r in nat.unpack_reads(hb)]
    cut = rng.integers(0, 161, size=len(reads))
    reads = [r[:c] if i % 2 else r[len(r) - c:] for i, (r, c) in enumerate(zip(reads, cut))]
    b = nat.pack_reads(reads, stride=40)
    n_ok = 0
    for orientation in ("forward", "reverse"):
        rec, cnt = be.run(b, orientation, flags=flags)
        orec, ocnt = pu.oracle_records(ot, reads, orientation, False, 130)
        pu.assert_records_equal(rec, orec, reads, orientation)
        pu.assert_counters_equal(cnt, ocnt)
        n_ok += int((orec["status"] == 0).sum())
    assert n_ok > 1000


@pytest.mark.parametrize("tag_len", [21, 22], ids=["half15-pair-rescue", "half16-list-rescue"])
def test_emul_long_half_tags_at_the_window_limit(tag_len):
    """Half tags of 15 nt are the longest the rescue kernel's 16-base window serves; 16-nt ones go
    through the list form.  3 % substitutions so that many reads need the rescue."""
    _long_half_tags("emul", tag_len, 20_000)


def _long_half_tags(kind, tag_len, n):
    from decombinator_amd import synth
    from oracle import oracle as orc
    ts = synth.make_tagset("human", "original", "b", n_v=30, n_j=8, seed=77 + tag_len, tag_len=tag_len)
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
    t = pu.native_tables(d)
    assert t.info()["pair_scan_bytes"] > 0
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], *ts.half_splits)
    be = pu.Backend(kind, d)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=5, sub_rate=0.03, n_rate=0.002), 0, n)
    reads = nat.unpack_reads(hb)
    for orientation in ("reverse", "forward"):
        rs = reads if orientation == "reverse" else [orc.revcomp(r) for r in reads]
        b = nat.pack_reads(rs)
        rec, cnt = be.run(b, orientation)
        orec, ocnt = pu.oracle_records(ot, rs, orientation, False, 130)
        pu.assert_records_equal(rec, orec, rs, orientation)
        pu.assert_counters_equal(cnt, ocnt)
    rescued = int(ocnt[nat.COUNTER_NAMES.index("verr1")] + ocnt[nat.COUNTER_NAMES.index("verr2")]) if "verr1" in nat.COUNTER_NAMES else 1
    assert rescued > 0


def test_emul_randomised_configurations():
    from tests import fuzz_util
    assert fuzz_util.run("emul", 16, 3000, seed=424242) > 1000


def _long_reads(kind, n):
    """Reads of 321..511 nt (merged 2x250 amplicons): the v2 kernels' third register shape (32 words per read, one read per
    lane); with DCRX_F_V1_KERNELS the three-launch form's list kernel, which walks the packed words in memory."""
    import numpy as np
    from decombinator_amd import synth
    from oracle import oracle as orc
    ts = synth.config_tagset(2)
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
    t = pu.native_tables(d)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], *ts.half_splits)
    be = pu.Backend(kind, d)
    rng = np.random.default_rng(5)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=77, read_len=300, sub_rate=0.01, n_rate=0.002), 0, n, stride=nat.stride_for(300))
    reads = nat.unpack_reads(hb)
    # pad to 321..511 nt with random flanks on either side (the rearrangement sits anywhere inside)
    out = []
    for i, r in enumerate(reads):
        extra = int(rng.integers(21, 212))
        left = int(rng.integers(0, extra + 1))
        fl = "".join("ACGT"[k] for k in rng.integers(0, 4, size=extra))
        out.append(fl[:left] + r + fl[left:])
    assert 321 <= min(map(len, out)) and max(map(len, out)) <= 511
    b = nat.pack_reads(out, stride=128)
    for orientation, flags in (("reverse", 0), ("forward", 0), ("both", 0), ("reverse", nat.F_V1_KERNELS)):
        rec, cnt = be.run(b, orientation, flags=flags)
        orec, ocnt = pu.oracle_records(ot, out, orientation, False, 130)
        pu.assert_records_equal(rec, orec, out, orientation)
        pu.assert_counters_equal(cnt, ocnt)
    assert int((orec["status"] == 0).sum()) >= 0
    return int(ocnt[nat.COUNTER_NAMES.index("read_count")])


def test_emul_reads_of_321_to_511_nt():
    assert _long_reads("emul", 3000) == 3000


def _reads_of_512_nt_and_more(kind, n, which=2, exc_share=0.02):
    """Reads of 512 nt and more (strides beyond 128 bytes): the long form — a two-pass scan with the one-base table (whole words
    with nothing but the OR of the entries, then the flagged words base by base), hit lists with plain-integer positions,
    dcr_frame with the rescue from the lists.  Real rearrangements in random flanks, substitutions, exception bytes, both
    strands; a uniform batch of 600 nt and a ragged one (512 .. 5 000 nt), the three orientations."""
    import random
    from decombinator_amd import synth
    from oracle import oracle as orc
    ts = synth.config_tagset(2) if which == 2 else synth.config3_tagsets()[0]
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
    t = pu.native_tables(d)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], *ts.half_splits)
    be = pu.Backend(kind, d)
    rng = random.Random(11)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=78, p_rearranged=0.8, sub_rate=0.01, n_rate=0.002), 0, 2 * n)
    cores = nat.unpack_reads(hb)
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))

    def lengthen(r, m):
        a = rng.randrange(0, m - len(r) + 1)
        s = rnd(a) + r + rnd(m - len(r) - a)
        if rng.random() < exc_share:      # a run of exception bytes and a lone one somewhere in the flanks or the rearrangement
            k = rng.randrange(0, m - 8)
            s = s[:k] + "NNN" + s[k + 3:]
        for _ in range(3 if exc_share > 0.1 else 1):
            if rng.random() < exc_share:
                k = rng.randrange(0, m)
                s = s[:k] + rng.choice("NRY") + s[k + 1:]
        return orc.revcomp(s) if rng.random() < 0.3 else s
    uniform = [lengthen(r, 600) for r in cores[:n]]
    ragged = [lengthen(r, rng.choice([512, 513, 527, 528, 529, 600, 777, 1500, 5000])) for r in cores[n:2 * n]]
    n_ok = 0
    for reads in (uniform, ragged):
        b = nat.pack_reads(reads)
        assert b.stride > 128
        for orientation in ("reverse", "forward", "both"):
            rec, cnt = be.run(b, orientation, flags=0)
            orec, ocnt = pu.oracle_records(ot, reads, orientation, False, 130)
            pu.assert_records_equal(rec, orec, reads, "long form, " + orientation)
            pu.assert_counters_equal(cnt, ocnt)
            n_ok += int((orec["status"] == 0).sum())
    return n_ok


def test_emul_reads_of_512_nt_and_more():
    assert _reads_of_512_nt_and_more("emul", 1500) > 1500
    assert _reads_of_512_nt_and_more("emul", 400, which=3) > 400
    # half of the reads with exception bytes (words that go base by base in the first pass, resets of the machine in both)
    assert _reads_of_512_nt_and_more("emul", 600, exc_share=0.5) > 300


def _synthetic_vs_oracle(kind, ts, n, orientation, flags, forward_strand=False, **synth_kw):
    """`n` synthetic reads of tag set `ts` through backend `kind` against the oracle: records and counters.  The
    generator writes the reverse strand; forward_strand hands the backend the reverse complements instead."""
    from oracle import oracle as orc
    vs, js = ts.half_splits
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=vs, j_half_split=js)
    t = pu.native_tables(d)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    hb = nat.synth_reads_host(t, nat.synth_cfg(**synth_kw), 0, n)
    reads = nat.unpack_reads(hb)
    if forward_strand:
        reads = [orc.revcomp(r) for r in reads]
        hb = nat.pack_reads(reads)
    rec, cnt = pu.Backend(kind, d).run(hb, orientation, flags=flags)
    orec, ocnt = pu.oracle_records(ot, reads, orientation, False, 130)
    pu.assert_records_equal(rec, orec, reads, orientation)
    pu.assert_counters_equal(cnt, ocnt)
    return int((orec["status"] == 0).sum())


_LEAN_CASES = [  # (substitution rate, exception-byte rate, read length): rescue-heavy, exception-heavy, odd, short, long, hopeless
    (0.03, 0.0005, 150), (0.10, 0.02, 150), (0.05, 0.05, 101), (0.04, 0.01, 75), (0.06, 0.01, 300), (0.15, 0.0, 60)]


@pytest.mark.parametrize("flags", [0, nat.F_V2_NO_LEAN_RESCUE, nat.F_V1_KERNELS], ids=["lean-rescue", "general-form", "three-launch"])
@pytest.mark.parametrize("chain", ["beta-original", "alpha-extended", "delta-original"])
def test_emul_rescue_forms_on_both_strands(chain, flags):
    """The lean rescue (straight-line half-tag rescue, up to four flagged pairs per gene), the general form and the
    three-launch form on reads that need the rescue often, carry exception bytes, end in half pairs, are too short
    for a 32-base window or long enough for several: each against the oracle, the reverse frame on the generator's
    strand and the forward frame on its reverse complement.  The walks' window-by-window form (read edge, exception
    bytes as mismatches) is what the general form runs here."""
    from decombinator_amd import synth
    ts = {"beta-original": synth.config_tagset(2), "alpha-extended": synth.config3_tagsets()[0],
          "delta-original": synth.config5_tagsets()[1]}[chain]
    n_ok = 0
    for k, (sub, nrate, length) in enumerate(_LEAN_CASES):
        n_ok += _synthetic_vs_oracle("emul", ts, 2500, "reverse", flags, seed=60 + k, sub_rate=sub, n_rate=nrate, read_len=length)
        n_ok += _synthetic_vs_oracle("emul", ts, 2500, "forward", flags, forward_strand=True, seed=80 + k, sub_rate=sub,
                                     n_rate=nrate, read_len=length)
    assert n_ok > 2000


def test_lean_forms_settle_nearly_all_reads_of_the_bench_workload():
    """Guards the kernels' cost model, not their results: on BASELINE config 2's reads the lean tail settles its entries
    but for a few in 10 000 and the lean rescue more than nine event entries in ten — the general form (the event
    kernel) is priced for a handful of reads per thousand.  (Counted by the host emulation, which runs the same code.)"""
    import ctypes as C
    from decombinator_amd import synth
    ts = synth.config_tagset(2)
    n = 200_000
    stats = (C.c_uint64 * 64)()
    pu.emul_lib().emul_v2_stats(stats)          # (the tallies are cumulative: reading them clears them)
    pu.emul_lib().emul_v2_lean()
    assert _synthetic_vs_oracle("emul", ts, n, "reverse", 0, seed=2, sub_rate=0.005, n_rate=0.0005) > 0.3 * n
    pu.emul_lib().emul_v2_stats(stats)
    none, multi, tail, events = (int(stats[k]) for k in range(4))
    assert none + multi + tail + events == n
    assert 0.30 * n < tail < 0.40 * n and 0.08 * n < events < 0.14 * n
    lean_tail = int(pu.emul_lib().emul_v2_lean())
    lean_rescue = int(stats[62])
    assert lean_tail > 0.999 * tail, (lean_tail, tail)
    assert lean_rescue > 0.93 * events, (lean_rescue, events)


def _n_clustered_reads(ts, n, seed):
    """Reads with CLUSTERED exception bytes, as sequencers write them: N tails and N heads of 1-40 nt, stretches of Ns inside the
    read (also across a tag or a junction), whole reads of Ns, and — beside such a run — up to five single exception bytes
    (N, IUPAC codes, lower case), so that both sides of the register frame's limit are met."""
    import random
    from oracle import oracle as orc
    vs, js = ts.half_splits
    d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
             j_regions=ts.j_regions, v_half_split=vs, j_half_split=js)
    t = pu.native_tables(d)
    reads = nat.unpack_reads(nat.synth_reads_host(t, nat.synth_cfg(seed=seed, p_rearranged=0.85, sub_rate=0.01, n_rate=0.0), 0, n))
    rng = random.Random(seed)
    out = []
    for r in reads:
        s = list(r)
        kind = rng.random()
        if kind < 0.3:
            k = rng.randrange(1, 41); s[len(s) - k:] = "N" * k
        elif kind < 0.5:
            k = rng.randrange(1, 41); s[:k] = "N" * k
        elif kind < 0.8:
            k = rng.randrange(2, 30); a = rng.randrange(0, len(s) - k); s[a:a + k] = "N" * k
        elif kind < 0.85:
            s = list("N" * len(s))
        for _ in range(rng.choice([0, 0, 1, 2, 3, 4, 5])):
            s[rng.randrange(len(s))] = rng.choice("NNNRYacgtn")
        out.append("".join(s))
    return d, out


@pytest.mark.parametrize("orientation", ["reverse", "forward", "both"])
def test_clustered_exception_bytes_through_the_register_frame(orientation):
    """N tails, N heads, stretches of Ns and all-N reads — the register frame holds a run of Ns as a range
    beside four single bytes (FrameReg, exc_layout) instead of leaving for the list kernel at the fifth exception byte — on the
    host emulation of the same device functions, records and counters against the oracle; allowNs on and off (the inter-tag N
    filter, decombine.py:553-556, sees the run)."""
    from oracle import oracle as orc
    from decombinator_amd import synth
    ts = synth.config_tagset(2)
    d, reads = _n_clustered_reads(ts, 6000, 91)
    vs, js = ts.half_splits
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    if orientation != "reverse":
        reads = [orc.revcomp(r) if k % 2 else r for k, r in enumerate(reads)]
    hb = nat.pack_reads(reads)
    for allow in (False, True):
        rec, cnt = pu.Backend("emul", d).run(hb, orientation, allow_ns=allow)
        orec, ocnt = pu.oracle_records(ot, reads, orientation, allow, 130)
        pu.assert_records_equal(rec, orec, reads, orientation)
        pu.assert_counters_equal(cnt, ocnt)
    assert int((orec["status"] == 0).sum()) > 800
