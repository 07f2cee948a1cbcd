"""Pins the CPU oracle (oracle/dcr_oracle.c) against the golden vectors that
oracle/gen_golden.py captured from the reference's own decombine.py: the dcr()
7-list (decombine.py:572-581), the frame, and the per-read Counter delta."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import golden_util as gu


@pytest.mark.parametrize("path", gu.golden_files(), ids=lambda p: p.split("/")[-1])
def test_oracle_matches_reference_vectors(path):
    fx = gu.load(path)
    ot = gu.oracle_tables(fx["tagset"])
    assert len(fx["cases"]) > 500
    for i, cs in enumerate(fx["cases"]):
        counts = np.zeros(orc.N_COUNTERS, dtype=np.uint64)
        ok, res = ot.decombine_read(cs["read"], gu.ORIENT[cs["orientation"]], cs["allowNs"],
                                    cs["lenthreshold"], counts)
        frame_read = cs["read"] if res.frame == 1 else orc.revcomp(cs["read"])
        got = gu.expect_from_result(frame_read, res)
        assert got == cs["expect"], (i, cs["label"], cs["read"])
        if cs["expect"] is not None:
            assert ("forward" if res.frame else "reverse") == cs["frame"], (i, cs["label"])
        assert gu.counts_dict(counts) == cs["counts"], (i, cs["label"], cs["read"])


def test_golden_covers_every_exit_path():
    """The fixtures must exercise every counter the reference can raise on this
    path (dcrfilter_tag_overlap is unreachable: DESIGN.md 'dead filter')."""
    seen = set()
    for path in gu.golden_files():
        for cs in gu.load(path)["cases"]:
            seen.update(cs["counts"])
    want = set(orc.COUNTER_NAMES) - {"frame_forward", "foundj2notj1", "dcrfilter_tag_overlap"}
    assert want <= seen, want - seen


def test_revcomp_table():
    assert orc.revcomp("ACGTN") == "NACGT"
    assert orc.revcomp("acgtn") == "nacgt"
    assert orc.revcomp("RYKMBVDHSWXU-.") == ".-AXWSDHBVKMRY"
    assert orc.revcomp("") == ""


def test_findall_contract():
    """acora contract as relied upon: overlaps reported, ordered
    by end position, unknown byte resets, duplicates collapse."""
    ot = orc.OracleTables(["AAAAAAAAAAAAAAAAAAAA", "ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGT"],
                          [40, 40, 40], ["A" * 60] * 3, ["CCCCCCCCCCGGGGGGGGGG"], [20], ["C" * 60], 10, 10)
    hits = ot.findall(0, 0, "A" * 22)
    assert hits == [(0, 0), (0, 1), (0, 2)]
    assert ot.findall(0, 0, "A" * 10 + "N" + "A" * 10) == []
    assert ot.findall(0, 0, "ACGT" * 6) == [(1, 0), (1, 4)]  # duplicate tag -> first index, once per occurrence
    # half1 automaton (10-mers): AAAAAAAAAA and ACGTACGTAC
    assert ot.findall(0, 1, "AAAAAAAAAAA") == [(0, 0), (0, 1)]
