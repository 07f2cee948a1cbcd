"""The per-row front half of `collapse` (decombinator_amd/collapse.py) against (1) the reference's own test
cases for it (reference tests/test_collapse.py:55-195: TestGetBarcodePositions, TestFindFirstSpacer), restated
here with the same inputs and expected values, and (2) tests/golden/collapse_front.json, generated from the
imported reference by oracle/gen_collapse_golden.py (1 500 barcode regions over the five oligos with
substituted / inserted / deleted / truncated spacers, Ns, every N1 length; plus the row loop of
read_in_data up to where grouping starts)."""
import collections as coll
import json
import os

import pytest

from decombinator_amd import collapse

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collapse_front.json")


@pytest.mark.parametrize("oligo,bcseq,want", [
    ("m13", "GTCGTGACTGGGAAAACCCTGGTTTCCGGTCGTGATAAAGTG", [22, 28, 36, 42]),
    ("i8", "GTCGTGATTTTCCGGTCGTGATAAAGTG", [8, 14, 22, 28]),
    ("i8_single", "GAAGCTATCACGACATCACTAC", [0, 6, 14, 20]),
    ("nebio", "CGGGCTTGGTATCGGCCGATCTACGGG", [0, 17]),
    ("takara", "CTCGTTAGGTTCGTACGGGGATTGCA", [0, 12]),
])
def test_get_barcode_positions_reference_cases(oligo, bcseq, want):
    """reference tests/test_collapse.py:55-125"""
    assert collapse.get_barcode_positions(bcseq, {"oligo": oligo, "allowNs": False}, coll.Counter()) == want


@pytest.mark.parametrize("spcr1,seq,start,end", [
    ("GTCGTGACTGGGAAAACCCTGG", "GTCGTGACTGGGAAAACCCTGGTTTCCGGTCGTGATAAAGTG", 0, 32),
    ("GTCGTGAT", "GTCGTGATTTTCCGGTCGTGATAAAGTG", 0, 18),
    ("ATCACGAC", "GAAGCTATCACGACATCACTAC", 0, 18),
    ("TACGGG", "CGGGCTTGGTATCGGCCGATCTACGGG", 18, 28),
    ("GTACGGG", "CTCGTTAGGTTCGTACGGGGATTGCA", 0, 19),
])
def test_find_first_spacer_reference_cases(spcr1, seq, start, end):
    """reference tests/test_collapse.py:128-195"""
    assert collapse.findFirstSpacer({"spcr1": spcr1}, seq, start, end) == [spcr1]


def test_golden_cases_positions_barcodes_quality_and_counters():
    fx = json.load(open(GOLDEN))
    params = fx["params"]
    n_fuzzy = 0
    for cs in fx["cases"]:
        args = {"oligo": cs["oligo"], "allowNs": cs["allowNs"]}
        c = coll.Counter()
        locs = collapse.get_barcode_positions(cs["bcseq"], args, c)
        assert locs == cs["locs"], cs
        assert dict(c) == cs["counts"], cs
        n_fuzzy += c["getbarcode_pass_regexmatch"]
        if locs:
            collapse.counts = coll.Counter()
            fields = ["1", "2", "3", "4", "ACGT", "id", "SEQ", "QUAL", cs["bcseq"], cs["bcqual"]]
            bc, bq = collapse.set_barcode(fields, locs, args)
            assert (bc, bq) == (cs["barcode"], cs["barcode_qual"]), cs
            assert dict(collapse.counts) == cs["set_counts"], cs
            if bq:
                assert bool(collapse.check_umi_quality(bq, params)) == cs["low_quality"], cs
    assert len(fx["cases"]) >= 1500 and n_fuzzy > 100


def test_golden_row_loop_matches_read_in_data_front():
    fx = json.load(open(GOLDEN))
    rows = None
    ok = 0
    for blk in fx["read_in"]:
        rows = blk["rows"] or rows
        args = {"oligo": blk["oligo"], "allowNs": blk["allowNs"], "lenthreshold": 130}
        keep = [i for i, e in enumerate(blk["expect"]) if e != "CRASH"]     # rows the reference itself does not survive
        collapse.counts = coll.Counter()
        got = collapse.read_in_rows([rows[i] for i in keep], args, fx["params"])
        want = [blk["expect"][i] for i in keep]
        assert [None if g is None else [g[0], g[1], g[2], g[3], g[4], g[5]] for g in got] == want, blk["oligo"]
        # the reference's counters over the same rows (its crash rows had counted input and position keys already)
        crash = len(blk["expect"]) - len(keep)
        cnt = dict(collapse.counts)
        if crash == 0:
            assert cnt == blk["counts"], (blk["oligo"], blk["allowNs"])
        ok += sum(1 for g in got if g is not None)
    assert ok > 150


def test_lines_and_lists_are_the_same_rows():
    fx = json.load(open(GOLDEN))
    rows = fx["read_in"][0]["rows"][:50]
    args = {"oligo": "m13", "allowNs": False, "lenthreshold": 130}
    collapse.counts = coll.Counter()
    a = collapse.read_in_rows(rows, args, fx["params"])
    collapse.counts = coll.Counter()
    b = collapse.read_in_rows([", ".join(r) + "\n" for r in rows], args, fx["params"])
    assert a == b
