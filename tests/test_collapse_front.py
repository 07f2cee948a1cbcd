"""The front half of `collapse` (decombinator_amd/collapse.py over libdcrx: dcrx_collapse_front, dcrx_spacer_search — every
spacer search decided natively, the indel form included) against (1) the reference's own test cases for it (reference
tests/test_collapse.py:55-195: TestGetBarcodePositions, TestFindFirstSpacer), restated here with the same inputs and expected
values, (2) tests/golden/collapse_front.json, generated from the imported reference by oracle/gen_collapse_golden.py (1 500
barcode regions over the five oligos with substituted / inserted / deleted / truncated spacers, Ns, every N1 length; plus the
row loop of read_in_data up to where grouping starts), and (3) the reference's own `regex` patterns
(tests/collapse_regex_ref.py, itself checked against that fixture here) on mutated barcode regions the fixture does not hold."""
import collections as coll
import json
import os

import pytest

from decombinator_amd import collapse
from tests import collapse_regex_ref as ref

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
    """Every fixture case through the native get_barcode_positions (positions and getbarcode_* counters) and through the regex
    checker (positions, counters, set_barcode, check_umi_quality): the checker is pinned here before it serves as an oracle."""
    fx = json.load(open(GOLDEN))
    params = fx["params"]
    n_fuzzy = 0
    for cs in fx["cases"]:
        args = {"oligo": cs["oligo"], "allowNs": cs["allowNs"]}
        c = coll.Counter()
        locs = collapse.get_barcode_positions(cs["bcseq"], args, c)
        assert locs == cs["locs"], cs
        assert dict(c) == cs["counts"], cs
        c2 = coll.Counter()
        assert ref.get_barcode_positions(cs["bcseq"], args, c2) == cs["locs"] and dict(c2) == cs["counts"], cs
        n_fuzzy += c["getbarcode_pass_regexmatch"]
        if locs:
            ref.counts = coll.Counter()
            fields = ["1", "2", "3", "4", "ACGT", "id", "SEQ", "QUAL", cs["bcseq"], cs["bcqual"]]
            bc, bq = ref.set_barcode(fields, locs, args)
            assert (bc, bq) == (cs["barcode"], cs["barcode_qual"]), cs
            assert dict(ref.counts) == cs["set_counts"], cs
            if bq:
                assert bool(ref.check_umi_quality(bq, params)) == cs["low_quality"], cs
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


def _random_rows(rng, oligo, n):
    """Rows whose barcode regions carry the oligo's spacers verbatim, substituted, with indels, truncated, doubled or
    absent, random N1 lengths, Ns, qualities around the thresholds, inter-tag lengths around the length threshold."""
    o = collapse.getOligo(oligo)
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))

    def mutate(s):
        s = list(s)
        kind = rng.random()
        if kind < 0.45:
            return "".join(s)
        for _ in range(rng.choice([1, 1, 2, 2, 3])):
            op = rng.random()
            if op < 0.7 and s:
                i = rng.randrange(len(s)); s[i] = rng.choice([c for c in "ACGT" if c != s[i]])
            elif op < 0.85:
                s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
            elif s:
                del s[rng.randrange(len(s))]
        return "".join(s)

    rows = []
    for k in range(n):
        s1 = mutate(o["spcr1"])
        n1 = rnd(rng.choice([6, 6, 6, 6, 5, 7, 4, 3, 8, 9, 2, 10]))
        if oligo in ("m13", "i8"):
            s2 = mutate(o["spcr2"])
            if rng.random() < 0.03:
                s2 = s2 + rnd(2) + s2
            seq = rnd(rng.choice([0, 0, 0, 1, 2, 5, 11])) + s1 + n1 + s2 + rnd(rng.choice([6, 6, 6, 7, 10, 4, 2, 0]))
        elif oligo == "i8_single":
            seq = n1 + s1 + rnd(rng.choice([6, 6, 8, 3, 0]))
        elif oligo == "nebio":
            seq = rnd(rng.choice([17, 17, 18, 16, 21, 5])) + s1 + rnd(rng.choice([0, 3, 5]))
        else:
            seq = rnd(rng.choice([12, 12, 11, 13, 3])) + s1 + rnd(rng.choice([0, 4, 7]))
        if rng.random() < 0.04:
            i = rng.randrange(len(seq)); seq = seq[:i] + "N" + seq[i + 1:]
        if rng.random() < 0.02:
            seq = rnd(rng.randrange(0, 40))
        qual = "".join(chr(33 + rng.choice([40, 40, 38, 37, 35, 30, 25, 20, 19, 12, 2])) for _ in seq)
        if rng.random() < 0.02:
            qual = qual[:rng.randrange(0, len(qual) + 1)]
        inter = rnd(rng.choice([40, 60, 90, 129, 130, 131, 150]))
        rows.append([str(rng.randrange(50)), str(rng.randrange(13)), str(rng.randrange(9)), str(rng.randrange(9)), rnd(rng.randrange(0, 12)),
                     f"read{k}", inter, "I" * len(inter), seq, qual])
    return rows


@pytest.mark.parametrize("oligo", ["m13", "i8", "i8_single", "nebio", "takara"])
def test_library_batch_equals_the_regex_functions_row_for_row(oligo):
    """dcrx_collapse_front (threaded C++: verbatim spacers, the {1s<=2} search and the indel search {2i+2d+1s<=2} all decided
    natively) against the reference's own regex patterns on 6 000 random rows per oligo: the same entry for every row and the
    same counters, with allowNs off and on, and not one row deferred."""
    import random
    rng = random.Random(77 + len(oligo))
    rows = _random_rows(rng, oligo, 6000)
    params = [20, 1, 30]
    for allow in (False, True):
        args = {"oligo": oligo, "allowNs": allow, "lenthreshold": 130}
        ref.counts = coll.Counter()
        want = []
        for r in rows:
            try:
                want.append(ref._row_front(r, args, params))
            except ZeroDivisionError:              # (an empty barcode quality string: the reference's own crash)
                want.append("CRASH")
        want_counts = dict(ref.counts)
        keep = [i for i, w in enumerate(want) if w != "CRASH"]
        ref.counts = coll.Counter()
        for r in (rows[i] for i in range(len(rows)) if want[i] == "CRASH"):     # what the crash rows had counted before they crashed
            try:
                ref._row_front(r, args, params)
            except ZeroDivisionError:
                pass
        crash_counts = ref.counts
        collapse.counts = coll.Counter()
        got = collapse.read_in_rows([rows[i] for i in keep], args, params)
        assert list(got) == [want[i] for i in keep]
        diff = coll.Counter(want_counts); diff.subtract(crash_counts)
        assert {k: v for k, v in collapse.counts.items() if v} == {k: v for k, v in diff.items() if v}
        assert int((got.status == 255).sum()) == 0


def test_spacer_search_equals_regex_findall_on_mutated_spacers():
    """dcrx_spacer_search against regex.findall for the three patterns of spacerSearch (collapse.py:204-212): every match string, in
    order, on 60 000 windows built around mutated spacers (insertions, deletions, substitutions, repeats, truncations) — the
    indel stage's choice among alignments is the regex module's backtracking order, restated natively; tools/fuzz_spacer_search.py
    runs the same comparison over millions."""
    import random
    rng = random.Random(4242)
    spacers = [v for o in collapse.OLIGOS.values() for v in o.values()]
    n_indel = 0
    for it in range(60000):
        sp = rng.choice(spacers)
        parts = []
        for _ in range(rng.randrange(1, 4)):
            t = list(sp)
            for _ in range(rng.randrange(0, 3)):
                r = rng.random()
                if r < 0.4 and len(t) > 1:
                    del t[rng.randrange(len(t))]
                elif r < 0.8:
                    t.insert(rng.randrange(len(t) + 1), rng.choice("ACGT"))
                else:
                    t[rng.randrange(len(t))] = rng.choice("ACGT")
            parts.append("".join(t))
            parts.append("".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 4))))
        s = "".join(parts)
        if rng.random() < 0.3:
            s = s[rng.randrange(0, 4):]
        if rng.random() < 0.3:
            s = s[:len(s) - rng.randrange(0, 4)]
        want = ref.spacerSearch(sp, s)
        got = collapse.spacerSearch(sp, s)
        assert got == want, (sp, s, got, want)
        n_indel += bool(want) and len(want[0]) != len(sp)
    assert n_indel > 3000


STAGE_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collapse_stage.json")


def _stage_to_front(tmp_path, capsys=None):
    """FASTQ files -> decombinator() -> rows -> the front half of collapse, against what the reference's read_in_data loop made of
    the same rows (tests/golden/collapse_stage.json, oracle/gen_collapse_golden.py)."""
    from decombinator_amd import decombine as dec, io as dio, synth, pipeline
    fx = json.load(open(STAGE_FIXTURE))
    stage = json.load(open(os.path.join(os.path.dirname(STAGE_FIXTURE), fx["stage"])))
    ts = stage["tagset"]
    synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"], v_jumps=ts["v_jumps"],
                 v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"], j_jumps=ts["j_jumps"],
                 j_names=ts["j_names"], j_regions=ts["j_regions"]).write(str(tmp_path / "tags"))
    (tmp_path / "SYNTH_1.fq").write_text(stage["fastq_r1"])
    (tmp_path / "SYNTH_2.fq").write_text(fx["fastq_r2"])
    (tmp_path / "out").mkdir()
    args = dio.create_args_dict(infile=str(tmp_path / "SYNTH_1.fq"), chain="b", bc_read="R2", dontgzip=True, dontcount=True,
                                orientation="reverse", allowNs=False, tagfastadir=str(tmp_path / "tags"), oligo=fx["oligo"],
                                outpath=str(tmp_path / "out") + os.sep, command="pipeline")
    rows = pipeline.run(args)                       # decombine (+ .n12) and the front half with the stage's own flags
    assert rows == fx["rows"]
    assert {k: v for k, v in collapse.counts.items() if v} == fx["counts"]
    collapse.counts.clear()
    front = collapse.read_in_rows(rows, {"oligo": fx["oligo"], "allowNs": False, "lenthreshold": 130}, fx["params"])
    assert [None if g is None else list(g) for g in front] == fx["expect"]
    assert {k: v for k, v in collapse.counts.items() if v} == fx["counts"]
    assert sum(1 for e in fx["expect"] if e) >= 20 and len(front.kept()) == sum(1 for e in fx["expect"] if e)
    # the `collapse` sub-command over the .n12 the pipeline wrote: the same rows pass
    n12 = str(tmp_path / "out" / "dcr_SYNTH_1_beta.n12")
    pipeline.main(["collapse", "-in", n12, "-ol", fx["oligo"], "-dz", "-op", str(tmp_path / "out") + os.sep])
    got = [ln.split(", ") for ln in open(str(tmp_path / "out" / "dcr_SYNTH_1_beta.n12u")).read().splitlines()]
    want = [e[2] + [e[5], e[3], e[4], e[0], e[1]] for e in fx["expect"] if e]
    assert got == want


def test_fastq_to_front_half_with_oracle_as_device(tmp_path, monkeypatch):
    from decombinator_amd import _native as nat
    from tests import test_host_stage as ths
    fx = json.load(open(STAGE_FIXTURE))
    stage = json.load(open(os.path.join(os.path.dirname(STAGE_FIXTURE), fx["stage"])))
    monkeypatch.setattr(nat, "decombine", ths._oracle_device(stage))
    _stage_to_front(tmp_path)


@pytest.mark.gpu
def test_fastq_to_front_half_through_hip_path(tmp_path):
    _stage_to_front(tmp_path)


def test_module_imports_in_a_clean_interpreter():
    """`from decombinator import collapse` as a library: nothing before it has imported collections.abc."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod in ("collapse", "translate", "io", "decombine", "sharded", "pipeline"):
        p = subprocess.run([sys.executable, "-S", "-c", f"import sys; sys.path.insert(0, {root!r}); "
                            f"import site; site.main(); import decombinator_amd.{mod}"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (mod, p.stderr[-2000:])
