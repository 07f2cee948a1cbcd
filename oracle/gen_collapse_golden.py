#!/usr/bin/env python3
"""Generates tests/golden/collapse_front.json from the reference's UNMODIFIED collapse.py, imported in the
build container (stand-ins for the two wheels that are absent offline and that the row front half never
calls: oracle/refshim/polyleven.py, oracle/refshim/pyrepseq).  TEST INFRASTRUCTURE: never imported by the
product.  Cases: barcode regions built from each oligo's spacers around random N6 / N12 / N17 barcodes, with
0-3 substitutions, insertions, deletions, truncations and Ns in the spacers and barcodes, random quality
strings; per case the reference's get_barcode_positions / set_barcode / check_umi_quality results and the
counter keys they bumped, and per oligo the whole read_in_data front over rows made of those regions."""
import collections as coll
import importlib.metadata
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference/src"]
_v = importlib.metadata.version
importlib.metadata.version = lambda n: "0" if n == "decombinator" else _v(n)
from decombinator import collapse as ref  # noqa: E402


def mutate(rng, s, nsub, nins, ndel):
    s = list(s)
    for _ in range(nsub):
        if s:
            i = rng.randrange(len(s)); s[i] = rng.choice([c for c in "ACGT" if c != s[i]])
    for _ in range(nins):
        s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
    for _ in range(ndel):
        if s:
            del s[rng.randrange(len(s))]
    return "".join(s)


def region(rng, oligo):
    o = ref.getOligo(oligo)
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
    kind = rng.random()
    ns, ni, nd = (0, 0, 0) if kind < 0.5 else rng.choice([(1, 0, 0), (2, 0, 0), (3, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1)])
    s1 = mutate(rng, o["spcr1"], ns, ni, nd)
    n1 = rnd(rng.choice([6, 6, 6, 6, 5, 7, 4, 3, 8, 9]))
    if oligo in ("m13", "i8"):
        s2 = mutate(rng, o["spcr2"], *rng.choice([(0, 0, 0), (0, 0, 0), (1, 0, 0), (2, 0, 0), (0, 1, 0), (0, 0, 1)]))
        seq = rnd(rng.choice([0, 0, 0, 1, 2, 5])) + s1 + n1 + s2 + rnd(rng.choice([6, 6, 6, 7, 10, 4, 2]))
    elif oligo == "i8_single":
        seq = n1 + s1 + rnd(rng.choice([6, 6, 8, 3]))
    elif oligo == "nebio":
        seq = rnd(rng.choice([17, 17, 18, 16, 21])) + s1 + rnd(rng.choice([0, 3, 5]))
    else:
        seq = rnd(rng.choice([12, 12, 11, 13])) + s1 + rnd(rng.choice([0, 4, 7]))
    if rng.random() < 0.05:
        i = rng.randrange(len(seq)); seq = seq[:i] + "N" + seq[i + 1:]
    qual = "".join(chr(33 + rng.choice([40, 40, 38, 37, 35, 30, 25, 20, 12, 2])) for _ in seq)
    return seq, qual


def main():
    rng = random.Random(20261003)
    params = [20, 1, 30]                        # minbcQ, bcQbelowmin, avgQthreshold defaults (reference io.py:409-411)
    out = {"params": params, "cases": [], "read_in": []}
    for oligo in ("m13", "i8", "i8_single", "nebio", "takara"):
        rows = []
        for k in range(300):
            seq, qual = region(rng, oligo)
            allow = rng.random() < 0.2
            args = {"oligo": oligo if k % 3 else oligo.upper(), "allowNs": allow}
            c = coll.Counter()
            locs = ref.get_barcode_positions(seq, args, c)
            case = {"oligo": args["oligo"], "allowNs": allow, "bcseq": seq, "bcqual": qual, "locs": locs, "counts": dict(c)}
            if locs:
                ref.counts = coll.Counter()
                fields = ["1", "2", "3", "4", "ACGT", "id", "SEQ", "QUAL", seq, qual]
                bc, bq = ref.set_barcode(fields, locs, args)
                case["barcode"], case["barcode_qual"], case["set_counts"] = bc, bq, dict(ref.counts)
                case["low_quality"] = bool(ref.check_umi_quality(bq, params)) if bq else None
            out["cases"].append(case)
            inter = "".join(rng.choice("ACGT") for _ in range(rng.choice([40, 60, 90, 129, 130, 131, 150])))
            rows.append([str(rng.randrange(50)), str(rng.randrange(13)), str(rng.randrange(9)), str(rng.randrange(9)),
                         "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 12))), f"read{k}", inter, "I" * len(inter), seq, qual])
        # the reference's own per-row loop, stopped where grouping starts: re-run with its functions, as read_in_data does
        for allow in (False, True):
            args = {"oligo": oligo, "allowNs": allow, "lenthreshold": 130}
            ref.counts = coll.Counter()
            res = []
            for line in rows:
                ref.counts["readdata_input_dcrs"] += 1
                locs = ref.get_barcode_positions(line[8], args, ref.counts)
                if not locs:
                    ref.counts["readdata_fail_no_bclocs"] += 1; res.append(None); continue
                bc, bq = ref.set_barcode(line, locs, args)
                if not bq:                                  # (an empty quality string divides by zero in the reference: not a row it survives)
                    res.append("CRASH"); continue
                if ref.check_umi_quality(bq, params):
                    ref.counts["readdata_fail_low_barcode_quality"] += 1; res.append(None); continue
                if len(line[6]) > args["lenthreshold"]:
                    ref.counts["readdata_fail_overlong_intertag_seq"] += 1; res.append(None); continue
                ref.counts["readdata_success"] += 1
                res.append([bc, bq, line[:5], line[6], line[7], line[5]])
            out["read_in"].append({"oligo": oligo, "allowNs": allow, "rows": rows if not allow else None,   # (same rows for both settings)
                                   "expect": res, "counts": dict(ref.counts)})
    path = os.path.join(HERE, "..", "tests", "golden", "collapse_front.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print(len(out["cases"]), "cases ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")
    stage_fixture(rng, params)


def stage_fixture(rng, params):
    """End-to-end fixture for FASTQ -> decombine -> rows -> collapse front half: the reads of the stage fixture
    (tests/golden/stage_human_extended_b.json, whose rows the reference's decombinator() produced) with a NEW second file
    whose records start with M13 barcode regions (spacers verbatim, substituted, with indels; N1 of 4-8 bases), and what the
    reference's read_in_data loop makes of the rows that result (their fields 8 and 9 are the first 42 bytes of the second
    file's sequence and quality, decombine.py:1021-1034)."""
    stage = json.load(open(os.path.join(HERE, "..", "tests", "golden", "stage_human_extended_b.json")))
    run = stage["runs"][0]
    assert run["bc_read"] == "R2"
    r1 = stage["fastq_r1"].split("\n")
    ids = [r1[i][1:].partition(" ")[0] for i in range(0, len(r1) - 1, 4)]
    bc_of = {}
    r2 = []
    for rid in ids:
        seq, qual = region(rng, "m13")
        seq = (seq + "".join(rng.choice("ACGT") for _ in range(60)))[:60]
        qual = (qual + "I" * 60)[:60]
        bc_of[rid] = (seq[:42], qual[:42])
        r2 += ["@" + rid, seq, "+", qual]
    rows = [r[:8] + list(bc_of[r[5]]) for r in run["rows"]]
    args = {"oligo": "M13", "allowNs": False, "lenthreshold": 130}
    ref.counts = coll.Counter()
    res = []
    for line in rows:
        ref.counts["readdata_input_dcrs"] += 1
        locs = ref.get_barcode_positions(line[8], args, ref.counts)
        if not locs:
            ref.counts["readdata_fail_no_bclocs"] += 1; res.append(None); continue
        bc, bq = ref.set_barcode(line, locs, args)
        if ref.check_umi_quality(bq, params):
            ref.counts["readdata_fail_low_barcode_quality"] += 1; res.append(None); continue
        if len(line[6]) > args["lenthreshold"]:
            ref.counts["readdata_fail_overlong_intertag_seq"] += 1; res.append(None); continue
        ref.counts["readdata_success"] += 1
        res.append([bc, bq, line[:5], line[6], line[7], line[5]])
    out = {"generator": "oracle/gen_collapse_golden.py: reference collapse.py functions on the stage fixture's rows with M13 barcode regions",
           "stage": "stage_human_extended_b.json", "oligo": "M13", "params": params, "fastq_r2": "\n".join(r2) + "\n", "rows": rows,
           "expect": res, "counts": dict(ref.counts)}
    path = os.path.join(HERE, "..", "tests", "golden", "collapse_stage.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print(len(rows), "rows,", sum(1 for x in res if x), "kept ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
