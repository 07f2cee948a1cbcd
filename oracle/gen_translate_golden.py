#!/usr/bin/env python3
"""Generates tests/golden/translate_cdr3.json from the reference's UNMODIFIED translate.py imported in the
build container (oracle/refshim/Bio stands in for Biopython: Seq.translate restates the standard table).
TEST INFRASTRUCTURE.  Gene tables: synthetic V regions that end in a conserved C + a CDR3 start, J regions
that hold an FGXG motif, with the `.translate` positions derived from them (as the reference's files hold
them); DCRs over every combination of deletions / inserts, in and out of frame, with stops, with broken
motifs, for both `command` values."""
import importlib.metadata
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference/src"]
_v = importlib.metadata.version
importlib.metadata.version = lambda n: "0" if n == "decombinator" else _v(n)
from decombinator import translate as ref  # noqa: E402


def main():
    rng = random.Random(20261004)
    cod = {}
    bases = "TCAG"; aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    for i, a in enumerate(bases):
        for j, b in enumerate(bases):
            for k, c in enumerate(bases):
                cod.setdefault(aas[16 * i + 4 * j + k], []).append(a + b + c)
    nt = lambda prot: "".join(rng.choice(cod[x]) for x in prot)
    rp = lambda k: "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(k))
    genes = {k: [] for k in ("v_regions", "j_regions", "v_names", "j_names", "v_translate_position", "v_translate_residue",
                             "j_translate_position", "j_translate_residue", "v_functionality", "j_functionality", "v_cdr1", "v_cdr2")}
    for i in range(8):
        lead = rp(rng.randrange(85, 100))
        prot = lead + "C" + "ASS" + rp(2)
        genes["v_regions"].append(nt(prot) + rng.choice(["", "A", "AG"]))
        genes["v_names"].append(f"TRBV{i + 1}*0{1 + i % 2}")
        genes["v_translate_position"].append(len(lead) + 1)
        genes["v_translate_residue"].append("C" if i != 6 else "W")
        genes["v_functionality"].append(rng.choice("FPO"))
        genes["v_cdr1"].append(rp(6)); genes["v_cdr2"].append(rp(5))
    for i in range(5):
        pre = rp(rng.randrange(3, 6))
        motif = rng.choice(["FGQG", "FGSG", "WGKG", "FGAG"])
        post = rp(rng.randrange(6, 9))
        genes["j_regions"].append(rng.choice(["", "T", "GA"]) + nt(pre + motif + post) + rng.choice("ACGT"))   # (a J exon ends inside a codon)
        genes["j_names"].append(f"TRBJ{i + 1}-1*01")
        genes["j_translate_position"].append(-(len(post) + 4))
        genes["j_translate_residue"].append(rng.choice(["FG.G", "[FW]G.G"]))
        genes["j_functionality"].append("F")
    for k, v in genes.items():
        setattr(ref, k, v)
    cases = []
    for command in ("pipeline", "translate"):
        for _ in range(400):
            v, j = rng.randrange(8), rng.randrange(5)
            vdel, jdel = rng.choice([0, 0, 1, 2, 3, 5, 8]), rng.choice([0, 0, 1, 2, 4, 7])
            ins = "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 1, 2, 3, 4, 5, 6, 9, 12])))
            dcr = [str(v), str(j), str(vdel), str(jdel), (" " + ins) if command == "translate" else ins]
            try:
                out = dict(ref.get_cdr3(dcr, ref.out_headers, {"command": command}))
            except IndexError:
                out = "IndexError"
            cases.append({"command": command, "dcr": dcr, "expect": out})
    path = os.path.join(HERE, "..", "tests", "golden", "translate_cdr3.json")
    json.dump({"genes": genes, "headers": ref.out_headers, "cases": cases}, open(path, "w"), separators=(",", ":"))
    prod = sum(1 for c in cases if c["expect"] != "IndexError" and c["expect"]["productive"] == "T")
    print(len(cases), "cases,", prod, "productive ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")
    stage_fixture(rng)


def stage_fixture(rng):
    """End-to-end fixture FASTQ -> decombine -> rows -> get_cdr3: the unique DCRs of the stage fixture's rows
    (tests/golden/stage_human_extended_b.json: what the reference's decombinator() returned for its FASTQ files) through the
    reference's get_cdr3 with `.translate` / `.cdrs` tables made up for that tag set (conserved-residue positions inside the V
    region with the residue that really stands there for most genes, J motifs as regular expressions)."""
    stage = json.load(open(os.path.join(HERE, "..", "tests", "golden", "stage_human_extended_b.json")))
    ts = stage["tagset"]
    from Bio.Seq import Seq
    genes = {"v_regions": [r.upper() for r in ts["v_regions"]], "j_regions": [r.upper() for r in ts["j_regions"]],
             "v_names": [n.upper() for n in ts["v_names"]], "j_names": [n.upper() for n in ts["j_names"]]}
    vpos, vres = [], []
    for r in genes["v_regions"]:
        aa = str(Seq(r).translate())
        p = rng.randrange(5, max(6, len(aa) - 12))
        vpos.append(p)
        vres.append(aa[p - 1] if rng.random() < 0.8 and aa[p - 1] != "*" else "C")
    genes.update(v_translate_position=vpos, v_translate_residue=vres,
                 j_translate_position=[-rng.randrange(6, 12) for _ in genes["j_regions"]],
                 j_translate_residue=[rng.choice(["[A-Z]G.G", "F...", "....", "FG.G"]) for _ in genes["j_regions"]],
                 v_functionality=[rng.choice("FPO") for _ in genes["v_regions"]], j_functionality=["F" for _ in genes["j_regions"]],
                 v_cdr1=["".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(6)) for _ in genes["v_regions"]],
                 v_cdr2=["".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(5)) for _ in genes["v_regions"]])
    for k, v in genes.items():
        setattr(ref, k, v)
    seen, dcrs = set(), []
    for row in stage["runs"][0]["rows"]:
        d = tuple(row[:5])
        if d not in seen:
            seen.add(d); dcrs.append(list(d))
    expect = []
    for d in dcrs:
        try:
            expect.append(dict(ref.get_cdr3(d, ref.out_headers, {"command": "pipeline"})))
        except IndexError:
            expect.append("IndexError")
    path = os.path.join(HERE, "..", "tests", "golden", "translate_stage.json")
    json.dump({"generator": "oracle/gen_translate_golden.py: reference get_cdr3 on the unique DCRs of the stage fixture's rows",
               "stage": "stage_human_extended_b.json", "genes": genes, "dcrs": dcrs, "expect": expect}, open(path, "w"), separators=(",", ":"))
    print(len(dcrs), "unique DCRs,", sum(1 for e in expect if e != "IndexError" and e["productive"] == "T"), "productive,",
          sum(1 for e in expect if e == "IndexError"), "IndexError ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
