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
    # inserts with IUPAC ambiguity codes (an `allowNs` run leaves N in its inserts; merged or consensus reads carry the other
    # codes): what Biopython's translate makes of such codons — a shared residue, B / Z / J, '*', X — comes out of the reference's
    # get_cdr3 through the shim's translate (oracle/refshim/Bio/Seq.py), not out of hand-written expectations.  (A generator of
    # their own: the cases above and the stage fixture below stay byte for byte what they were.)
    rng2 = random.Random(20261005)
    n_amb = 0
    for command in ("pipeline", "translate"):
        for _ in range(200):
            v, j = rng2.randrange(8), rng2.randrange(5)
            vdel, jdel = rng2.choice([0, 0, 1, 2, 3, 6]), rng2.choice([0, 0, 1, 3, 4])
            ins = [rng2.choice("ACGT") for _ in range(rng2.choice([3, 4, 5, 6, 7, 8, 9, 12]))]
            for _ in range(rng2.choice([1, 1, 2, 3])):
                ins[rng2.randrange(len(ins))] = rng2.choice("NNNRYKMSWBDHVN")
            ins = "".join(ins)
            dcr = [str(v), str(j), str(vdel), str(jdel), (" " + ins) if command == "translate" else ins]
            try:
                out = dict(ref.get_cdr3(dcr, ref.out_headers, {"command": command}))
            except IndexError:
                out = "IndexError"
            cases.append({"command": command, "dcr": dcr, "expect": out, "ambiguous": True})
            n_amb += 1
    # ... and codons placed in frame behind the V region that Biopython gives a letter of its own: RAY -> B (D or N), SAR -> Z (E or Q),
    # MTY -> J (I or L), TAR / TRA -> '*' (every reading a stop), YTA -> L, GCN -> A, NNN -> X
    rng3 = random.Random(20261007)
    for command in ("pipeline", "translate"):
        for special in ("RAY", "SAR", "MTY", "TAR", "TRA", "YTA", "GCN", "NNN", "RAYSARMTY"):
            for _ in range(6):
                v, j = rng3.randrange(8), rng3.randrange(5)
                vdel = next(d for d in range(0, 3) if (len(genes["v_regions"][v]) - d) % 3 == 0)
                jdel = rng3.choice([0, 0, 1, 2])
                ins = "".join(rng3.choice("ACGT") for _ in range(3 * rng3.randrange(0, 2))) + special + "".join(rng3.choice("ACGT") for _ in range(rng3.randrange(0, 6)))
                dcr = [str(v), str(j), str(vdel), str(jdel), (" " + ins) if command == "translate" else ins]
                try:
                    out = dict(ref.get_cdr3(dcr, ref.out_headers, {"command": command}))
                except IndexError:
                    out = "IndexError"
                cases.append({"command": command, "dcr": dcr, "expect": out, "ambiguous": True})
                n_amb += 1
    print(n_amb, "cases with ambiguity codes in the insert,",
          sum(1 for c in cases if c.get("ambiguous") and c["expect"] != "IndexError" and any(x in c["expect"]["junction_aa"] for x in "XBZJ")), "with X/B/Z/J in junction_aa")
    path = os.path.join(HERE, "..", "tests", "golden", "translate_cdr3.json")
    json.dump({"genes": genes, "headers": ref.out_headers, "cases": cases}, open(path, "w"), separators=(",", ":"))
    prod = sum(1 for c in cases if c["expect"] != "IndexError" and c["expect"]["productive"] == "T")
    print(len(cases), "cases,", prod, "productive ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")
    stage_fixture(rng)
    coding_stage_fixture()


def coding_stage_fixture():
    """A second end-to-end fixture FASTQ -> decombine -> rows -> get_cdr3 whose germlines CODE: V regions are the coding strand
    of a protein that ends in the conserved C and the start of a CDR3, J regions hold an FGXG motif in frame, and most reads
    are in-frame rearrangements of them — so that the productive branch of get_cdr3 (translate.py:312-350: conserved residues
    found, no stop, in frame) is what most rows take.  The reference's own decombinator() decombines the files (rows), the
    reference's get_cdr3 translates the rows' unique DCRs (expect)."""
    import contextlib
    import io as _io
    import tempfile
    ROOT = os.path.normpath(os.path.join(HERE, ".."))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from decombinator_amd import synth
    from oracle import ref_driver
    rng = random.Random(20261006)
    bases = "TCAG"; aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    cod = {}
    for i, a in enumerate(bases):
        for j, b in enumerate(bases):
            for k, c in enumerate(bases):
                cod.setdefault(aas[16 * i + 4 * j + k], []).append(a + b + c)
    nt = lambda prot: "".join(rng.choice(cod[x]) for x in prot)
    rp = lambda k: "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(k))
    n_v, n_j = 12, 6
    v_regions, v_tags, v_jumps, v_names, j_regions, j_tags, j_jumps, j_names = [], [], [], [], [], [], [], []
    vpos, vres, jpos, jres = [], [], [], []
    for i in range(n_v):
        lead = rp(rng.randrange(94, 104))
        region = nt(lead + "C" + "ASS" + rp(1))                       # ~300 nt, a multiple of 3: the frame of the region's first base
        jump = rng.choice([36, 39, 40, 43, 44, 53])
        v_regions.append(region); v_jumps.append(jump); v_tags.append(region[len(region) - jump:len(region) - jump + 20])
        v_names.append(f"TRBV{i + 1}-1*01")
        vpos.append(len(lead) + 1); vres.append("C")
    for i in range(n_j):
        pre = rp(rng.randrange(5, 8))
        motif = rng.choice(["FGQG", "FGSG", "FGAG", "FGPG"])
        post = rp(rng.randrange(9, 12))
        lead_nt = rng.choice(["", "T", "GA"])                         # (a J gene's first bases belong to the junction's last codon)
        region = lead_nt + nt(pre + motif + post) + rng.choice("ACGT")
        j_regions.append(region); j_jumps.append(20); j_tags.append(region[20:40])
        j_names.append(f"TRBJ{i + 1}-1*01")
        jpos.append(-(len(post) + 4)); jres.append("FG.G")
    assert len(set(v_tags)) == n_v and len(set(j_tags)) == n_j
    ts = synth.TagSet(species="human", tags="extended", chain="b", v_tags=v_tags, v_jumps=v_jumps, v_names=v_names, v_regions=v_regions,
                      j_tags=j_tags, j_jumps=j_jumps, j_names=j_names, j_regions=j_regions)
    comp = str.maketrans("ACGT", "TGCA")
    rnd = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
    r1, r2 = [], []
    n_pairs = 600
    for i in range(n_pairs):
        kind = rng.random()
        if kind < 0.15:
            sense = rnd(150)                                          # background
        else:
            v, j = rng.randrange(n_v), rng.randrange(n_j)
            vdel, jdel = rng.choice([0, 0, 1, 2, 3, 4, 6]), rng.choice([0, 0, 1, 2, 3, 5])
            lead_j = len(j_regions[j]) - 1 - 3 * ((len(j_regions[j]) - 1) // 3)      # bases of the J region in front of its first whole codon
            # the insert that keeps the J region's codons in the V region's frame (most reads), or one that does not
            need = (-(len(v_regions[v]) - vdel) - (lead_j - jdel)) % 3
            ilen = need + 3 * rng.randrange(0, 4)
            if kind > 0.85:
                ilen += rng.choice([1, 2])
            ins = rnd(ilen)
            up = rng.randrange(64, 92)                                # bases of the V region in front of its end (holds the tag: jump <= 53)
            amplicon = v_regions[v][len(v_regions[v]) - up:len(v_regions[v]) - vdel] + ins + j_regions[j][jdel:]
            sense = (rnd(rng.randrange(0, 12)) + amplicon + rnd(150))[:150]
            if rng.random() < 0.03:
                k = rng.randrange(150); sense = sense[:k] + rng.choice("ACGT") + sense[k + 1:]
        read = sense.translate(comp)[::-1]                            # R1 holds the antisense strand (orientation reverse)
        q1 = "".join(chr(rng.randrange(35, 74)) for _ in read)
        bc = rnd(42); tail = rnd(108)
        q2 = "".join(chr(rng.randrange(35, 74)) for _ in range(150))
        name = f"COD:{i}:{rng.randrange(1000, 9999)}"
        r1.append(f"@{name} 1:N:0:AAAA\n{read}\n+\n{q1}\n")
        r2.append(f"@{name} 2:N:0:AAAA\n{bc + tail}\n+\n{q2}\n")
    fq1, fq2 = "".join(r1), "".join(r2)
    m = ref_driver.module()
    with tempfile.TemporaryDirectory() as td:
        tagdir = os.path.join(td, "tags"); ts.write(tagdir)
        open(os.path.join(td, "CODING_1.fq"), "w").write(fq1)
        open(os.path.join(td, "CODING_2.fq"), "w").write(fq2)
        outdir = os.path.join(td, "out") + os.sep
        os.makedirs(outdir)
        args = dict(infile=os.path.join(td, "CODING_1.fq"), chain="b", bc_read="R2", suppresssummary=True, dontgzip=True, dontcheck=True,
                    dontcount=True, extension="n12", prefix="dcr_", orientation="reverse", tags="extended", species="human", allowNs=False,
                    lenthreshold=130, tagfastadir=tagdir, nobarcoding=False, bclength=42, outpath=outdir, dontsave=False,
                    command="decombine", sampling_analysis=False)
        with contextlib.redirect_stdout(_io.StringIO()):
            rows = m.decombinator(dict(args))
    genes = {"v_regions": [r.upper() for r in v_regions], "j_regions": [r.upper() for r in j_regions], "v_names": v_names, "j_names": j_names,
             "v_translate_position": vpos, "v_translate_residue": vres, "j_translate_position": jpos, "j_translate_residue": jres,
             "v_functionality": ["F"] * n_v, "j_functionality": ["F"] * n_j,
             "v_cdr1": [rp(6) for _ in range(n_v)], "v_cdr2": [rp(5) for _ in range(n_v)]}
    for k, v in genes.items():
        setattr(ref, k, v)
    seen, dcrs = set(), []
    for row in rows:
        d = tuple(row[:5])
        if d not in seen:
            seen.add(d); dcrs.append(list(d))
    expect = []
    for d in dcrs:
        try:
            expect.append(dict(ref.get_cdr3(d, ref.out_headers, {"command": "pipeline"})))
        except IndexError:
            expect.append("IndexError")
    n_prod = sum(1 for e in expect if e != "IndexError" and e["productive"] == "T")
    assert n_prod >= 0.3 * len(dcrs), (n_prod, len(dcrs))
    vs, js = ts.half_splits
    tagset = {"species": ts.species, "tags": ts.tags, "chain": ts.chain, "v_tags": v_tags, "v_jumps": v_jumps, "v_names": v_names,
              "v_regions": v_regions, "j_tags": j_tags, "j_jumps": j_jumps, "j_names": j_names, "j_regions": j_regions,
              "v_half_split": vs, "j_half_split": js}
    path = os.path.join(HERE, "..", "tests", "golden", "translate_stage_coding.json")
    json.dump({"generator": "oracle/gen_translate_golden.py coding_stage_fixture: reference decombinator() + get_cdr3 on coding germlines",
               "tagset": tagset, "fastq_r1": fq1, "fastq_r2": fq2, "rows": rows, "genes": genes, "dcrs": dcrs, "expect": expect},
              open(path, "w"), separators=(",", ":"))
    print(len(rows), "rows,", len(dcrs), "unique DCRs,", n_prod, "productive ->", os.path.normpath(path), os.path.getsize(path) // 1024, "KiB")


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
