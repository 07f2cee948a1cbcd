#!/usr/bin/env python3
"""Rebuilds a PARTIAL human extended tag set from the reference's own fixtures and checks
how many rows of the reference's golden `.n12` the path reproduces with it.

Container-only (reads /root/reference/tests/resources).  The real tag / germline files
(git submodule Decombinator-Tags-FASTAs) are absent, but every row of
tests/resources/dcr_TINY_1_{alpha,beta}.n12 together with the matching row of the `.tsv`
(whose `sequence` column is germline V[:len-vdel] + insert + germline J[jdel:]) pins, for the
genes that occur:
    tag   = the germline 20-mer at the V-tag start / before the J-tag end of the inter-tag window
    jump  = V: region length - tag offset;  J: tag offset (20 in every fixture row)
    region= the germline as far as the fixtures show it; the deleted end is padded with 'N'
Indices that never occur get a never-matching dummy so that gene indices line up.

Writes tests/golden/tiny_<chain>.json: the reconstructed tag set, the TINY FASTQ pair (the
reference's test data), the rows of the reference's `.n12` fixture, what the reference itself
returns with the reconstructed set, and which fixture rows that reproduces.
"""
from __future__ import annotations

import csv
import json
import os
import sys
import tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from decombinator_amd import synth  # noqa: E402
from oracle import ref_driver  # noqa: E402

RES = "/root/reference/tests/resources"
J_JUMP = 20


def hamming(a, b):
    return sum(x != y for x, y in zip(a, b)) + abs(len(a) - len(b))


def rebuild(chain: str):
    name = {"a": "alpha", "b": "beta"}[chain]
    rows = [ln.rstrip("\n").split(", ") for ln in open(f"{RES}/dcr_TINY_1_{name}.n12")]
    tsv = {r["decombinator_id"]: r for r in csv.DictReader(open(f"{RES}/dcr_TINY_1_{name}.tsv"), delimiter="\t")}
    vinfo, jinfo = defaultdict(list), defaultdict(list)
    for r in rows:
        v, j, vdel, jdel, ins, inter = int(r[0]), int(r[1]), int(r[2]), int(r[3]), r[4], r[6]
        vpart = len(inter) - len(ins) - (20 + J_JUMP - jdel)   # V bases inside the inter-tag window
        if vpart < 20:
            continue
        # what the read itself shows: tag onwards to the (deleted) V end, J start to the tag end
        vrec = {"tag": inter[:20], "germ_tag": None, "prefix": None, "tail": inter[:vpart], "vdel": vdel, "name": None}
        jrec = {"tag": inter[-20:], "germ_tag": None, "head": inter[vpart + len(ins):], "jdel": jdel, "name": None,
                "after": ""}
        t = tsv.get(", ".join(r[:5]))
        if t is not None:
            seq = t["sequence"]
            # align the window to the germline-built sequence: insert must match exactly, V part nearly
            best = None
            for off in range(0, len(seq) - len(inter) + 1):
                if seq[off + vpart: off + vpart + len(ins)] != ins:
                    continue
                d = hamming(seq[off: off + vpart], inter[:vpart])
                if best is None or d < best[0]:
                    best = (d, off)
            if best is not None and best[0] <= 3:
                off = best[1]
                vrec.update(germ_tag=seq[off: off + 20], prefix=seq[:off], tail=seq[off: off + vpart], name=t["v_call"])
                jstart = off + vpart + len(ins)
                jrec.update(germ_tag=seq[off + len(inter) - 20: off + len(inter)], head=seq[jstart: off + len(inter)],
                            after=seq[off + len(inter):], name=t["j_call"])
        vinfo[v].append(vrec)
        jinfo[j].append(jrec)
    nv = max(int(r[0]) for r in rows) + 1
    nj = max(int(r[1]) for r in rows) + 1
    ts = synth.TagSet(species="human", tags="extended", chain=chain)
    dummy_i = [0]

    def dummy():
        dummy_i[0] += 1
        n = dummy_i[0]
        # 20-mers over {AT}/{CG} blocks that no read contains
        s = "".join("ACGT"[(n >> (2 * k)) & 3] for k in range(10))
        return "TTTTTAAAAA" + s

    def pick_tag(recs):
        germ = [x["germ_tag"] for x in recs if x["germ_tag"]]
        return Counter(germ or [x["tag"] for x in recs]).most_common(1)[0][0]

    for v in range(nv):
        if v in vinfo:
            recs = vinfo[v]
            tag = pick_tag(recs)
            info = min(recs, key=lambda x: (x["vdel"], x["germ_tag"] is None))   # least-deleted read shows most of the end
            prefix = next((x["prefix"] for x in recs if x["prefix"] is not None), "")
            region = prefix + tag + info["tail"][20:] + "N" * info["vdel"]
            jump = len(region) - len(prefix)
            name = next((x["name"] for x in recs if x["name"]), f"V{v}")
            ts.v_tags.append(tag); ts.v_jumps.append(jump); ts.v_names.append(name); ts.v_regions.append(region)
        else:
            ts.v_tags.append(dummy()); ts.v_jumps.append(40); ts.v_names.append(f"UNSEEN_V{v}"); ts.v_regions.append("N" * 60)
    for j in range(nj):
        if j in jinfo:
            recs = jinfo[j]
            tag = pick_tag(recs)
            info = min(recs, key=lambda x: (x["jdel"], x["germ_tag"] is None))
            after = next((x["after"] for x in recs if x["after"]), "")
            region = "N" * info["jdel"] + info["head"][:-20] + tag + after
            assert region.find(tag) == J_JUMP, (j, region.find(tag))
            name = next((x["name"] for x in recs if x["name"]), f"J{j}")
            ts.j_tags.append(tag); ts.j_jumps.append(J_JUMP); ts.j_names.append(name); ts.j_regions.append(region)
        else:
            ts.j_tags.append(dummy()); ts.j_jumps.append(J_JUMP); ts.j_names.append(f"UNSEEN_J{j}"); ts.j_regions.append("N" * 60)
    return ts, rows


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    fq1 = open(f"{RES}/TINY_1.fq").read()
    fq2 = open(f"{RES}/TINY_2.fq").read()
    for chain in ("a", "b"):
        ts, fixture_rows = rebuild(chain)
        import contextlib
        import io
        m = ref_driver.module()
        with tempfile.TemporaryDirectory() as td:
            tagdir = os.path.join(td, "tags"); ts.write(tagdir)
            open(os.path.join(td, "TINY_1.fq"), "w").write(fq1)
            open(os.path.join(td, "TINY_2.fq"), "w").write(fq2)
            args = dict(infile=os.path.join(td, "TINY_1.fq"), chain=chain, bc_read="R2", suppresssummary=True,
                        dontgzip=True, dontcheck=True, dontcount=True, extension="n12", prefix="dcr_",
                        orientation="reverse", tags="extended", species="human", allowNs=False, lenthreshold=130,
                        tagfastadir=tagdir, nobarcoding=False, bclength=42, outpath=td + os.sep, dontsave=True,
                        command="decombine", sampling_analysis=False)
            with contextlib.redirect_stdout(io.StringIO()):
                got = m.decombinator(dict(args))
            counts = {k: v for k, v in m.counts.items() if isinstance(v, int)}
        fix = [tuple(r) for r in fixture_rows]
        gotset = Counter(tuple(r) for r in got)
        reproduced = [i for i, r in enumerate(fix) if gotset[r] > 0]
        print(f"chain {chain}: {len(ts.v_tags)} V / {len(ts.j_tags)} J entries; reference with the reconstructed "
              f"set returns {len(got)} rows; {len(reproduced)} of {len(fix)} fixture rows reproduced")
        v_split, j_split = ts.half_splits
        json.dump({
            "generator": "oracle/rebuild_tiny_tagset.py",
            "source": "reference tests/resources/dcr_TINY_1_%s.{n12,tsv} + TINY_{1,2}.fq" % {"a": "alpha", "b": "beta"}[chain],
            "tagset": {"species": ts.species, "tags": ts.tags, "chain": ts.chain, "v_tags": ts.v_tags,
                       "v_jumps": ts.v_jumps, "v_names": ts.v_names, "v_regions": ts.v_regions, "j_tags": ts.j_tags,
                       "j_jumps": ts.j_jumps, "j_names": ts.j_names, "j_regions": ts.j_regions,
                       "v_half_split": v_split, "j_half_split": j_split},
            "fastq_r1": fq1, "fastq_r2": fq2,
            "reference_fixture_rows": fixture_rows,
            "reproduced_fixture_rows": reproduced,
            "rows_with_reconstructed_tagset": got,
            "counts_with_reconstructed_tagset": counts,
        }, open(os.path.join(out_dir, f"tiny_{ {'a': 'alpha', 'b': 'beta'}[chain] }.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
