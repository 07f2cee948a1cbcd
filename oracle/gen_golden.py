#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Container-only (needs /root/reference); the outputs are data — tag sets,
reads, and what the reference's own unmodified decombine.py returned for them —
and are committed so that the GPU box (which has no reference tree) can pin the
oracle and the HIP path against them.

    python oracle/gen_golden.py            # writes tests/golden/*.json
    python oracle/gen_golden.py --bulk N   # additionally cross-checks the C oracle
                                           # against the reference on N random reads
                                           # per tag set (not committed)

Each fixture: {"tagset": {...}, "cases": [{"label", "read" (FASTQ frame),
"orientation", "allowNs", "lenthreshold", "expect" (dcr()'s 7-list or null),
"frame", "counts" (reference Counter delta)}]}.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from decombinator_amd import synth  # noqa: E402
from oracle import casegen, ref_driver  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def tagset_dict(ts: synth.TagSet) -> dict:
    v_split, j_split = ts.half_splits
    return {
        "species": ts.species, "tags": ts.tags, "chain": ts.chain,
        "v_tags": ts.v_tags, "v_jumps": ts.v_jumps, "v_names": ts.v_names, "v_regions": ts.v_regions,
        "j_tags": ts.j_tags, "j_jumps": ts.j_jumps, "j_names": ts.j_names, "j_regions": ts.j_regions,
        "v_half_split": v_split, "j_half_split": j_split,
    }


def run_cases(ts: synth.TagSet, cases, tagdir: str):
    """cases: (label, fastq_frame_read, orientation, allowNs, lenthreshold)."""
    ts.write(tagdir)
    ref = ref_driver.RefChain(tagdir, ts.species, ts.tags, ts.chain)
    out = []
    for label, read, orientation, allow_ns, lenthr in cases:
        recom, frame, delta = ref.decombine_read(read, orientation, allow_ns, lenthr)
        out.append({
            "label": label, "read": read, "orientation": orientation, "allowNs": allow_ns,
            "lenthreshold": lenthr, "expect": recom if recom else None, "frame": frame,
            "counts": delta,
        })
    return out


def build_cases(ts, seed: int, n_mix: int, read_len: int = 150, sub_rate: float = 0.005):
    rng = np.random.default_rng(seed)
    cases = []
    for label, read, kw in casegen.engineered_cases(ts, rng, read_len):
        orientation = kw.get("orientation", "reverse")
        fq = casegen.revcomp(read) if "orientation" not in kw else read
        cases.append((label, fq, orientation, kw.get("allow_ns", False), kw.get("lenthreshold", 130)))
    for _ in range(n_mix):
        sense = casegen.mixture_read(ts, rng, read_len, sub_rate=sub_rate, n_rate=0.01)
        cases.append(("mix", casegen.revcomp(sense), "reverse", False, 130))
    return cases


def summarize(name, cases):
    from collections import Counter
    c = Counter()
    ok = 0
    for cs in cases:
        ok += cs["expect"] is not None
        for k, v in cs["counts"].items():
            c[k] += v
    print(f"{name}: {len(cases)} cases, {ok} decombined; counters: {dict(sorted(c.items()))}")


def bulk_crosscheck(ts, tagdir, n, seed):
    """Reference (Python) vs the C oracle on n random mixture reads."""
    from oracle import oracle as orc
    ts.write(tagdir)
    ref = ref_driver.RefChain(tagdir, ts.species, ts.tags, ts.chain)
    vs, js = ts.half_splits
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions],
                          ts.j_tags, ts.j_jumps, [r.upper() for r in ts.j_regions], vs, js)
    rng = np.random.default_rng(seed)
    bad = 0
    tot = np.zeros(orc.N_COUNTERS, dtype=np.uint64)
    ref.reload()
    for i in range(n):
        sense = casegen.mixture_read(ts, rng, 150, sub_rate=0.01, n_rate=0.01)
        fq = casegen.revcomp(sense)
        recom, frame, _ = ref.decombine_read(fq, "reverse")
        ok, res = ot.decombine_read(fq, 0, counts=tot)
        mine = None
        if ok:
            rc = orc.revcomp(fq)
            mine = [res.v, res.j, res.vdel, res.jdel, rc[res.ins_start:res.ins_start + res.ins_len],
                    res.v_start, res.j_end]
        if (recom or None) != mine:
            bad += 1
            if bad < 5:
                print("MISMATCH", fq, recom, mine)
    refc = ref.counts()
    for idx, nm in enumerate(orc.COUNTER_NAMES):
        if int(tot[idx]) != int(refc.get(nm, 0)) and nm != "frame_forward":
            bad += 1
            print("COUNTER MISMATCH", nm, int(tot[idx]), refc.get(nm, 0))
    print(f"bulk {ts.species}/{ts.tags}/{ts.chain}: {n} reads, {bad} mismatches")
    return bad


def make_edge_tagset(seed: int = 14) -> synth.TagSet:
    """Hand-built set that breaks every regularity of the real ones: unequal tag
    lengths, a tag that is a suffix of another, duplicate tag strings, a V tag
    that is also a J tag, regions shorter than the walk window (negative Python
    slice starts), jumps that leave the region, a distant J tag (jump 70) and a
    2-nt half tag."""
    rng = np.random.default_rng(seed)
    rs = lambda n: casegen.rand_seq(rng, n)
    ts = synth.TagSet(species="human", tags="extended", chain="b")
    base = [rs(20) for _ in range(10)]
    v_tags = list(base[:6])
    v_tags.append(rs(4) + base[0])            # 24 nt, base[0] is its suffix  -> two hits at one end
    v_tags.append(base[1][:10] + rs(8))       # 18 nt sharing half1 with base[1]
    v_tags.append(rs(12) + base[2][10:])      # 22 nt, its last 10 = half2 of base[2] (different split offset)
    v_tags.append(base[3])                    # duplicate string of gene 3
    v_tags.append(base[4][:10] + base[5][10:])  # chimera of two tags' halves
    v_tags.append(rs(20))
    for i, t in enumerate(v_tags):
        jump = int(rng.choice([36, 39, 40, 44, 53, 25]))
        if i == 5:
            length = 30        # region shorter than jump: tag is not inside it
        elif i == 11:
            length = 8         # region shorter than the 10-nt window
        elif i == 2:
            length = 64        # walk reaches the left edge of the region
        else:
            length = int(rng.integers(120, 200))
        if length >= jump:
            right = rs(max(0, jump - len(t)))
            reg = (rs(max(0, length - jump)) + t + right)
            reg = reg[:length - jump] + (t + right)[:jump] if length - jump >= 0 else reg
        else:
            reg = rs(length)
        ts.v_tags.append(t); ts.v_jumps.append(jump); ts.v_names.append(f"EDGEV{i}"); ts.v_regions.append(reg)
    j_tags = [rs(20) for _ in range(4)]
    j_tags.append(base[0])                    # same string as V gene 0
    j_tags.append(rs(12))                     # 12 nt: half1 10 nt, half2 2 nt
    j_tags.append(j_tags[1][:10] + rs(10))    # shares half1 with J gene 1
    for i, t in enumerate(j_tags):
        jump = 70 if i == 2 else (0 if i == 3 else 20)
        reg = rs(jump) + t + rs(int(rng.integers(8, 60)))
        if i == 0:
            reg = reg[:jump + len(t) + 3]     # J region ends 3 nt after the tag
        ts.j_tags.append(t); ts.j_jumps.append(jump); ts.j_names.append(f"EDGEJ{i}"); ts.j_regions.append(reg)
    return ts


def gen_stage_fixture(out_path: str, seed: int = 77, n_pairs: int = 700):
    """Whole-stage known answer: the reference's own decombinator() (decombine.py:881-1202) on a
    synthetic paired FASTQ, for bc_read R2 and R1, orientation reverse and both.  Stores the
    FASTQ text, the returned rows and the summary-file body."""
    import glob
    import io as _io
    import contextlib
    ts = synth.make_tagset("human", "extended", "b", n_v=20, n_j=8, seed=21, n_shared_groups=3)
    rng = np.random.default_rng(seed)
    r1, r2 = [], []
    for i in range(n_pairs):
        sense = casegen.mixture_read(ts, rng, 150, sub_rate=0.01, n_rate=0.03)
        read = casegen.revcomp(sense) if i % 7 else sense            # a few sense-strand reads for `both`
        if i % 50 == 3:
            read = read[:int(rng.integers(30, 149))]                  # ragged lengths
        q1 = "".join(chr(int(c)) for c in rng.integers(35, 74, size=len(read)))
        bc = casegen.rand_seq(rng, 42)
        if i % 40 == 5:
            bc = bc[:10] + "N" + bc[11:]
        tail = casegen.rand_seq(rng, 108)
        q2 = "".join(chr(int(c)) for c in rng.integers(35, 74, size=150))
        name = f"SYN:{i}:{int(rng.integers(1000, 9999))}"
        r1.append(f"@{name} 1:N:0:AAAA\n{read}\n+\n{q1}\n")
        r2.append(f"@{name} 2:N:0:AAAA\n{bc + tail}\n+\n{q2}\n")
    fq1, fq2 = "".join(r1), "".join(r2)
    runs = []
    m = ref_driver.module()
    with tempfile.TemporaryDirectory() as td:
        tagdir = os.path.join(td, "tags"); ts.write(tagdir)
        open(os.path.join(td, "SYNTH_1.fq"), "w").write(fq1)
        open(os.path.join(td, "SYNTH_2.fq"), "w").write(fq2)
        for bc_read, orientation, allow in (("R2", "reverse", False), ("R2", "both", True), ("R1", "reverse", False)):
            outdir = os.path.join(td, f"out_{bc_read}_{orientation}") + os.sep
            os.makedirs(outdir)
            args = dict(infile=os.path.join(td, "SYNTH_1.fq"), chain="b", bc_read=bc_read, suppresssummary=False,
                        dontgzip=True, dontcheck=False, dontcount=True, extension="n12", prefix="dcr_",
                        orientation=orientation, tags="extended", species="human", allowNs=allow, lenthreshold=130,
                        tagfastadir=tagdir, nobarcoding=False, bclength=42, outpath=outdir, dontsave=False,
                        command="decombine", sampling_analysis=False)
            with contextlib.redirect_stdout(_io.StringIO()):
                rows = m.decombinator(dict(args))
            logs = glob.glob(outdir + "Logs/*.csv")
            assert len(logs) == 1
            body = open(logs[0]).read().split("\n")
            keep = [ln for ln in body if not ln.startswith(("Directory,", "DateFinished,", "TimeFinished,", "TimeTaken"))]
            runs.append({"bc_read": bc_read, "orientation": orientation, "allowNs": allow, "rows": rows,
                         "summary_lines": keep, "log_name_tail": os.path.basename(logs[0]).split("_", 3)[3]})
            print(f"stage fixture {bc_read}/{orientation}: {len(rows)} rows")
    with open(out_path, "w") as f:
        json.dump({"generator": "oracle/gen_golden.py gen_stage_fixture",
                   "source": "reference decombinator() (decombine.py:881-1202), unmodified, + oracle/refshim stand-ins",
                   "tagset": tagset_dict(ts), "fastq_r1": fq1, "fastq_r2": fq2, "runs": runs}, f, separators=(",", ":"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bulk", type=int, default=0)
    args = ap.parse_args()
    if not ref_driver.available():
        sys.exit("reference tree not present: golden vectors can only be generated in the build container")
    os.makedirs(GOLDEN, exist_ok=True)

    sets = {
        # small sets keep the fixtures compact; structure as in SURVEY.md §8(d)
        "human_original_b": (synth.make_tagset("human", "original", "b", n_v=24, n_j=13, seed=11), 101, 1500),
        "human_extended_a": (synth.make_tagset("human", "extended", "a", n_v=30, n_j=20, seed=12,
                                               n_shared_groups=4, lowercase_fasta=True), 102, 1000),
        "mouse_original_g": (synth.make_tagset("mouse", "original", "g", n_v=12, n_j=4, seed=13,
                                               n_shared_groups=2), 103, 800),
        "edge_extended_b": (make_edge_tagset(), 104, 600),
    }
    bad = 0
    for name, (ts, seed, n_mix) in sets.items():
        with tempfile.TemporaryDirectory() as td:
            cases = build_cases(ts, seed, n_mix, sub_rate=0.02 if "mouse" in name else 0.005)
            res = run_cases(ts, cases, td)
            summarize(name, res)
            with open(os.path.join(GOLDEN, f"dcr_{name}.json"), "w") as f:
                json.dump({"generator": "oracle/gen_golden.py", "source":
                           "reference src/decombinator/decombine.py (unmodified) + oracle/refshim stand-ins",
                           "tagset": tagset_dict(ts), "cases": res}, f, separators=(",", ":"))
            if args.bulk:
                bad += bulk_crosscheck(ts, td, args.bulk, seed + 1000)
    gen_stage_fixture(os.path.join(GOLDEN, "stage_human_extended_b.json"))
    if args.bulk:
        print("bulk cross-check mismatches:", bad)
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
