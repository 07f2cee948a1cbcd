#!/usr/bin/env python3
"""Whole-stage known answers for input variants the first stage fixture does not have:
gzipped input, wrapped (multi-line) records with CRLF line ends, files of unequal length,
an odd record count in R1 mode, sampling_analysis, forward orientation, other
bclength / lenthreshold.  The reference's own decombinator() (decombine.py:881-1202) is run
on files derived from tests/golden/stage_human_extended_b.json; container-only.
Output: tests/golden/stage_variants.json (file texts, arguments, returned rows, counters).
"""
from __future__ import annotations

import contextlib
import gzip
import io
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decombinator_amd import synth  # noqa: E402
from oracle import ref_driver  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def records(text):
    lines = text.split("\n")
    return [lines[i:i + 4] for i in range(0, len(lines) - 1, 4)]


def wrap(rec, width, eol):
    name, seq, plus, qual = rec
    out = [name]
    out += [seq[i:i + width] for i in range(0, len(seq), width)] or [""]
    out.append(plus)
    out += [qual[i:i + width] for i in range(0, len(qual), width)] or [""]
    return eol.join(out) + eol


def main():
    base = json.load(open(os.path.join(GOLDEN, "stage_human_extended_b.json")))
    tsd = base["tagset"]
    ts = synth.TagSet(species=tsd["species"], tags=tsd["tags"], chain=tsd["chain"], v_tags=tsd["v_tags"],
                      v_jumps=tsd["v_jumps"], v_names=tsd["v_names"], v_regions=tsd["v_regions"], j_tags=tsd["j_tags"],
                      j_jumps=tsd["j_jumps"], j_names=tsd["j_names"], j_regions=tsd["j_regions"])
    r1, r2 = records(base["fastq_r1"])[:260], records(base["fastq_r2"])[:260]
    plain1 = "".join("\n".join(r) + "\n" for r in r1)
    plain2 = "".join("\n".join(r) + "\n" for r in r2)
    variants = [
        # name, file-1 text, file-2 text, gz, args
        ("gz_sampling_short_r2", plain1, "".join("\n".join(r) + "\n" for r in r2[:-7]), True,
         dict(bc_read="R2", orientation="reverse", allowNs=False, sampling_analysis=True, bclength=42, lenthreshold=130)),
        ("short_r1_both", "".join("\n".join(r) + "\n" for r in r1[:-11]), plain2, False,
         dict(bc_read="R2", orientation="both", allowNs=False, sampling_analysis=False, bclength=30, lenthreshold=60)),
        ("r1_mode_wrapped_crlf_odd", "".join(wrap(r, 60, "\r\n") for r in r1[:-1]), None, False,
         dict(bc_read="R1", orientation="forward", allowNs=True, sampling_analysis=True, bclength=6, lenthreshold=130)),
        ("r1_mode_no_final_newline", plain1[:-1], None, True,
         dict(bc_read="R1", orientation="both", allowNs=False, sampling_analysis=False, bclength=0, lenthreshold=130)),
    ]
    m = ref_driver.module()
    runs = []
    with tempfile.TemporaryDirectory() as td:
        tagdir = os.path.join(td, "tags"); ts.write(tagdir)
        for name, t1, t2, gz, extra in variants:
            d = os.path.join(td, name); os.makedirs(d)
            ext = ".fq.gz" if gz else ".fq"
            op = gzip.open if gz else open
            with op(os.path.join(d, "VAR_1" + ext), "wb") as f:
                f.write(t1.encode())
            if t2 is not None:
                with op(os.path.join(d, "VAR_2" + ext), "wb") as f:
                    f.write(t2.encode())
            args = dict(infile=os.path.join(d, "VAR_1" + ext), chain="b", suppresssummary=True, dontgzip=True,
                        dontcheck=True, dontcount=True, extension="n12", prefix="dcr_", tags="extended", species="human",
                        tagfastadir=tagdir, nobarcoding=False, outpath=d + os.sep, dontsave=True, command="decombine")
            args.update(extra)
            with contextlib.redirect_stdout(io.StringIO()):
                rows = m.decombinator(dict(args))
            counts = {k: v for k, v in m.counts.items() if isinstance(v, int)}
            runs.append({"name": name, "fastq_r1": t1, "fastq_r2": t2, "gz": gz, "args": extra, "rows": rows,
                         "counts": counts})
            print(f"{name}: {len(rows)} rows, read_count {counts.get('read_count')}")
    json.dump({"generator": "oracle/gen_stage_variants.py",
               "source": "reference decombinator() (decombine.py:881-1202) + oracle/refshim stand-ins",
               "tagset": tsd, "runs": runs}, open(os.path.join(GOLDEN, "stage_variants.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
