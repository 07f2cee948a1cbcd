#!/usr/bin/env python3
"""Golden vectors for the FASTQ/FASTA record reader: the records the REFERENCE's readfq
(src/decombinator/decombine.py:228-265) yields over its own opener (text mode, :118-123)
for a set of small files.  Container-only (imports /root/reference); the fixture
tests/golden/readfq_cases.json holds the file texts and the expected records only.
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_driver  # noqa: E402

CASES = {
    "plain": "@r1 desc\nACGT\n+\nIIII\n@r2\nAC\nGT\n+r2\nII\nII\n",
    "fasta_mix": ">f1 x\nACGT\nAC\n>f2\nTT\n@q\nAA\n+\nII\n",
    "no_final_newline": "@r1\nACGT\n+\nIIII",
    "no_final_newline_long_qual": "@r1\nACGT\n+\nIIIII",
    "crlf": "@r1 a\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nAC\r\n+\r\nII\r\n",
    "cr_only": "@r1 a\rACGT\r+\rIIII\r",
    "leading_junk": "junk\n\n@r1\nAC\n+\n@I\n@r2\nGG\n+\nII\n",
    "empty_file": "",
    "empty_sequence": "@r1\n+\nII\n@r2\nAC\n+\nII\n",
    "truncated_quality": "@r1\nACGT\n+\nII\n",
    "header_only": "@r1\n",
    "lone_at": "@",
    "lone_at_after_record": "@r1\nAC\n+\nII\n@",
    "lone_plus_at_eof": "@r1\nAC\n+",
    "blank_lines": "@r1\nAC\n\nGT\n+\nIIII\n\n@r2\nA\n+\nI\n",
    "tab_in_header": "@r1\tx y\nAC\n+\nII\n",
    "quality_starts_with_at": "@r1\nACGT\n+\n@@@@\n@r2\nAC\n+\n@+\n",
    "long_quality": "@r1\nAC\n+\nIIIIII\n@r2\nGG\n+\nII\n",
    "lowercase_and_n": "@r1\nacgtNNRY\n+\nIIIIIIII\n",
}


def random_case(rng: random.Random) -> str:
    out = []
    for k in range(rng.randint(1, 12)):
        fasta = rng.random() < 0.2
        n = rng.randint(0, 40)
        seq = "".join(rng.choice("ACGTN") for _ in range(n))
        width = rng.choice([5, 7, 1000])
        lines = [seq[i:i + width] for i in range(0, len(seq), width)] or ([""] if rng.random() < 0.5 else [])
        nl = rng.choice(["\n", "\n", "\r\n"])
        out.append((">" if fasta else "@") + f"id{k}" + rng.choice(["", " extra words", "/1"]) + nl)
        out.extend(x + nl for x in lines)
        if not fasta:
            qual = "".join(rng.choice("!I@+>#5") for _ in range(n))
            qlines = [qual[i:i + width] for i in range(0, len(qual), width)] or [""]
            out.append("+" + nl)
            out.extend(x + nl for x in qlines)
    text = "".join(out)
    if rng.random() < 0.3 and text:
        text = text[:-rng.randint(1, min(6, len(text)))]
    return text


def main():
    m = ref_driver.module()
    rng = random.Random(20240607)
    cases = dict(CASES)
    for i in range(40):
        cases[f"random_{i:02d}"] = random_case(rng)
    fixture = []
    with tempfile.TemporaryDirectory() as td:
        for name, text in cases.items():
            p = os.path.join(td, "x.fq")
            with open(p, "wb") as f:
                f.write(text.encode())
            opener = m.opener_check({"infile": p})
            with opener(p, "rt") as fh:
                recs = [list(r) for r in m.readfq(fh)]
            fixture.append({"name": name, "text": text, "records": recs})
    out = os.path.join(ROOT, "tests", "golden", "readfq_cases.json")
    json.dump({"generator": "oracle/gen_readfq_golden.py",
               "source": "reference decombine.py readfq (:228-265) over opener_check (:118-123)",
               "cases": fixture}, open(out, "w"), separators=(",", ":"))
    print(len(fixture), "cases ->", out, sum(len(c["records"]) for c in fixture), "records")


if __name__ == "__main__":
    main()
