#!/usr/bin/env python3
"""usage: tools/fuzz_spacer_search.py [N per oligo, default 1000000] [seed] — dcrx_spacer_search (csrc/dcrx_collapse.cpp: the
three searches of spacerSearch, collapse.py:204-212, decided natively) against the `regex` module's findall on mutated barcode
regions, N per oligo spacer: insertions, deletions, substitutions, repeats, truncations, and — for a fifth of the cases — random
patterns over two- and three-letter alphabets (runs and repeats: the alignments among which the regex engine's backtracking order
chooses).  Container or GPU box (host code only).  Prints the number of cases whose matches came from the indel stage."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decombinator_amd import collapse          # noqa: E402
from tests import collapse_regex_ref as ref    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    spacers = sorted({v for o in collapse.OLIGOS.values() for v in o.values()})
    t0 = time.time()
    for sp0 in spacers:
        n_indel = n_sub = n_none = 0
        for it in range(n):
            sp = sp0
            if rng.random() < 0.2:
                alpha = "AC" if rng.random() < 0.5 else "ACG"
                sp = "".join(rng.choice(alpha) for _ in range(rng.randrange(4, 9)))
                s = "".join(rng.choice(alpha) for _ in range(rng.randrange(3, 30)))
            else:
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
            if got != want:
                print("DIFF", sp, s, "regex:", want, "native:", got)
                sys.exit(1)
            if not want:
                n_none += 1
            elif len(want[0]) != len(sp):
                n_indel += 1
            elif want[0] != sp:
                n_sub += 1
        print(f"{sp0}: {n} cases equal ({n_indel} decided by the indel stage, {n_sub} by substitutions, {n_none} without a match), {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
