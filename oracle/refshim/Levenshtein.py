"""Stand-in for Levenshtein==0.25.1 (reference pyproject.toml:20).  TEST
INFRASTRUCTURE for oracle/gen_golden.py only.  hamming(): number of differing
positions; a length difference adds that many (rapidfuzz pads by default)."""


def hamming(s1, s2, *, pad=True, processor=None, score_cutoff=None):
    if len(s1) != len(s2) and not pad:
        raise ValueError("Sequences are not the same length.")
    m = min(len(s1), len(s2))
    return sum(1 for a, b in zip(s1[:m], s2[:m]) if a != b) + max(len(s1), len(s2)) - m
