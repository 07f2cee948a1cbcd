"""Stand-in for polyleven (pinned 0.8, reference pyproject.toml): only what the reference's
collapse.py calls.  TEST INFRASTRUCTURE (oracle/): lets the build container import the reference's
collapse.py unmodified to generate fixtures; the wheel cannot be installed offline."""


def levenshtein(a, b, k=None):
    """Edit distance (insertions, deletions, substitutions, unit costs)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    d = prev[-1]
    if k is not None and d > k:
        return k + 1
    return d
