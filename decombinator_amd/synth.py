"""Synthetic tag sets in the reference's on-disk formats.

The real tag / germline data (git submodule Decombinator-Tags-FASTAs, reference
.gitmodules:1-3) is not available offline, so benchmarks and parity tests run on
seeded synthetic sets with the same structure:

* `.tags`  : one line per gene, `TAG JUMP NAME` (parsed like reference
             decombine.py:820-866: column 0 = tag, column 1 = integer jump);
* `.fasta` : one record per gene in tag order, header `>acc|NAME|...`
             (decombine.py:683-696; translate.py:188-191 reads field 1).

V regions are 280-340 nt with the 20-nt tag `jump` bases from the 3' end
(jump in {36,39,40,43,44,53}); J regions are 47-66 nt with the tag at offset
`jump` = 20.  Full tags are pairwise >= 3 mismatches apart, but some tags
share an identical half so that the reference's `indices` loops
(decombine.py:298-300, :342-346) see more than one candidate.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

V_JUMPS = (36, 39, 40, 43, 44, 53)


@dataclass
class TagSet:
    """One chain's tables, as import_tcr_info holds them (decombine.py:593-746)."""
    species: str
    tags: str          # "extended" | "original"
    chain: str         # a | b | g | d
    v_tags: List[str] = field(default_factory=list)
    v_jumps: List[int] = field(default_factory=list)
    v_names: List[str] = field(default_factory=list)
    v_regions: List[str] = field(default_factory=list)
    j_tags: List[str] = field(default_factory=list)
    j_jumps: List[int] = field(default_factory=list)
    j_names: List[str] = field(default_factory=list)
    j_regions: List[str] = field(default_factory=list)

    @property
    def half_splits(self):
        # decombine.py:657-661
        return (10, 10) if self.tags == "extended" else (10, 6)

    def file_stem(self, gene: str) -> str:
        # decombine.py:191-201
        return f"{self.species}_{self.tags}_TR{self.chain.upper()}{gene.upper()}"

    def write(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        for gene, tags, jumps, names, regions in (
            ("v", self.v_tags, self.v_jumps, self.v_names, self.v_regions),
            ("j", self.j_tags, self.j_jumps, self.j_names, self.j_regions),
        ):
            stem = os.path.join(directory, self.file_stem(gene))
            with open(stem + ".tags", "w") as f:
                for t, jmp, nm in zip(tags, jumps, names):
                    f.write(f"{t} {jmp} {nm}\n")
            with open(stem + ".fasta", "w") as f:
                for i, (nm, reg) in enumerate(zip(names, regions)):
                    f.write(f">SYN{i:05d}|{nm}|synthetic|F|\n")
                    for o in range(0, len(reg), 60):
                        f.write(reg[o:o + 60] + "\n")


def _rand_seq(rng: np.random.Generator, n: int) -> str:
    return _BASES[rng.integers(0, 4, size=n)].tobytes().decode("ascii")


def _hamming(a: str, b: str) -> int:
    return sum(x != y for x, y in zip(a, b))


def _mutate(rng, s: str, nmut: int) -> str:
    s = list(s)
    for p in rng.choice(len(s), size=nmut, replace=False):
        s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + int(rng.integers(0, 3))) % 4]
    return "".join(s)


def _make_tags(rng, n: int, tag_len: int, split: int, n_shared_groups: int) -> List[str]:
    """n tags of tag_len nt, pairwise Hamming >= 3, with groups sharing a half."""
    tags: List[str] = []

    def ok(t):
        return all(_hamming(t, u) >= 3 for u in tags if len(u) == len(t)) and t not in tags

    # groups: A = h1+h2, B = h1+h2' (shares half1 with A), C = h1''+h2 (shares half2 with A)
    for _ in range(n_shared_groups):
        if len(tags) + 3 > n:
            break
        while True:
            a = _rand_seq(rng, tag_len)
            b = a[:split] + _mutate(rng, a[split:], 4)
            c = _mutate(rng, a[:split], 4) + a[split:]
            trial = [a, b, c]
            if all(ok(t) for t in trial) and _hamming(b, c) >= 3:
                tags.extend(trial)
                break
    while len(tags) < n:
        t = _rand_seq(rng, tag_len)
        if ok(t):
            tags.append(t)
    order = rng.permutation(len(tags))
    return [tags[i] for i in order]


def make_tagset(species: str = "human", tags: str = "original", chain: str = "b",
                n_v: int = 60, n_j: int = 13, seed: int = 20260102, tag_len: int = 20,
                n_shared_groups: int = 3, lowercase_fasta: bool = False) -> TagSet:
    """Seeded synthetic tag set.  Every region contains its own tag exactly once
    at the canonical offset, and no other gene's full tag."""
    rng = np.random.default_rng(seed)
    ts = TagSet(species=species, tags=tags, chain=chain)
    v_split, j_split = ts.half_splits
    ts.v_tags = _make_tags(rng, n_v, tag_len, v_split, n_shared_groups)
    ts.j_tags = _make_tags(rng, n_j, tag_len, j_split, min(n_shared_groups, max(1, n_j // 6)))
    for i, t in enumerate(ts.v_tags):
        jump = int(V_JUMPS[rng.integers(0, len(V_JUMPS))])
        length = int(rng.integers(280, 341))
        left = _rand_seq(rng, length - jump)
        right = _rand_seq(rng, jump - tag_len)
        reg = left + t + right
        ts.v_jumps.append(jump)
        ts.v_names.append(f"TR{chain.upper()}V{i + 1}")
        ts.v_regions.append(reg.lower() if lowercase_fasta else reg)
    for i, t in enumerate(ts.j_tags):
        jump = 20
        length = int(rng.integers(47, 67))
        reg = _rand_seq(rng, jump) + t + _rand_seq(rng, max(0, length - jump - tag_len))
        ts.j_jumps.append(jump)
        ts.j_names.append(f"TR{chain.upper()}J{i + 1}")
        ts.j_regions.append(reg.lower() if lowercase_fasta else reg)
    return ts


# BASELINE.json configs -> synthetic stand-ins (sizes from SURVEY.md §8(d))
def config_tagset(config: int) -> TagSet:
    if config == 2 or config == 4:
        return make_tagset("human", "original", "b", n_v=60, n_j=13, seed=20260101 + config)
    if config == 3:
        raise ValueError("config 3 is two chains: use config3_tagsets()")
    if config == 5:
        return make_tagset("mouse", "original", "g", n_v=12, n_j=4, seed=20260106, n_shared_groups=2)
    raise ValueError(f"no synthetic tag set for config {config}")


def config5_tagsets():
    """Mouse gamma and delta: the two chains of BASELINE config 5 (both take the `original` tag set:
    the reference rewrites inputargs["tags"] for mouse and for gamma/delta, decombine.py:640-654)."""
    return (make_tagset("mouse", "original", "g", n_v=12, n_j=4, seed=20260106, n_shared_groups=2),
            make_tagset("mouse", "original", "d", n_v=16, n_j=2, seed=20260107, n_shared_groups=2))


def config3_tagsets():
    return (make_tagset("human", "extended", "a", n_v=104, n_j=61, seed=20260104, n_shared_groups=5),
            make_tagset("human", "extended", "b", n_v=88, n_j=14, seed=20260105, n_shared_groups=4))
