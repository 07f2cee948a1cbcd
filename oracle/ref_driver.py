"""Imports the reference's decombine.py UNMODIFIED in the build container.

TEST INFRASTRUCTURE, container-only: needs /root/reference, which does not
exist on the GPU box.  Used by oracle/gen_golden.py (to produce the committed
fixtures under tests/golden/) and by tests that are skipped when the reference
tree is absent.

The three wheels the reference imports at decombine.py:108-111 (acora, Bio,
Levenshtein) are not installable offline; oracle/refshim/ holds stand-ins
restating their published behaviour, and importlib.metadata.version is patched
because the package is not installed (decombine.py:884).
"""
from __future__ import annotations

import collections
import contextlib
import importlib
import io
import os
import sys

REFERENCE_SRC = "/root/reference/src"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "decombinator", "decombine.py"))


_mod = None


def module():
    """The reference's decombinator.decombine module object."""
    global _mod
    if _mod is None:
        if not available():
            raise RuntimeError("reference tree not present")
        for p in (_SHIM, REFERENCE_SRC):
            if p not in sys.path:
                sys.path.insert(0, p)
        from importlib import metadata
        _orig = metadata.version

        def _version(name):
            if name == "decombinator":
                return "5.0.0.dev0"
            return _orig(name)

        metadata.version = _version
        _mod = importlib.import_module("decombinator.decombine")
    return _mod


class RefChain:
    """import_tcr_info + dcr for one chain, driven exactly as the reference's
    read loop drives them (decombine.py:889, :998-1013)."""

    def __init__(self, tagdir: str, species: str, tags: str, chain: str):
        self.m = module()
        self.args = {
            "infile": "synthetic.fq", "chain": chain, "tags": tags, "species": species,
            "tagfastadir": tagdir, "allowNs": False, "lenthreshold": 130,
            "orientation": "reverse",
        }
        with contextlib.redirect_stdout(io.StringIO()):
            self.m.import_tcr_info(self.args)
        # snapshot of the tables for inspection
        self.v_seqs = list(self.m.v_seqs)
        self.j_seqs = list(self.m.j_seqs)

    def reload(self):
        with contextlib.redirect_stdout(io.StringIO()):
            self.m.import_tcr_info(self.args)

    def counts(self) -> collections.Counter:
        return self.m.counts

    def dcr(self, read: str, allow_ns: bool = False, lenthreshold: int = 130):
        """dcr(read, inputargs) on the frame as given; returns (result, counter delta)."""
        self.args["allowNs"] = allow_ns
        self.args["lenthreshold"] = lenthreshold
        before = collections.Counter(self.m.counts)
        out = self.m.dcr(read, self.args)
        after = self.m.counts
        delta = {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}
        return out, delta

    def decombine_read(self, vdj: str, orientation: str = "reverse", allow_ns: bool = False,
                       lenthreshold: int = 130):
        """The orientation dispatch of the read loop, decombine.py:998-1013."""
        m = self.m
        self.args["allowNs"] = allow_ns
        self.args["lenthreshold"] = lenthreshold
        before = collections.Counter(m.counts)
        m.counts["read_count"] += 1
        if orientation == "reverse":
            recom = m.dcr(m.revcomp(vdj), self.args)
            frame = "reverse"
        elif orientation == "forward":
            recom = m.dcr(vdj, self.args)
            frame = "forward"
        else:
            recom = m.dcr(m.revcomp(vdj), self.args)
            frame = "reverse"
            if not recom:
                recom = m.dcr(vdj, self.args)
                frame = "forward"
        if recom:
            m.counts["vj_count"] += 1
        after = m.counts
        delta = {k: after[k] - before.get(k, 0) for k in after if after[k] != before.get(k, 0)}
        return recom, frame, delta
