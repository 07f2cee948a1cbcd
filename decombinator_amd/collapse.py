"""Front half of the reference's `collapse` stage, per `.n12` row: spacer
search, UMI (barcode) extraction, barcode quality filter, inter-tag length filter — everything
`read_in_data` does to a row before it starts grouping rows (reference
src/decombinator/collapse.py:482-565, with get_barcode_positions :367-479, set_barcode :278-326,
check_umi_quality :343-353 and the spacer searches :192-236).  Grouping and clustering (the
stateful rest of the stage) are outside this build.

read_in_rows() / read_in_data() run the batch entry of the library (dcrx_collapse_front,
csrc/dcrx_collapse.cpp: threaded C++ over the row text libdcrx assembled): all three spacer searches
of the reference — verbatim, up to two substitutions, the indel form `{2i+2d+1s<=2}` (:198-201) — are
decided there, with the reference's counters.  The per-row functions the reference's tests call
(getOligo, spacerSearch, findFirstSpacer, findSecondSpacer, get_barcode_positions) keep their names,
arguments and counter keys and run on the same native code (dcrx_spacer_search, a one-row batch).
The reference's own `regex` patterns live in tests/collapse_regex_ref.py, as the differential checker.
"""
from __future__ import annotations

import collections as coll
import collections.abc

counts = coll.Counter()

OLIGOS = {
    "m13": {"spcr1": "GTCGTGACTGGGAAAACCCTGG", "spcr2": "GTCGTGAT"},
    "i8": {"spcr1": "GTCGTGAT", "spcr2": "GTCGTGAT"},
    "i8_single": {"spcr1": "ATCACGAC"},
    "nebio": {"spcr1": "TACGGG"},
    "takara": {"spcr1": "GTACGGG"},
}


def getOligo(oligo_name):
    """collapse.py:174-189."""
    if oligo_name.lower() not in OLIGOS:
        print("Error: Failed to recognise oligo name. Please choose from " + str(list(OLIGOS.keys())))
        raise SystemExit
    return OLIGOS[oligo_name.lower()]


def spacerSearch(subseq, seq):
    """collapse.py:204-212: the matches of `subseq` in `seq`, verbatim, else with up to two substitutions, else with one inserted
    or deleted base — the strings regex.findall returns, found by libdcrx (dcrx_spacer_search)."""
    from . import _native as nat
    return [seq[a:a + n] for a, n in nat.spacer_search(seq, subseq)[0]]


def findFirstSpacer(oligo, seq, oligo_start, oligo_end):
    """collapse.py:215-219."""
    return spacerSearch(oligo["spcr1"], seq[oligo_start:oligo_end])


def findSecondSpacer(oligo, seq):
    """collapse.py:222-227."""
    return spacerSearch(oligo["spcr2"], seq[len(oligo["spcr1"]):])


def get_barcode_positions(bcseq, inputargs, counts):
    """collapse.py:367-479 for one barcode region: [b1start, b1end(, b2start, b2end)] or None, the reference's getbarcode_*
    keys bumped in `counts` — a one-row batch through dcrx_collapse_front."""
    from . import _native as nat
    name = str.lower(inputargs["oligo"])
    if name not in nat.COLLAPSE_OLIGOS:
        raise ValueError("The flag for the -ol input must be one of M13, I8, I8_single, NEBIO, or TAKARA.")
    if not bcseq.isascii() or "," in bcseq or "\n" in bcseq:
        raise ValueError("barcode region is not plain sequence text")
    row = ", ".join(["0", "0", "0", "0", "A", "id", "A", "I", bcseq, "I" * len(bcseq)]) + "\n"
    rows, _, cnt = nat.collapse_front(row.encode("ascii"), name, inputargs["allowNs"], 1 << 30, [0, len(bcseq) + 1, 0], n_threads=1)
    for key, v in zip(nat.COLLAPSE_COUNTERS, cnt.tolist()):
        if v and key.startswith("getbarcode_"):
            counts[key] += int(v)
    r = rows[0]
    if r["b1start"] < 0 and r["b1end"] < 0:
        return None
    locs = [int(r["b1start"]), int(r["b1end"])]
    return locs if name in ("nebio", "takara") else locs + [int(r["b2start"]), int(r["b2end"])]


def _rows_text(data) -> bytes:
    """The rows as `.n12` text (fields joined by ", ", one row per line): what libdcrx's batch entry reads."""
    chunks = getattr(data, "_tagged_chunks", None)
    if chunks is not None:                                   # the rows decombinator() returned: already text
        return b"".join(blob for _, blob, _ in chunks())
    out = []
    for line in data:
        if isinstance(line, (bytes, bytearray)):
            line = line.decode("utf-8", "replace")
        out.append(line.rstrip("\n") if isinstance(line, str) else ", ".join(str(x) for x in line))
    return ("\n".join(out) + ("\n" if out else "")).encode("utf-8")


class FrontRows(coll.abc.Sequence):
    """What the front half leaves of each input row, in input order: None for a dropped row, else
    (barcode, barcode_qualstring, dcr, seq, seq_qualstring, seq_id) — built from the library's per-row records and the row
    text only when asked for (a run has millions of rows; `status`, `barcode` and `kept()` give the columns at once)."""

    def __init__(self, text, rows, offsets, decided):
        self._text, self._rows, self._off, self._decided = text, rows, offsets, decided       # decided: {row index: entry} of the rows the regex path settled
        self.status = rows["status"]

    def __len__(self):
        return len(self._rows)

    def __eq__(self, other):
        if isinstance(other, (list, FrontRows)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def kept(self):
        """Indices of the rows that passed."""
        import numpy as np
        ok = self._rows["status"] == 0
        for k, v in self._decided.items():
            ok[k] = v is not None
        return np.nonzero(ok)[0]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if k in self._decided:
            return self._decided[k]
        r = self._rows[k]
        if r["status"] != 0:
            return None
        line = self._text[int(self._off[k]):int(self._off[k + 1])].decode("utf-8", "replace").rstrip("\n").split(", ")
        return (r["barcode"][:r["barcode_len"]].decode("ascii"), r["barcode_qual"][:r["barcode_qual_len"]].decode("ascii"),
                line[:5], line[6], line[7], line[5])


def read_in_rows(data, inputargs, barcode_quality_parameters, n_threads: int = 0):
    """What read_in_data (collapse.py:482-565) does to each row before grouping, over a whole batch.  `data`: the rows
    decombinator() returned (N12Rows), lists of fields, or `.n12` lines.  Returns a FrontRows (one entry per input row:
    None for a dropped row, else (barcode, barcode_qualstring, dcr, seq, seq_qualstring, seq_id)); the module's `counts`
    receives the reference's keys."""
    from . import _native as nat
    name = str.lower(inputargs["oligo"])
    if name not in nat.COLLAPSE_OLIGOS:
        raise ValueError("The flag for the -ol input must be one of M13, I8, I8_single, NEBIO, or TAKARA.")
    text = _rows_text(data)
    rows, offs, cnt = nat.collapse_front(text, name, inputargs["allowNs"], inputargs["lenthreshold"], barcode_quality_parameters,
                                         n_threads=n_threads)
    for key, v in zip(nat.COLLAPSE_COUNTERS, cnt.tolist()):
        if v:
            counts[key] += int(v)
    import numpy as np
    bad = np.nonzero(rows["status"] == nat.CF_DEFER)[0]
    if len(bad):      # not ten fields of ASCII text: the reference would fail on such a row too (an IndexError / a wrong column)
        k = int(bad[0])
        raise ValueError(f"row {k} is not a ten-field ASCII `.n12` row: " + text[int(offs[k]):int(offs[k + 1])].decode("utf-8", "replace")[:200])
    return FrontRows(text, rows, offs, {})


def read_in_data(data, inputargs, barcode_quality_parameters, lev_threshold_fraction=None, dont_count=True, opener=None):
    """The reference's entry (collapse.py:482-565) as far as this build goes: the rows' front half.  With
    inputargs["command"] == "collapse", `data` is the path of an `.n12` file (opened with `opener`, as the reference does);
    otherwise the rows decombinator() returned.  lev_threshold_fraction and the grouping it steers are outside this build."""
    if inputargs.get("command") == "collapse":
        fh = (opener or open)(data, "rt")
        try:
            data = fh.read().splitlines()
        finally:
            fh.close()
    if not data:
        raise ValueError("No reads found in input file. Check .n12 and log files for errors.")       # :508-511
    if not dont_count:
        print("Reading data in...")
    return read_in_rows(data, inputargs, barcode_quality_parameters)
