"""Front half of the reference's `collapse` stage, per `.n12` row: spacer
search, UMI (barcode) extraction, barcode quality filter, inter-tag length filter — everything
`read_in_data` does to a row before it starts grouping rows (reference
src/decombinator/collapse.py:482-565, with get_barcode_positions :367-479, set_barcode :278-326,
check_umi_quality :343-353 and the spacer searches :192-236).  Grouping and clustering (the
stateful rest of the stage) are outside this build.

read_in_rows() / read_in_data() run the batch entry of the library (dcrx_collapse_front,
csrc/dcrx_collapse.cpp: threaded C++ over the row text libdcrx assembled): spacers verbatim or with up
to two substitutions are decided there, with the reference's counters.  A row that reaches the
reference's indel search (`{2i+2d+1s<=2}`, :198-201 — about one row in a thousand of real data)
comes back undecided and goes through the per-row functions below, which are the reference's own
`regex` patterns (regex is a pinned dependency of the reference, pyproject.toml): same values row for
row.  The per-row functions keep the reference's names, arguments and counter keys, so that its tests
read the same here.
"""
from __future__ import annotations

import collections as coll
import collections.abc

import regex

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


_SUBS = {}
_INDEL = {}


def _findall_exact(subseq: str, seq: str):
    """regex.findall(subseq, seq) for a plain DNA word: its non-overlapping occurrences, left to right."""
    out, at, n = [], seq.find(subseq), len(subseq)
    while at >= 0:
        out.append(subseq)
        at = seq.find(subseq, at + n)
    return out


def findSubs(subseq, seq):
    """collapse.py:192-195: up to two substitutions."""
    pat = _SUBS.get(subseq)
    if pat is None:
        pat = _SUBS[subseq] = regex.compile("(" + subseq + "){1s<=2}")
    return pat.findall(seq)


def findSubsInsOrDels(subseq, seq):
    """collapse.py:198-201."""
    pat = _INDEL.get(subseq)
    if pat is None:
        pat = _INDEL[subseq] = regex.compile("(" + subseq + "){2i+2d+1s<=2}")
    return pat.findall(seq)


def spacerSearch(subseq, seq):
    """collapse.py:204-212: exact, then substitutions, then substitutions or an indel."""
    found = _findall_exact(subseq, seq)
    if not found:
        found = findSubs(subseq, seq)
    if not found:
        found = findSubsInsOrDels(subseq, seq)
    return found


def findFirstSpacer(oligo, seq, oligo_start, oligo_end):
    return list(spacerSearch(oligo["spcr1"], seq[oligo_start:oligo_end]))          # collapse.py:215-219


def findSecondSpacer(oligo, seq):
    return list(spacerSearch(oligo["spcr2"], seq[len(oligo["spcr1"]):]))           # collapse.py:222-227


def getSpacerPositions(bcseq, spacers):
    """collapse.py:230-237 (the search start advances by the spacers' lengths only: reproduced)."""
    positions, startpos = [], 0
    for x in spacers:
        positions.append(bcseq.find(x, startpos))
        startpos += len(x)
    return positions


def filterShortandLongBarcodes(b1len, b2end, bcseq, counts):
    """collapse.py:240-253."""
    if b1len <= 3:
        counts["getbarcode_fail_n1tooshort"] += 1
        return False
    if b1len >= 9:
        counts["getbarcode_fail_n1toolong"] += 1
        return False
    if b2end > len(bcseq):
        counts["getbarcode_fail_n2pastend"] += 1
        return False
    return True


def logExactOrRegexMatch(spacers, oligo, counts):
    if spacers == list(oligo.values()):                                             # collapse.py:256-260
        counts["getbarcode_pass_exactmatch"] += 1
    else:
        counts["getbarcode_pass_regexmatch"] += 1


def logFuzzyMatching(b1len, bclength, spacers, oligo, counts):
    """collapse.py:263-275."""
    exact = spacers == list(oligo.values())
    if b1len == bclength and not exact:
        counts["getbarcode_pass_fuzzymatch_rightlen"] += 1
    elif b1len in [4, 5] and not exact:
        counts["getbarcode_pass_fuzzymatch_short"] += 1
    elif b1len >= 7 and not exact:
        counts["getbarcode_pass_fuzzymatch_long"] += 1
    elif b1len == bclength:
        counts["getbarcode_pass_other"] += 1


def get_barcode_positions(bcseq, inputargs, counts):
    """collapse.py:367-479: start/stop of N1 (and N2) in the barcode region, or None."""
    name = str.lower(inputargs["oligo"])
    if name not in ("i8", "i8_single", "m13", "nebio", "takara"):
        raise ValueError("The flag for the -ol input must be one of M13, I8, I8_single, NEBIO, or TAKARA.")
    if "N" in bcseq and inputargs["allowNs"] == False:  # noqa: E712  (:390-394)
        counts["getbarcode_fail_N"] += 1
        return None
    oligo = getOligo(name)
    if name == "nebio":
        oligo_start, oligo_end = 18, 28
    elif name == "takara":
        oligo_start, oligo_end = 0, 19
    else:
        oligo_start, oligo_end = 0, 10 + len(oligo["spcr1"])
    spacers = findFirstSpacer(oligo, bcseq, oligo_start, oligo_end)
    if not len(spacers) == 1:                                                       # :413-416
        counts["getbarcode_fail_nospacerfound"] += 1
        return None
    if name not in ("i8_single", "nebio", "takara"):
        spacers += findSecondSpacer(oligo, bcseq)
        if not len(spacers) == 2:                                                   # :423-426
            counts["getbarcode_fail_not2spacersfound"] += 1
            return None
    spacer_positions = getSpacerPositions(bcseq, spacers)
    if name in ("nebio", "takara"):
        bclength = 17 if name == "nebio" else 12
        b1start, b1end = 0, bclength
        logExactOrRegexMatch(spacers, oligo, counts)
        logFuzzyMatching(b1end - b1start, bclength, spacers, oligo, counts)
        return [b1start, b1end]
    bclength = 6
    if name == "i8_single":
        b1start = 0
        b1end = spacer_positions[0]
        b2start = spacer_positions[0] + len(spacers[0])
    else:
        b1start = spacer_positions[0] + len(spacers[0])
        b1end = spacer_positions[1]
        b2start = spacer_positions[1] + len(spacers[1])
    b2end = b2start + bclength
    b1len = b1end - b1start
    if not filterShortandLongBarcodes(b1len, b2end, bcseq, counts):
        return None
    logExactOrRegexMatch(spacers, oligo, counts)
    logFuzzyMatching(b1len, bclength, spacers, oligo, counts)
    return [b1start, b1end, b2start, b2end]


def set_barcode(fields, bc_locs, inputargs):
    """collapse.py:278-326: the barcode and its quality string from row fields 8 and 9; an N1 of other
    than six bases is padded with 'S' / cut to five bases + 'L' (quality '?')."""
    if str.lower(inputargs["oligo"]) in ["nebio", "takara"]:
        return fields[8][bc_locs[0]:bc_locs[1]], fields[9][bc_locs[0]:bc_locs[1]]
    n1 = bc_locs[1] - bc_locs[0]
    if n1 == 6:
        barcode = fields[8][bc_locs[0]:bc_locs[1]] + fields[8][bc_locs[2]:bc_locs[3]]
        qual = fields[9][bc_locs[0]:bc_locs[1]] + fields[9][bc_locs[2]:bc_locs[3]]
    elif n1 < 6:
        barcode = fields[8][bc_locs[0]:bc_locs[1]] + "S" * (6 - n1) + fields[8][bc_locs[2]:bc_locs[3]]
        qual = fields[9][bc_locs[0]:bc_locs[1]] + "?" * (6 - n1) + fields[9][bc_locs[2]:bc_locs[3]]
        counts["readdata_short_barcode"] += 1
    else:
        barcode = fields[8][bc_locs[0]:bc_locs[0] + 5] + "L" + fields[8][bc_locs[2]:bc_locs[3]]
        qual = fields[9][bc_locs[0]:bc_locs[0] + 5] + "?" * (6 - n1) + fields[9][bc_locs[2]:bc_locs[3]]   # a negative count: no '?' at all (reproduced)
        counts["readdata_long_barcode"] += 1
    return barcode, qual


def get_qual_scores(qualstring):
    return [ord(x) - 33 for x in qualstring]                                        # collapse.py:332-335


def check_umi_quality(qualstring, parameters):
    """collapse.py:343-353: True when the barcode FAILS the check (the reference's sense)."""
    q = get_qual_scores(qualstring)
    number_below_min = sum([x < parameters[0] for x in q])
    average_quality = sum(q) / len(q)
    return number_below_min > parameters[1] or average_quality < parameters[2]


def _row_front(line, inputargs, barcode_quality_parameters):
    """One row through the per-row functions (the reference's loop body, :540-563)."""
    counts["readdata_input_dcrs"] += 1
    bc_locs = get_barcode_positions(line[8], inputargs, counts)
    if not bc_locs:
        counts["readdata_fail_no_bclocs"] += 1
        return None
    barcode, qual = set_barcode(line, bc_locs, inputargs)
    if check_umi_quality(qual, barcode_quality_parameters):
        counts["readdata_fail_low_barcode_quality"] += 1
        return None
    if len(line[6]) > inputargs["lenthreshold"]:
        counts["readdata_fail_overlong_intertag_seq"] += 1
        return None
    counts["readdata_success"] += 1
    return (barcode, qual, line[:5], line[6], line[7], line[5])


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
    decided = {}
    import numpy as np
    for k in np.nonzero(rows["status"] == nat.CF_DEFER)[0].tolist():      # the indel search (and malformed rows): the per-row functions
        line = text[int(offs[k]):int(offs[k + 1])].decode("utf-8", "replace").rstrip("\n").split(", ")
        decided[k] = _row_front(line, inputargs, barcode_quality_parameters)
    return FrontRows(text, rows, offs, decided)


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
