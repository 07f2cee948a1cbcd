"""Host side of the `decombine` stage: same entry points and argument meaning as
the reference's src/decombinator/decombine.py, with the per-read body of its
read loop (reference decombine.py:998-1013: revcomp -> dcr -> vanalysis /
janalysis -> get_*_deletions) replaced by batched calls into libdcrx.so, the
HIP implementation (decombinator_amd/csrc, through decombinator_amd/_native.py).

What stays on the host, as BASELINE.json's north_star asks: FASTQ reading
(`readfq`), barcode slicing, row assembly for the `.n12` output, the summary
log.  There is no CPU implementation of the hot path here: without the built
library or without a GPU, `decombinator()` raises.

Mirrored reference functions (file:line in /root/reference/src/decombinator/decombine.py):
  opener_check :118   fastq_check :126   revcomp :182   read_tcr_file :187
  readfq :228         import_tcr_info :593   get_v_tags :820   get_j_tags :847
  sort_permissions :869   decombinator :881
"""
from __future__ import annotations

import collections as coll
import collections.abc
import gzip
import itertools
import os
import sys
from time import strftime, time

import numpy as np

from . import _native as nat

__version__ = "0.1.0"

chainnams = {"a": "alpha", "b": "beta", "g": "gamma", "d": "delta"}

# Bio.Seq's ambiguous-DNA complement (both cases, U like T), as reference revcomp() applies it
_COMP = str.maketrans("ACGTMRWSYKVHDBXNUacgtmrwsykvhdbxnu", "TGCAKYWSRMBDHVXNAtgcakywsrmbdhvxna")

# reads sent to the GPU per dcrx_decombine call
BATCH_READS = 1 << 20
# wall seconds of the last decombinator() call by phase: FASTQ read, 2-bit pack, device call (H2D + kernels
# + D2H), row assembly
stage_seconds: dict = {}
stage_info: dict = {}          # how the last stage read its input: sharded_input, byte_ranges (this rank's), rank, world

# module-level state kept for callers that used the reference's globals
counts: coll.Counter = coll.Counter()
_current = None  # the ChainTables of the last import_tcr_info()


def opener_check(inputargs):
    """gzip.open for *.gz inputs, open otherwise (reference :118-123)."""
    return gzip.open if inputargs["infile"].endswith(".gz") else open


def revcomp(read: str) -> str:
    """Reverse complement as Bio.Seq does it (reference :182-184)."""
    return read.translate(_COMP)[::-1]


def sort_permissions(fl):
    """Output files are made world read/writable like the reference does (:869-873)."""
    if oct(os.stat(fl).st_mode)[4:] != "666":
        os.chmod(fl, 0o666)


def readfq(fp):
    """FASTA/FASTQ record generator with the behaviour of the reference's readfq
    (:228-265, Heng Li's reader): yields (name, seq, qual-or-None); name is the
    header without its first character up to the first space; sequence and quality
    may span lines; a FASTQ record whose quality is cut short by EOF comes out as a
    FASTA record."""
    pending = None  # a header line already consumed; an EMPTY one counts as none (the reference tests truthiness)
    while True:
        if not pending:
            for line in fp:
                if line[0] in ">@":
                    pending = line[:-1]
                    break
        if not pending:
            return
        name = pending[1:].partition(" ")[0]
        pending = None
        chunks = []
        for line in fp:
            if line[0] in "@+>":
                pending = line[:-1]
                break
            chunks.append(line[:-1])
        seq = "".join(chunks)
        if not pending or pending[0] != "+":
            yield name, seq, None
            if not pending:
                return
            continue
        pending = None
        qchunks, got = [], 0
        complete = False
        for line in fp:
            qchunks.append(line[:-1])
            got += len(line) - 1
            if got >= len(seq):
                complete = True
                break
        if not complete:
            yield name, seq, None
            return
        yield name, seq, "".join(qchunks)


def _new_summary_file(summaryname, logpath, date, chain_given, chain, samplenam):
    """Opens `summaryname`, or the first free `..._Summary<N>.csv` (reference :141-155, :1082-1095)."""
    if not os.path.exists(summaryname):
        return summaryname, open(summaryname, "wt")
    for i in range(2, 10000):
        name = logpath + date + "_"
        if chain_given:
            name += chainnams[chain] + "_"
        name += samplenam + "_Decombinator_Summary" + str(i) + ".csv"
        if not os.path.exists(name):
            return name, open(name, "wt")
    raise RuntimeError("no free summary file name")


def fastq_check(inputargs, opener, samplenam, summaryname, logpath, chain=None) -> None:
    """Rudimentary FASTQ sanity check (reference :126-179): fewer than four lines is a
    ValueError (after writing the two-line empty-input log); then a '@' header, a '+'
    separator and equal sequence/quality lengths are required."""
    chain = chain if chain is not None else (_current.chain if _current else None)
    with opener(inputargs["infile"], "rt") as possfq:
        head = list(itertools.islice(possfq, 4))
        if len(head) < 4:
            if inputargs["suppresssummary"] == False:  # noqa: E712
                inout_name = "_".join(f"{samplenam}".split("_")[:-1]) + f"_{chainnams[chain]}"
                summstr = "OutputFile," + inout_name + "\nNumberReadsInput," + "0"
                name, fh = _new_summary_file(summaryname, logpath, strftime("%Y_%m_%d"), inputargs["chain"], chain, samplenam)
                print(summstr, file=fh)
                fh.close()
                sort_permissions(name)
            raise ValueError(
                "There are fewer than four lines in this file, and thus it is not a valid FASTQ file. "
                "Please check input and try again.")
        # the reference validates the NEXT four lines (its islice continues on the same handle);
        # a one-record file therefore has nothing to validate (do not crash)
        read = list(itertools.islice(possfq, 0, 4))
    if len(read) < 4:
        read = head
    if read[0][0] != "@":
        raise ValueError(f"Expected @ symbol at beginning of file for valid FASTQ. Found {read[0][0]}.")
    if read[2][0] != "+":
        raise ValueError(f"Expected + symbol at beginning of third line for valid FASTQ. Found {read[2][0]}.")
    if len(read[1]) != len(read[3]):
        raise ValueError(
            f"Length of read to match length of read quality. Found read length = {len(read[1])} "
            f"and read quality length = {len(read[3])}")


def read_tcr_file(species, tagset, gene, filetype, expected_dir_name, chain):
    """Path of `<species>_<tagset>_TR<CHAIN><GENE>.<filetype>`: working directory first,
    then the tag/FASTA directory (reference :187-225).  The reference would now try GitHub;
    this build is offline-only and exits with the reference's message instead."""
    expected_file = f"{species}_{tagset}_TR{chain.upper()}{gene.upper()}.{filetype}"
    if os.path.isfile(expected_file):
        return expected_file
    cand = expected_dir_name + os.sep + expected_file
    if os.path.isfile(cand):
        return cand
    print("Cannot find following file locally or online:", expected_file)
    print("Please either run Decombinator with internet access, or point Decombinator to local copies "
          "of the tag and FASTA files with the '-tf' flag.")
    sys.exit()


def _parse_tags(handle, half_split):
    seqs, jumps = [], []
    for line in handle:
        fields = line.rstrip("\n").split()
        seqs.append(fields[0])
        jumps.append(int(fields[1]))
    half1 = [s[0:half_split] for s in seqs]
    half2 = [s[half_split:] for s in seqs]
    return [seqs, half1, half2, jumps]


def get_v_tags(file_v, half_split):
    """[v_seqs, half1_v_seqs, half2_v_seqs, jump_to_end_v] from a .tags file (reference :820-844)."""
    return _parse_tags(file_v, half_split)


def get_j_tags(file_j, half_split):
    """[j_seqs, half1_j_seqs, half2_j_seqs, jump_to_start_j] (reference :847-866)."""
    return _parse_tags(file_j, half_split)


def _read_fasta(path):
    """(id, sequence) per record in file order; what SeqIO.parse(path, 'fasta') gives the
    reference at :690 (sequence lines joined, header up to the first whitespace as id)."""
    out, header, chunks = [], None, []
    with open(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if header is not None:
                    out.append((header.split()[0] if header.split() else "", "".join(chunks)))
                header, chunks = line[1:], []
            elif header is not None:
                chunks.append(line.strip())
    if header is not None:
        out.append((header.split()[0] if header.split() else "", "".join(chunks)))
    return out


class ChainTables:
    """What import_tcr_info() leaves in module globals in the reference (:593-746), for one
    chain, plus the compiled device tables."""

    def __init__(self, chain, tags, species, v, j, v_genes, j_genes, v_half_split, j_half_split):
        self.chain, self.tags, self.species = chain, tags, species
        self.v_seqs, self.half1_v_seqs, self.half2_v_seqs, self.jump_to_end_v = v
        self.j_seqs, self.half1_j_seqs, self.half2_j_seqs, self.jump_to_start_j = j
        self.v_genes, self.j_genes = v_genes, j_genes
        self.v_regions = [s.upper() for _, s in v_genes]
        self.j_regions = [s.upper() for _, s in j_genes]
        self.v_half_split, self.j_half_split = v_half_split, j_half_split
        if len(self.v_regions) < len(self.v_seqs) or len(self.j_regions) < len(self.j_seqs):
            raise ValueError("fewer FASTA records than tags: tag index i needs FASTA record i")
        self.tables = nat.Tables(self.v_seqs, self.jump_to_end_v, self.v_regions[:len(self.v_seqs)],
                                 self.j_seqs, self.jump_to_start_j, self.j_regions[:len(self.j_seqs)],
                                 v_half_split, j_half_split)
        self.max_read_len = int(self.tables.info()["max_read_len"])


def import_tcr_info(inputargs) -> ChainTables:
    """Chain resolution, tag-set/species validation, tag + FASTA loading (reference :593-746).
    Returns the tables (also kept as the module's current tables) and resets `counts`."""
    global counts, _current
    counts = coll.Counter()
    nochain_error = ("TCR chain not recognised. \n Please either include (one) chain name in the file name "
                     "(i.e. alpha/beta/gamma/delta),\n or use the '-c' flag with an explicit chain option "
                     "(a/b/g/d, case-insensitive).")
    inner = [x for x in chainnams.values() if x in inputargs["infile"].lower()]
    if len(inner) == 1:
        counts["chain_detected"] = 1
    if inputargs["chain"]:
        c = inputargs["chain"].upper()
        chain = {"A": "a", "ALPHA": "a", "TRA": "a", "TCRA": "a", "B": "b", "BETA": "b", "TRB": "b", "TCRB": "b",
                 "G": "g", "GAMMA": "g", "TRG": "g", "TCRG": "g", "D": "d", "DELTA": "d", "TRD": "d",
                 "TCRD": "d"}.get(c)
        if chain is None:
            print(nochain_error)
            sys.exit()
    elif counts["chain_detected"] == 1:
        chain = inner[0][0]
    else:
        print(nochain_error)
        sys.exit()

    print("Importing TCR", chainnams[chain], "gene sequences...")
    # the reference rewrites inputargs["tags"] in place for mouse and for gamma/delta (:640-654)
    if inputargs["tags"] == "extended" and inputargs["species"] == "mouse":
        print("Please note that there is currently no extended tag set for mouse TCR genes.\n"
              " Decombinator will now switch the tag set in use from 'extended' to 'original'.")
        inputargs["tags"] = "original"
    if inputargs["tags"] == "extended" and chain in ("g", "d"):
        print("Please note that there is currently no extended tag set for gamma/delta TCR genes.\n"
              " Decombinator will now switch the tag set in use from 'extended' to 'original'.")
        inputargs["tags"] = "original"
    if inputargs["tags"] == "extended":
        v_half_split, j_half_split = 10, 10
    elif inputargs["tags"] == "original":
        v_half_split, j_half_split = 10, 6
    else:
        print("Tag set unrecognised; should be either 'extended' or 'original' for human, or just 'original' "
              "for mouse. \n Please check tag set and species flag.")
        sys.exit()
    if inputargs["species"] not in ["human", "mouse"]:
        print("Species not recognised. Please select either 'human' (default) or 'mouse'.")
        sys.exit()

    parsed, genes = {}, {}
    for gene, split in (("v", v_half_split), ("j", j_half_split)):
        fasta = read_tcr_file(inputargs["species"], inputargs["tags"], gene, "fasta", inputargs["tagfastadir"], chain)
        genes[gene] = _read_fasta(fasta)
        tagf = read_tcr_file(inputargs["species"], inputargs["tags"], gene, "tags", inputargs["tagfastadir"], chain)
        with open(tagf, "r") as fh:
            parsed[gene] = (get_v_tags if gene == "v" else get_j_tags)(fh, split)
    _current = ChainTables(chain, inputargs["tags"], inputargs["species"], parsed["v"], parsed["j"],
                           genes["v"], genes["j"], v_half_split, j_half_split)
    return _current


def dcr(read, inputargs, tcr: ChainTables | None = None):
    """dcr(read) for ONE read in the frame as given (reference :534-585), through the GPU
    path.  Returns the reference's 7-list or None and adds to `counts`.  Convenience for
    interactive use and tests; the pipeline calls the library on batches."""
    tcr = tcr or _current
    if tcr is None:
        raise RuntimeError("import_tcr_info() has not been called")
    rec, cnt = nat.decombine(tcr.tables, nat.pack_reads([read]), "forward", inputargs["allowNs"],
                             inputargs["lenthreshold"])
    _add_counts(cnt, skip=("read_count", "vj_count", "frame_forward"))
    r = rec[0]
    if int(r["status"]) != 0:
        return None
    s, l = int(r["ins_start"]), int(r["ins_len"])
    return [int(r["v"]), int(r["j"]), int(r["vdel"]), int(r["jdel"]), read[s:s + l], int(r["v_start"]), int(r["j_end"])]


def _add_counts(cnt, skip=()):
    for i, name in enumerate(nat.COUNTER_NAMES):
        if name in skip or name == "frame_forward":
            continue
        v = int(cnt[i])
        if v:
            counts[name] += v


def assemble_rows(records, reads, quals, ids, bcs, bcqs, sampling_tails=None):
    """`.n12` rows from device records (reference :1012-1039): for each decombined read
    [v, j, vdel, jdel, insert, id, inter-tag seq, inter-tag qual, barcode, barcode qual]
    (+ v_tail with sampling_analysis)."""
    rows = []
    for k in np.nonzero(records["status"] == 0)[0]:
        r = records[k]
        vdj, q = reads[k], quals[k]
        if int(r["frame"]) == 0:              # reverse: dcr() saw revcomp(vdj) (:1015-1017)
            frame_read, frame_q = revcomp(vdj), q[::-1]
        else:                                 # forward (:1018-1020)
            frame_read, frame_q = vdj, q
        s, l = int(r["ins_start"]), int(r["ins_len"])
        a, b = int(r["v_start"]), int(r["j_end"])
        row = [str(int(r["v"])), str(int(r["j"])), str(int(r["vdel"])), str(int(r["jdel"])),
               frame_read[s:s + l], ids[k], frame_read[a:b], frame_q[a:b], bcs[k], bcqs[k]]
        if sampling_tails is not None:
            row.append(sampling_tails[k])
        rows.append(row)
    return rows


class _Spans:
    """One batch of read pairs as byte spans into the FASTQ batches' text buffers: what the
    reference's loop body slices out of record1 / record2 (:963-984)."""
    __slots__ = ("v_text", "v_start", "v_len", "q_start", "q_len", "id_text", "id_start", "id_len",
                 "bc_text", "bc_start", "bc_len", "bcq_start", "bcq_len", "tail_start", "tail_len", "last")


def _clip(off, length, lo, hi=None):
    """Span of python's s[lo:hi] (0 <= lo <= hi) inside the span (off, length): uint64 offsets, uint32 lengths (a million of
    each per batch: no temporaries in other types — the first form spent 16 ms of a batch's 80 here)."""
    length = np.asarray(length, dtype=np.uint32)
    b = length if hi is None else np.minimum(length, np.uint32(hi))
    if lo == 0:
        return off, b
    a = np.minimum(length, np.uint32(lo))
    return off + a, b - a


def _next_spans(rd1, rd2, bclength, sampling):
    """The next BATCH_READS iterations of the reference's `for record1, record2 in zip(fq1, fq2)`
    (:961): with bc_read R2 one record from each file, with R1 two consecutive records of the
    one file (zip over the same generator).  None when the zip is exhausted."""
    none_qual = "'NoneType' object is not subscriptable"      # what record[2][...] raises on a FASTA record
    sp = _Spans()
    if rd2 is not None:
        b1 = rd1.next(BATCH_READS, copy=False)      # (windows on the readers' stores: a batch is done with before the next is asked for)
        if b1.n == 0:
            return None
        b2 = rd2.next(b1.n, copy=False)
        sp.last = b2.n < b1.n          # zip() stops with the shorter file
        if sp.last:
            b1.truncate(b2.n)
        if b1.n == 0:
            return None
        if (b2.qual_len == nat.NO_QUAL).any():
            raise TypeError(none_qual)                           # record2[2][:bclength] (:972)
        sp.v_text = sp.id_text = b1.text
        sp.bc_text = b2.text
        sp.v_start, sp.v_len = b1.seq_off, b1.seq_len                                   # vdj = record1[1] (:970)
        sp.q_start, sp.q_len = b1.qual_off, b1.qual_len                                 # NO_QUAL: fails when sliced
        sp.bc_start, sp.bc_len = _clip(b2.seq_off, b2.seq_len, 0, bclength)             # :971
        sp.bcq_start, sp.bcq_len = _clip(b2.qual_off, b2.qual_len, 0, bclength)         # :972
        sp.id_start, sp.id_len = b1.name_off, b1.name_len
        r2_off, r2_len = b2.seq_off, b2.seq_len
    else:
        b = rd1.next(2 * BATCH_READS, copy=False)
        sp.last = b.n < 2 * BATCH_READS
        n = b.n // 2                   # an unpaired last record is consumed and dropped by zip()
        if n == 0:
            return None
        ev, od = slice(0, 2 * n, 2), slice(1, 2 * n, 2)
        if (b.qual_len[ev] == nat.NO_QUAL).any():
            raise TypeError(none_qual)                           # record1[2][bclength:] (:981)
        sp.v_text = sp.id_text = sp.bc_text = b.text
        sp.v_start, sp.v_len = _clip(b.seq_off[ev], b.seq_len[ev], bclength)            # :980
        sp.q_start, sp.q_len = _clip(b.qual_off[ev], b.qual_len[ev], bclength)          # :981
        sp.bc_start, sp.bc_len = _clip(b.seq_off[ev], b.seq_len[ev], 0, bclength)       # :982
        sp.bcq_start, sp.bcq_len = _clip(b.qual_off[ev], b.qual_len[ev], 0, bclength)   # :983
        sp.id_start, sp.id_len = b.name_off[ev], b.name_len[ev]
        r2_off, r2_len = b.seq_off[od], b.seq_len[od]
    sp.tail_start = sp.tail_len = None
    if sampling:                                                 # v_tail = record2[1][bclength:bclength+31]
        sp.tail_start, sp.tail_len = _clip(r2_off, r2_len, bclength, bclength + 31)
    return sp


_FIELD_SEP = ", "      # the `.n12` separator (io.py:507-509); no field of a well-formed FASTQ can hold it


class N12Rows(coll.abc.Sequence):
    """The rows decombinator() returns (reference: a list of 10/11-field lists, :1039), kept as
    the text libdcrx assembled and turned into lists only when somebody asks: iteration, indexing,
    len() and == with a list behave like the reference's list; write_out_intermediate() writes
    the text without ever building the lists."""

    def __init__(self):
        self._chunks = []      # (bytes blob with _FIELD_SEP between fields and "\n" after each row | None, list | None)
        self._tags = []        # per chunk: the value of _tag when it was added (the batch index in decombinator())
        self._tag = 0
        self._len = 0
        self._lists = None

    def _add_blob(self, blob: bytes, n: int) -> None:
        if n:
            self._chunks.append([blob, None, n])
            self._len += n
            self._lists = None
            self._tags.append(self._tag)

    def _tagged_chunks(self):
        """(tag, text, number of rows) per chunk — the tag is the batch the rows came from (sharded runs merge by it)."""
        out = []
        for (blob, rows, n), tag in zip(self._chunks, self._tags):
            if blob is None:
                blob = "".join(_FIELD_SEP.join(str(x) for x in r) + "\n" for r in rows).encode("utf-8")
            out.append((tag, bytes(blob), n))
        return out

    def extend(self, rows) -> None:
        rows = list(rows)
        if rows:
            self._chunks.append([None, rows, len(rows)])
            self._len += len(rows)
            self._lists = None
            self._tags.append(self._tag)

    def append(self, row) -> None:
        self.extend([row])

    @staticmethod
    def _split(blob: bytes):
        rows = [ln.split(_FIELD_SEP) for ln in blob.decode("utf-8", "replace").split("\n")]
        rows.pop()             # the text ends with a newline
        return rows

    def __len__(self):
        return self._len

    def __iter__(self):
        for blob, rows, _ in self._chunks:
            yield from (rows if rows is not None else self._split(blob))

    def _all(self):
        if self._lists is None:
            self._lists = list(iter(self))
        return self._lists

    def __getitem__(self, k):
        return self._all()[k]

    def __eq__(self, other):
        if isinstance(other, (list, N12Rows)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __repr__(self):
        return f"N12Rows({self._len} rows)"

    def write_text(self, fh, joiner: str = ", ") -> None:
        """One row per line, fields joined by `joiner` (reference io.py:507-509)."""
        sep, j = _FIELD_SEP.encode(), joiner.encode()
        for blob, rows, _ in self._chunks:
            if rows is not None:
                fh.write("".join(joiner.join(map(str, r)) + "\n" for r in rows))
            else:
                fh.flush()                     # the FASTQ's own bytes, as the reference's text mode writes them
                fh.buffer.write(blob if sep == j else blob.replace(sep, j))


def assemble_rows_spans(records, sp, into=None):
    """assemble_rows for a batch held as spans, in bulk: libdcrx writes the rows as text
    (dcrx_assemble_rows), which is split back into the reference's list-of-lists.  A batch whose
    text holds the separator byte (never in a real FASTQ) takes the per-row path below."""
    hit = records["status"] == 0
    if not hit.any():
        return None if into is not None else []
    if (sp.q_len[hit] == nat.NO_QUAL).any():
        raise TypeError("'NoneType' object is not subscriptable")   # tcrQ = vdjqual[...] on a FASTA record
    tail = None if sp.tail_start is None else (sp.bc_text, sp.tail_start, sp.tail_len)
    try:
        blob, n = nat.assemble_rows_blob(records, (sp.v_text, sp.v_start, sp.v_len), (sp.v_text, sp.q_start, sp.q_len),
                                         (sp.id_text, sp.id_start, sp.id_len), (sp.bc_text, sp.bc_start, sp.bc_len),
                                         (sp.bc_text, sp.bcq_start, sp.bcq_len), tail, _FIELD_SEP)
    except nat.SeparatorClash:
        rows = _assemble_rows_spans_py(records, sp)
        if into is not None:
            into.extend(rows)
            return None
        return rows
    if into is not None:
        into._add_blob(blob, n)
        return None
    rows = N12Rows._split(blob)
    assert len(rows) == n
    return rows


def _assemble_rows_spans_py(records, sp):
    """Per-row form of assemble_rows_spans (also its cross-check in the tests)."""
    rows = []

    def cut(text, off, length, k):
        o = int(off[k])
        return text[o:o + int(length[k])].decode("latin-1")
    for k in np.nonzero(records["status"] == 0)[0]:
        r = records[k]
        if int(sp.q_len[k]) == nat.NO_QUAL:
            raise TypeError("'NoneType' object is not subscriptable")   # tcrQ = vdjqual[...] on a FASTA record
        vdj, q = cut(sp.v_text, sp.v_start, sp.v_len, k), cut(sp.v_text, sp.q_start, sp.q_len, k)
        if int(r["frame"]) == 0:
            frame_read, frame_q = revcomp(vdj), q[::-1]
        else:
            frame_read, frame_q = vdj, q
        s, l = int(r["ins_start"]), int(r["ins_len"])
        a, b = int(r["v_start"]), int(r["j_end"])
        o = int(sp.id_start[k])
        row = [str(int(r["v"])), str(int(r["j"])), str(int(r["vdel"])), str(int(r["jdel"])),
               frame_read[s:s + l], sp.id_text[o:o + int(sp.id_len[k])].decode("utf-8", "replace"), frame_read[a:b],
               frame_q[a:b], cut(sp.bc_text, sp.bc_start, sp.bc_len, k), cut(sp.bc_text, sp.bcq_start, sp.bcq_len, k)]
        if sp.tail_start is not None:
            row.append(cut(sp.bc_text, sp.tail_start, sp.tail_len, k))
        rows.append(row)
    return rows


def _summary_text(inputargs, chain, samplenam, date, timetaken):
    """The Decombinator summary CSV body (reference :1097-1195), line for line."""
    inout_name = "_".join(f"{samplenam}".split("_")[:-1]) + f"_{chainnams[chain]}"
    lines = ["Property,Value", "Directory," + os.getcwd(), "InputFile," + inout_name, "OutputFile," + inout_name,
             "DateFinished," + date, "TimeFinished," + strftime("%H:%M:%S"),
             "TimeTaken(Seconds)," + str(round(timetaken, 2)), "", "InputArguments:,"]
    for s in ["species", "chain", "extension", "tags", "dontgzip", "allowNs", "orientation", "lenthreshold",
              "bc_read", "bclength"]:
        lines.append(s + "," + str(inputargs[s]))
    counts["pc_decombined"] = counts["vj_count"] / counts["read_count"]
    lines += ["", "NumberReadsInput," + str(counts["read_count"]),
              "NumberReadsDecombined," + str(counts["vj_count"]),
              "PercentReadsDecombined," + str(round(counts["pc_decombined"], 3)),
              "", "ReadsAssignedUsingHalfTags:,",
              "V1error," + str(counts["verr1"]), "V2error," + str(counts["verr2"]),
              "J1error," + str(counts["jerr1"]), "J2error," + str(counts["jerr2"]),
              "", "ReadsFilteredOut:,",
              "AmbiguousBaseCall(DCR)," + str(counts["dcrfilter_intertagN"]),
              "AmbiguousBaseCall(Barcode)," + str(counts["dcrfilter_barcodeN"]),
              "OverlongInterTagSeq," + str(counts["dcrfilter_toolong_intertag"]),
              "ImpossibleDeletions," + str(counts["dcrfilter_imposs_deletion"]),
              "OverlappingTagBoundaries," + str(counts["dcrfilter_tag_overlap"]),
              "", "ReadsFailedAssignment:,",
              "MultipleVtagMatches," + str(counts["multiple_v_matches"]),
              "VTagAtEndRead," + str(counts["v_del_failed_tag_at_end"]),
              "VDeletionsUndetermined," + str(counts["v_del_failed"]),
              "FoundV1HalfTagNotV2," + str(counts["foundv1notv2"]),
              "FoundV2HalfTagNotV1," + str(counts["foundv2notv1"]),
              "NoVDetected," + str(counts["no_vtags_found"]),
              "MultipleJTagMatches," + str(counts["multiple_j_matches"]),
              "JDeletionsUndermined," + str(counts["j_del_failed"]),
              "FoundJ1HalfTagNotJ2," + str(counts["foundj1notj2"]),
              "FoundJ2HalfTagNotJ1," + str(counts["foundj2notj1"]),
              "NoJDetected," + str(counts["no_j_assigned"])]
    return "\n".join(lines)


def _decombinator_setup(inputargs: dict, rank: int, world: int, state: dict) -> None:
    """This rank's part of the stage before anything is read: tables, (rank 0) log directory and FASTQ check.  No collective
    in here: a sharded run exchanges what this raised before it goes on (decombinator())."""
    print("Running Decombinator (MI355X / HIP build) version", __version__)
    opener = opener_check(inputargs)
    tcr = import_tcr_info(inputargs)
    chain = tcr.chain
    samplenam = str(inputargs["infile"].split(".")[0])
    if os.sep in samplenam:
        samplenam = samplenam.split(os.sep)[-1]

    summaryname = logpath = None
    date = strftime("%Y_%m_%d")
    if inputargs["suppresssummary"] == False:  # noqa: E712
        logpath = inputargs["outpath"] + f"Logs{os.sep}"
        if rank == 0:
            os.makedirs(logpath, exist_ok=True)
        summaryname = logpath + date + "_"
        if inputargs["chain"]:
            summaryname += chainnams[chain] + "_"
        summaryname += samplenam + "_Decombinator_Summary.csv"
    if inputargs["dontcheck"] == False and rank == 0:  # noqa: E712
        # (the reference crashes here with suppresssummary=True; this build checks anyway)
        fastq_check(inputargs, opener, samplenam, summaryname, logpath, chain)

    if inputargs["orientation"] not in nat.ORIENTATIONS:
        raise ValueError("orientation must be forward, reverse or both")
    if inputargs["nobarcoding"] == False and inputargs["bc_read"] not in ("R1", "R2"):  # noqa: E712
        raise ValueError("bc_read must be R1 or R2")
    state.update(tcr=tcr, chain=chain, samplenam=samplenam, summaryname=summaryname, logpath=logpath, date=date)


def _decombinator_loop(inputargs: dict, rank: int, world: int, state: dict, plan_shards=None) -> None:
    """The read loop (reference :948-1050) over this rank's records: every rank of a sharded run comes here (the shard plan is
    a collective), and leaves its rows in `state`."""
    tcr = state["tcr"]
    bclength = inputargs["bclength"]
    counts["start_time"] = time()
    stage_seconds.clear()
    print("Decombining FASTQ data...")
    outdata = N12Rows()
    orientation = inputargs["orientation"]

    if inputargs["nobarcoding"] == False:  # noqa: E712
        paired = inputargs["bc_read"] == "R2"
        gz = inputargs["infile"].endswith(".gz")       # opener_check: one opener for both files (:118-123)
        path1 = inputargs["infile"]
        path2 = inputargs["infile"].replace("1.f", "2.f") if paired else None
        # a sharded run reads its own records only (sharded.plan_fastq_shards: byte ranges of whole records, the files of a
        # pair cut at the same record); files that cannot be cut are read whole by every rank and the batches dealt round-robin
        ranges = None
        if world > 1 and plan_shards is not None:
            ranges = plan_shards([path1] + ([path2] if paired else []), world, rank, 1 if paired else 2)
        stage_info.clear()
        stage_info.update(sharded_input=ranges is not None, byte_ranges=ranges, rank=rank, world=world)
        if ranges is not None:
            rd1 = nat.FastqReader(path1, False, byte_range=ranges[0])
            rd2 = nat.FastqReader(path2, False, byte_range=ranges[1]) if paired else None
        else:
            rd1 = nat.FastqReader(path1, gz)
            rd2 = nat.FastqReader(path2, gz) if paired else None
        sampling = bool(inputargs.get("sampling_analysis"))
        try:
            batch_index = -1
            while True:
                t0 = time()
                spans = _next_spans(rd1, rd2, bclength, sampling)
                if spans is None:
                    break
                batch_index += 1
                if ranges is None and batch_index % world != rank:     # another rank's batch: read, not processed
                    if spans.last:
                        break
                    continue
                outdata._tag = rank if ranges is not None else batch_index
                n = len(spans.v_start)
                if inputargs["allowNs"] == False:  # noqa: E712    counted, never dropped (:985-989)
                    counts["dcrfilter_barcodeN"] += nat.count_prefix_byte(spans.bc_text, spans.bc_start, spans.bc_len,
                                                                          1 << 30, "N")
                before = counts["read_count"]
                counts["read_count"] += n
                if inputargs["dontcount"] == False and counts["read_count"] // 100000 > before // 100000:  # noqa: E712
                    print("\t read", (counts["read_count"] // 100000) * 100000)
                t1 = time()
                # reads of up to 511 nt run on the register shapes; longer ones (merged pairs, long amplicons: the reference has no
                # length limit, decombine.py:228-265, :534-585) leave the batch for a call of their own — one read per lane from
                # memory, dcrx's long form — and come back into their places; beyond 65 535 nt (the 16-bit lengths and offsets of
                # the record) nothing decombines them: said before anything of the batch is processed
                longest = int(spans.v_len.max()) if n else 0
                if longest > tcr.max_read_len:
                    raise ValueError(f"a read of {longest} nt exceeds the {tcr.max_read_len} nt this build decombines "
                                     f"(dcrx_tables_info.max_read_len); trim or split the reads")
                t2 = t1
                if longest > nat.FAST_MAX_READ_LEN:
                    is_long = spans.v_len > nat.FAST_MAX_READ_LEN
                    rec = np.empty(n, dtype=nat.RECORD_DTYPE)
                    cnt = np.zeros(nat.N_COUNTERS, dtype=np.uint64)
                    for idx in (np.nonzero(~is_long)[0], np.nonzero(is_long)[0]):
                        if len(idx) == 0:
                            continue
                        tp = time()
                        batch = nat.pack_reads_span(spans.v_text, spans.v_start[idx], spans.v_len[idx])
                        t2 += time() - tp
                        r_part, c_part = nat.decombine(tcr.tables, batch, orientation, inputargs["allowNs"], inputargs["lenthreshold"])
                        rec[idx] = r_part
                        cnt += c_part.astype(np.uint64)
                else:
                    batch = nat.pack_reads_span(spans.v_text, spans.v_start, spans.v_len)
                    t2 = time()
                    rec, cnt = nat.decombine(tcr.tables, batch, orientation, inputargs["allowNs"], inputargs["lenthreshold"])
                t3 = time()
                _add_counts(cnt, skip=("read_count",))
                assemble_rows_spans(rec, spans, into=outdata)
                t4 = time()
                for key, dt in (("read", t1 - t0), ("pack", t2 - t1), ("device", t3 - t2), ("rows", t4 - t3)):
                    stage_seconds[key] = stage_seconds.get(key, 0.0) + dt
                if spans.last:
                    break
        finally:
            tc = time()
            rd1.close()
            if rd2 is not None:
                rd2.close()
            stage_seconds["close"] = time() - tc
    else:
        # reference behaviour: with nobarcoding the read loop never runs
        if inputargs["extension"] == "n12":
            print("Non-barcoding option selected, but default output file extension (n12) detected. "
                  "Automatically changing to 'nbc'.")

    state.update(outdata=outdata)


def decombinator(inputargs: dict, shard=None, reduce_counts=None, exchange_error=None, plan_shards=None) -> list:
    """The decombine stage (reference decombinator(), :881-1202): returns the 10-field rows
    that write_out_intermediate() turns into the `.n12` file, as an N12Rows sequence (a lazy
    list of lists).

    shard = (rank, world): this process reads and decombines its contiguous share of the records (plan_shards, given by
    decombinator_sharded: byte ranges of whole records; rows tagged with the rank) — or, for files that cannot be cut, the
    batches whose index is `rank` modulo `world` of the whole file (rows tagged with their batch);
    reduce_counts(counts), when given, is called once the loop is over and must leave the sums over all ranks in
    `counts`; only rank 0 creates the log directory, checks the FASTQ, prints the totals and writes the summary log.
    exchange_error(exc_or_None), when given, is called by every rank before the first collective with whatever this
    rank's part raised: it must raise on every rank when any rank failed (a rank that died alone would leave the
    others waiting in reduce_counts for ever).  decombinator_amd.sharded.decombinator_sharded drives this."""
    rank, world = shard if shard is not None else (0, 1)
    state = {}
    for part in (_decombinator_setup, _decombinator_loop):
        err = None
        try:
            if part is _decombinator_setup:
                part(inputargs, rank, world, state)
            else:
                part(inputargs, rank, world, state, plan_shards)
        except BaseException as e:      # (SystemExit too: the reference leaves through sys.exit() for bad chains and tag sets)
            if exchange_error is None:
                raise
            err = e
        if exchange_error is not None:
            exchange_error(err)
    outdata, chain, samplenam, summaryname, logpath, date = (state[k] for k in ("outdata", "chain", "samplenam", "summaryname", "logpath", "date"))
    if reduce_counts is not None:
        reduce_counts(counts)
    counts["end_time"] = time()
    timetaken = counts["end_time"] - counts["start_time"]
    if rank != 0:
        return outdata
    print("Analysed", "{:,}".format(counts["read_count"]), "reads, finding", "{:,}".format(counts["vj_count"]),
          chainnams[chain], "VJ rearrangements")
    print("Reading from", inputargs["infile"] + ", writing to variable")
    print("Took", str(round(timetaken, 2)), "seconds")

    if inputargs["suppresssummary"] == False:  # noqa: E712
        name, fh = _new_summary_file(summaryname, logpath, date, inputargs["chain"], chain, samplenam)
        print(_summary_text(inputargs, chain, samplenam, date, timetaken), file=fh)
        fh.close()
        sort_permissions(name)
    return outdata
