"""The native batch FASTQ reader (dcrx_fastq_*) and the host's readfq() against records the
reference's readfq yields (tests/golden/readfq_cases.json, made by oracle/gen_readfq_golden.py),
plus the span packer against the contiguous packer.  CPU only: no kernel runs here."""
import gzip
import json
import os

import numpy as np
import pytest

from decombinator_amd import _native as nat
from decombinator_amd import decombine as host

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "readfq_cases.json")))["cases"]


def _expected(case):
    return [tuple(r) for r in case["records"]]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_host_readfq_matches_reference(case, tmp_path):
    p = tmp_path / "x.fq"
    p.write_bytes(case["text"].encode())
    with host.opener_check({"infile": str(p)})(str(p), "rt") as fh:
        assert list(host.readfq(fh)) == _expected(case)


@pytest.mark.parametrize("gz", [False, True], ids=["plain", "gz"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_native_reader_matches_reference(case, gz, tmp_path):
    p = tmp_path / ("x.fq.gz" if gz else "x.fq")
    if gz:
        with gzip.open(p, "wb") as f:
            f.write(case["text"].encode())
    else:
        p.write_bytes(case["text"].encode())
    for batch in (1, 3, 1 << 20):
        got = []
        with nat.FastqReader(str(p)) as rd:
            while True:
                b = rd.next(batch)
                if b.n == 0:
                    break
                assert b.n <= batch
                got += b.records()
            assert rd.next(batch).n == 0          # stays exhausted
        assert got == _expected(case), (case["name"], batch)


def test_native_reader_large_file_crosses_buffer_refills(tmp_path):
    rng = np.random.default_rng(5)
    n = 60000
    seqs = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(100, 200)))) for _ in range(200)]
    p = tmp_path / "big.fq"
    with open(p, "w", newline="") as f:
        for i in range(n):
            s = seqs[i % len(seqs)]
            eol = "\r\n" if i % 7 == 0 else "\n"
            f.write(f"@read{i} 1:N:0{eol}{s}{eol}+{eol}{'I' * len(s)}{eol}")
    assert os.path.getsize(p) > (8 << 20)          # several 4 MiB refills
    with host.opener_check({"infile": str(p)})(str(p), "rt") as fh:
        exp = list(host.readfq(fh))
    got = []
    with nat.FastqReader(str(p)) as rd:
        while True:
            b = rd.next(25000)
            if b.n == 0:
                break
            got += b.records()
    assert got == exp and len(got) == n


def test_gz_flag_rejects_plain_file(tmp_path):
    p = tmp_path / "plain.fq.gz"
    p.write_bytes(b"@r\nAC\n+\nII\n")
    with pytest.raises(RuntimeError):
        nat.FastqReader(str(p))
    with pytest.raises(RuntimeError):
        nat.FastqReader(str(tmp_path / "missing.fq"))


def test_span_packer_equals_contiguous_packer():
    rng = np.random.default_rng(11)
    reads = ["".join(rng.choice(list("ACGTNacgtRY-"), p=[.22, .22, .22, .22] + [.015] * 8, size=int(rng.integers(0, 161))))
             for _ in range(70000)]                                  # >= 2^16 reads: the threaded branch
    ref = nat.pack_reads(reads)
    # the same reads scattered in a text with separators, in the order given by start[]
    text = ("#".join(reads) + "#").encode("latin-1")
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    start = np.zeros(len(reads), dtype=np.uint64)
    start[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
    got = nat.pack_reads_span(text, start, lens, stride=ref.stride)
    assert np.array_equal(got.packed, ref.packed)
    assert np.array_equal(got.exc_read, ref.exc_read) and np.array_equal(got.exc_pos, ref.exc_pos)
    assert np.array_equal(got.exc_chr, ref.exc_chr)
    assert np.array_equal(got.lens, ref.lens)
    assert nat.unpack_reads(got) == reads


def test_count_prefix_byte():
    text = b"NACGTxxACGNTyyACGT"
    start = np.array([0, 7, 14], dtype=np.uint64)
    length = np.array([5, 5, 4], dtype=np.uint32)
    assert nat.count_prefix_byte(text, start, length, 3, "N") == 1
    assert nat.count_prefix_byte(text, start, length, 4, "N") == 2
    assert nat.count_prefix_byte(text, start, length, 100, "N") == 2
    assert nat.count_prefix_byte(text, start, length, 0, "N") == 0


def test_bulk_row_assembly_equals_per_row_assembly():
    """dcrx_assemble_rows against the per-row Python form on random records: both frames, slices
    that clamp, ambiguity codes, quality strings longer / shorter than the read."""
    rng = np.random.default_rng(3)
    n = 3000
    alphabet = list("ACGTNRYacgtn-")
    seqs, quals, ids, bcs, bcqs, tails = [], [], [], [], [], []
    for k in range(n):
        L = int(rng.integers(0, 120))
        seqs.append("".join(rng.choice(alphabet, size=L)))
        quals.append("".join(chr(int(c)) for c in rng.integers(33, 127, size=max(0, L + int(rng.integers(-3, 4))))))
        ids.append(f"id:{k}\tx/é" if k % 17 == 0 else f"id:{k}")
        bcs.append("".join(rng.choice(list("ACGTN"), size=int(rng.integers(0, 12)))))
        bcqs.append("I" * int(rng.integers(0, 12)))
        tails.append("".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 31)))))
    rec = np.zeros(n, dtype=nat.RECORD_DTYPE)
    rec["status"] = rng.choice([0, 0, 6, 9], size=n)
    rec["frame"] = rng.integers(0, 2, size=n)
    for f, hi in (("v", 200), ("j", 60), ("vdel", 40), ("jdel", 40), ("v_start", 130), ("j_end", 140), ("ins_start", 130),
                  ("ins_len", 30)):
        rec[f] = rng.integers(0, hi, size=n)

    def spans(strs):
        bs = [s.encode("utf-8") for s in strs]
        text = b"|".join(bs) + b"|"
        lens = np.array([len(b) for b in bs], dtype=np.uint32)
        start = np.zeros(len(bs), dtype=np.uint64)
        start[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)
        return text, start, lens

    class SP:
        pass
    for with_tail in (False, True):
        sp = SP()
        # seq and qual share one text, like a FASTQ batch
        text, start, lens = spans([x for pair in zip(seqs, quals) for x in pair])
        sp.v_text = sp.id_text = text
        sp.v_start, sp.v_len = start[0::2], lens[0::2]
        sp.q_start, sp.q_len = start[1::2], lens[1::2]
        t2, s2, l2 = spans([x for four in zip(ids, bcs, bcqs, tails) for x in four])
        sp.id_text = sp.bc_text = t2
        sp.id_start, sp.id_len = s2[0::4], l2[0::4]
        sp.bc_start, sp.bc_len = s2[1::4], l2[1::4]
        sp.bcq_start, sp.bcq_len = s2[2::4], l2[2::4]
        sp.tail_start, sp.tail_len = (s2[3::4], l2[3::4]) if with_tail else (None, None)
        bulk = host.assemble_rows_spans(rec, sp)
        slow = host._assemble_rows_spans_py(rec, sp)
        assert bulk == slow and len(bulk) == int((rec["status"] == 0).sum())
        # and the list form used by the reference-shaped API
        lst = host.assemble_rows(rec, seqs, quals, ids, bcs, bcqs, tails if with_tail else None)
        assert lst == bulk


def test_separator_inside_a_field_falls_back_to_the_per_row_path():
    rec = np.zeros(2, dtype=nat.RECORD_DTYPE)
    rec["frame"] = 1
    rec["j_end"] = 4
    text = b"ACGT|IIII|id, x|AC|II|ACGT|IIII|id2|AC|II|"
    lens = np.array([4, 4, 5, 2, 2, 4, 4, 3, 2, 2], dtype=np.uint32)
    start = np.zeros(10, dtype=np.uint64)
    start[1:] = np.cumsum(lens[:-1].astype(np.uint64) + 1)

    class SP:
        pass
    sp = SP()
    sp.v_text = sp.id_text = sp.bc_text = text
    sp.v_start, sp.v_len = start[0::5], lens[0::5]
    sp.q_start, sp.q_len = start[1::5], lens[1::5]
    sp.id_start, sp.id_len = start[2::5], lens[2::5]
    sp.bc_start, sp.bc_len = start[3::5], lens[3::5]
    sp.bcq_start, sp.bcq_len = start[4::5], lens[4::5]
    sp.tail_start = sp.tail_len = None
    with pytest.raises(nat.SeparatorClash):
        nat.assemble_rows_blob(rec, (text, sp.v_start, sp.v_len), (text, sp.q_start, sp.q_len), (text, sp.id_start, sp.id_len),
                               (text, sp.bc_start, sp.bc_len), (text, sp.bcq_start, sp.bcq_len))
    rows = host.assemble_rows_spans(rec, sp)
    assert rows == [["0", "0", "0", "0", "", "id, x", "ACGT", "IIII", "AC", "II"],
                    ["0", "0", "0", "0", "", "id2", "ACGT", "IIII", "AC", "II"]]
    out = host.N12Rows()
    host.assemble_rows_spans(rec, sp, into=out)
    assert out == rows and len(out) == 2 and out[1][5] == "id2"


def test_parallel_fast_path_equals_the_generator(tmp_path, monkeypatch):
    """Four-line FASTQ goes through the reader's parallel fast path (a block cut at record starts, pieces parsed by
    several threads, the block itself as the text buffer); DCRX_FASTQ_SERIAL=1 keeps the line-by-line generator.  Same
    records either way, for batch sizes that do and do not divide the file, with names that hold spaces, '@' and '+' at
    the start of quality lines, long and short records — and files the strict grammar does not take (a wrapped record
    in the middle, an unterminated last line, CRLF) fall back as a whole batch."""
    import numpy as np
    rng = np.random.default_rng(5)
    def fastq(n, wrap_at=None, crlf=False, cut_tail=False):
        out = []
        for i in range(n):
            L = int(rng.integers(1, 400))
            seq = "".join(rng.choice(list("ACGTN"), size=L))
            qual = "".join(rng.choice(list("@+>#IJK5"), size=L + (int(rng.integers(0, 3)) if i % 97 == 0 else 0)))
            if wrap_at is not None and i == wrap_at:
                out.append(f"@r{i} x y\n{seq[:L // 2]}\n{seq[L // 2:]}\n+\n{qual[:L // 2]}\n{qual[L // 2:]}\n")
            else:
                out.append(f"@r{i} lane:{i % 7} +@\n{seq}\n+{'r%d' % i if i % 5 == 0 else ''}\n{qual}\n")
        text = "".join(out)
        if cut_tail:
            text = text[:-1]
        return text.replace("\n", "\r\n") if crlf else text

    def records(path, batch):
        got = []
        with nat.FastqReader(str(path)) as rd:
            while True:
                b = rd.next(batch)
                if b.n == 0:
                    break
                t = bytes(b.text)
                for k in range(b.n):
                    q = None if b.qual_len[k] == nat.NO_QUAL else t[b.qual_off[k]:b.qual_off[k] + b.qual_len[k]]
                    got.append((t[b.name_off[k]:b.name_off[k] + b.name_len[k]], t[b.seq_off[k]:b.seq_off[k] + b.seq_len[k]], q))
                if b.n < batch:
                    break
        return got

    cases = {"plain": fastq(30000), "wrapped": fastq(20000, wrap_at=11111), "crlf": fastq(8000, crlf=True),
             "unterminated": fastq(9000, cut_tail=True)}
    for name, text in cases.items():
        p = tmp_path / f"{name}.fq"
        p.write_bytes(text.encode())
        monkeypatch.setenv("DCRX_FASTQ_SERIAL", "1")
        want = records(p, 4096)
        monkeypatch.delenv("DCRX_FASTQ_SERIAL")
        assert len(want) >= 8000
        for batch in (4096, 7001, 1 << 20):
            assert records(p, batch) == want, (name, batch, "mapped")           # parsed from the mapped file while it is strict
            monkeypatch.setenv("DCRX_FASTQ_NO_MMAP", "1")
            assert records(p, batch) == want, (name, batch, "buffered")         # blocks read into the reader's own buffer
            monkeypatch.delenv("DCRX_FASTQ_NO_MMAP")


def test_fast_paths_equal_the_generator_on_random_line_soup(tmp_path, monkeypatch):
    """Differential test: files assembled at random from FASTQ-like fragments (records of one to three sequence lines,
    FASTA records, blank lines, stray '+' and '@' lines, CR/LF and lone CR, a missing final newline, over- and
    under-long qualities) — the mapped and the buffered fast path must give exactly the generator's records, whatever
    they make of the file (mostly: hand it over at the first line that is not strict four-line FASTQ)."""
    import numpy as np
    rng = np.random.default_rng(77)
    alphabet = list("ACGTN")

    def seq(n):
        return "".join(rng.choice(alphabet, size=n))

    def fragment():
        k = int(rng.integers(0, 12))
        L = int(rng.integers(0, 60))
        s = seq(L)
        q = "".join(rng.choice(list("@+>!IJ#"), size=L))
        if k <= 5:
            return f"@r{int(rng.integers(0, 1000))} d\n{s}\n+\n{q}\n"                      # strict record (possibly empty sequence)
        if k == 6:
            h = L // 2
            return f"@w\n{s[:h]}\n{s[h:]}\n+\n{q[:h]}\n{q[h:]}\n"                          # wrapped
        if k == 7:
            return f">fa x\n{s}\n"                                                          # FASTA
        if k == 8:
            return "\n"
        if k == 9:
            return f"@long\n{s}\n+\n{q}{q[:3]}\n"                                           # quality longer than the sequence
        if k == 10:
            return f"@short\n{s}\n+\n{q[:max(0, L - 2)]}\n"                                 # quality shorter: runs into the next lines
        return "+stray\n" if rng.random() < 0.5 else "@stray\n"

    def records(path, batch):
        got = []
        with nat.FastqReader(str(path)) as rd:
            while True:
                b = rd.next(batch)
                if b.n == 0:
                    break
                t = bytes(b.text)
                for k in range(b.n):
                    q = None if b.qual_len[k] == nat.NO_QUAL else t[b.qual_off[k]:b.qual_off[k] + b.qual_len[k]]
                    got.append((t[b.name_off[k]:b.name_off[k] + b.name_len[k]], t[b.seq_off[k]:b.seq_off[k] + b.seq_len[k]], q))
                if b.n < batch:
                    break
        return got

    for trial in range(120):
        strict_prefix = "".join(f"@p{i}\n{seq(20)}\n+\n{'I' * 20}\n" for i in range(int(rng.integers(0, 40))))
        text = strict_prefix + "".join(fragment() for _ in range(int(rng.integers(0, 40))))
        mode = int(rng.integers(0, 4))
        if mode == 1:
            text = text.replace("\n", "\r\n")
        elif mode == 2:
            text = text.replace("\n", "\r")
        if rng.random() < 0.3 and text.endswith(("\n", "\r")):
            text = text[:-1]
        p = tmp_path / f"soup{trial}.fq"
        p.write_bytes(text.encode())
        batch = int(rng.choice([1, 3, 16, 1000]))
        monkeypatch.setenv("DCRX_FASTQ_SERIAL", "1")
        want = records(p, batch)
        monkeypatch.delenv("DCRX_FASTQ_SERIAL")
        assert records(p, batch) == want, (trial, "mapped", batch)
        monkeypatch.setenv("DCRX_FASTQ_NO_MMAP", "1")
        assert records(p, batch) == want, (trial, "buffered", batch)
        monkeypatch.delenv("DCRX_FASTQ_NO_MMAP")
