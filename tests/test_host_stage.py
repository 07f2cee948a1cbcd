"""Host side of the decombine stage (decombinator_amd/decombine.py, io.py) against the
whole-stage known answer captured from the reference's own decombinator()
(tests/golden/stage_human_extended_b.json): rows of the `.n12` and the summary body.

Without a GPU the device call is replaced IN THE TEST by the oracle (a stand-in for the
device, so that FASTQ reading, barcode slicing, R1-mode pairing, row assembly, counters and
the log are checked on CPU); `-m gpu` runs the same comparison through the real HIP path."""
import io as _io
import json
import os
import time

import numpy as np
import pytest

from decombinator_amd import _native as nat
from decombinator_amd import decombine as dec
from decombinator_amd import io as dio
from decombinator_amd import synth
from tests import parity_util as pu

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "stage_human_extended_b.json")


@pytest.fixture(scope="module")
def stage():
    return json.load(open(FIXTURE))


@pytest.fixture()
def workdir(tmp_path, stage):
    ts = stage["tagset"]
    t = synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"],
                     v_jumps=ts["v_jumps"], v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"],
                     j_jumps=ts["j_jumps"], j_names=ts["j_names"], j_regions=ts["j_regions"])
    t.write(str(tmp_path / "tags"))
    (tmp_path / "SYNTH_1.fq").write_text(stage["fastq_r1"])
    (tmp_path / "SYNTH_2.fq").write_text(stage["fastq_r2"])
    return tmp_path


def _oracle_device(stage):
    """nat.decombine stand-in backed by the oracle (CPU tests only)."""
    from tests import golden_util as gu
    ot = gu.oracle_tables(stage["tagset"])

    def fake(tables, batch, orientation="reverse", allow_ns=False, lenthreshold=130, flags=0):
        reads = nat.unpack_reads(batch)
        return pu.oracle_records(ot, reads, orientation, allow_ns, lenthreshold)
    return fake


def _run_and_compare(stage, workdir, run):
    outdir = workdir / f"out_{run['bc_read']}_{run['orientation']}"
    outdir.mkdir()
    args = dio.create_args_dict(infile=str(workdir / "SYNTH_1.fq"), chain="b", bc_read=run["bc_read"], dontgzip=True,
                                dontcount=True, orientation=run["orientation"], allowNs=run["allowNs"],
                                tagfastadir=str(workdir / "tags"), outpath=str(outdir) + os.sep, command="decombine")
    rows = dec.decombinator(args)
    assert rows == run["rows"]
    logs = list((outdir / "Logs").glob("*.csv"))
    assert len(logs) == 1 and logs[0].name.split("_", 3)[3] == run["log_name_tail"]
    body = logs[0].read_text().split("\n")
    keep = [ln for ln in body if not ln.startswith(("Directory,", "DateFinished,", "TimeFinished,", "TimeTaken"))]
    assert keep == run["summary_lines"]
    out = dio.write_out_intermediate(rows, args, ".n12")
    text = open(out).read()
    assert text == "".join(", ".join(r) + "\n" for r in run["rows"])
    assert os.path.basename(out) == "dcr_SYNTH_1_beta.n12" and oct(os.stat(out).st_mode)[-3:] == "666"


@pytest.mark.parametrize("k", [0, 1, 2], ids=["R2-reverse", "R2-both-allowNs", "R1-reverse"])
def test_stage_host_logic_with_oracle_as_device(stage, workdir, monkeypatch, k):
    monkeypatch.setattr(nat, "decombine", _oracle_device(stage))
    _run_and_compare(stage, workdir, stage["runs"][k])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2], ids=["R2-reverse", "R2-both-allowNs", "R1-reverse"])
def test_stage_through_hip_path(stage, workdir, k):
    _run_and_compare(stage, workdir, stage["runs"][k])


def test_readfq_behaviour(stage):
    recs = list(dec.readfq(_io.StringIO(stage["fastq_r1"])))
    assert len(recs) == 700 and recs[0][0].startswith("SYN:0:") and all(len(r[1]) == len(r[2]) for r in recs)
    # multi-line FASTQ, FASTA records, header without space, quality cut short by EOF
    txt = "@a x y\nAC\nGT\n+\nII\nII\n>b\nAAAA\nCC\n@c\nACGT\n+anything\nIIII\n@d\nACGT\n+\nII\n"
    got = list(dec.readfq(_io.StringIO(txt)))
    assert got == [("a", "ACGT", "IIII"), ("b", "AAAACC", None), ("c", "ACGT", "IIII"), ("d", "ACGT", None)]
    assert list(dec.readfq(_io.StringIO(""))) == []


def test_revcomp_matches_biopython_table():
    assert dec.revcomp("ACGTNacgtnRYKMBVDHSWXU-") == "-AXWSDHBVKMRYnacgtNACGT"


def test_import_tcr_info_rules(workdir, capsys):
    base = dict(infile="sample_beta_1.fq", chain=None, tags="extended", species="human",
                tagfastadir=str(workdir / "tags"))
    t = dec.import_tcr_info(dict(base))
    assert t.chain == "b" and dec.counts["chain_detected"] == 1 and (t.v_half_split, t.j_half_split) == (10, 10)
    assert t.half1_v_seqs[0] == t.v_seqs[0][:10] and t.half2_j_seqs[0] == t.j_seqs[0][10:]
    assert t.tables.info()["n_v"] == 20
    for bad in (dict(base, infile="x_1.fq"), dict(base, chain="q"), dict(base, tags="weird", chain="b"),
                dict(base, species="dog", chain="b"), dict(base, chain="a")):   # alpha files are not there
        with pytest.raises(SystemExit):
            dec.import_tcr_info(bad)
    args = dict(base, chain="g")
    with pytest.raises(SystemExit):        # gamma switches to `original`, whose files are missing here
        dec.import_tcr_info(args)
    assert args["tags"] == "original"      # rewritten in place like the reference (decombine.py:648-654)


def test_fastq_check_and_empty_input_log(workdir, monkeypatch, stage):
    monkeypatch.setattr(nat, "decombine", _oracle_device(stage))
    out = workdir / "o"
    out.mkdir()
    empty = workdir / "empty_merge.fq"
    empty.write_text("")
    args = dio.create_args_dict(infile=str(empty), chain="b", bc_read="R2", tagfastadir=str(workdir / "tags"),
                                outpath=str(out) + os.sep)
    date = time.strftime("%Y_%m_%d")
    for n in ("", "2"):   # the second run must open ..._Summary2.csv (reference tests/test_decombine.py:75-94)
        with pytest.raises(ValueError, match="fewer than four lines"):
            dec.decombinator(dict(args))
        log = out / "Logs" / f"{date}_beta_empty_merge_Decombinator_Summary{n}.csv"
        assert log.read_text() == "OutputFile,empty_beta\nNumberReadsInput,0\n"
    bad = workdir / "bad_1.fq"
    bad.write_text("@r\nACGT\n+\nIIII\n>r2\nACGT\n+\nIIII\n")
    with pytest.raises(ValueError, match="Expected @ symbol"):
        dec.decombinator(dict(args, infile=str(bad)))
    bad.write_text("@r\nACGT\n+\nIIII\n@r2\nACGT\n+\nIII\n")
    with pytest.raises(ValueError, match="read quality"):
        dec.decombinator(dict(args, infile=str(bad)))


def test_cli_parser_flags():
    a = dio.cli_args(["decombine", "-in", "x_1.fq", "-br", "R2", "-c", "b", "-bl", "30", "-or", "both", "-N",
                      "-tg", "original", "-sp", "mouse", "-ln", "100", "-tfdir", "T", "-dz", "-op", "o/"])
    assert (a["command"], a["bclength"], a["orientation"], a["allowNs"], a["tags"], a["species"], a["lenthreshold"],
            a["tagfastadir"], a["dontgzip"], a["outpath"]) == ("decombine", 30, "both", True, "original", "mouse",
                                                                100, "T", True, "o/")
    p = dio.cli_args(["pipeline", "-in", "x_1.fq", "-c", "b", "-br", "R2", "-bl", "42", "-ol", "M13"])
    assert p["command"] == "pipeline" and p["oligo"] == "M13"


# ---- the reference's own golden .n12 (tests/resources/dcr_TINY_1_{alpha,beta}.n12) -----------
# The real tag files are not available offline; oracle/rebuild_tiny_tagset.py reconstructs the
# part of the human extended set that the fixtures pin.  With it, every row of the reference's
# fixture must come out, in order (reference tests/test_pipeline.py:63-84, test_subparsers.py:104-125).

def _tiny(chain_name, tmp_path):
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"tiny_{chain_name}.json")))
    ts = fx["tagset"]
    t = synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"],
                     v_jumps=ts["v_jumps"], v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"],
                     j_jumps=ts["j_jumps"], j_names=ts["j_names"], j_regions=ts["j_regions"])
    t.write(str(tmp_path / "tags"))
    (tmp_path / "TINY_1.fq").write_text(fx["fastq_r1"])
    (tmp_path / "TINY_2.fq").write_text(fx["fastq_r2"])
    args = dio.create_args_dict(infile=str(tmp_path / "TINY_1.fq"), chain=ts["chain"], bc_read="R2", dontgzip=True,
                                dontcount=True, tagfastadir=str(tmp_path / "tags"), outpath=str(tmp_path) + os.sep,
                                command="decombine")
    return fx, args


def _check_tiny(fx, args):
    rows = dec.decombinator(args)
    assert len(fx["reproduced_fixture_rows"]) == len(fx["reference_fixture_rows"])   # 35/35 alpha, 48/48 beta
    assert rows == fx["reference_fixture_rows"]
    assert rows == fx["rows_with_reconstructed_tagset"]
    for k, v in fx["counts_with_reconstructed_tagset"].items():
        if k not in ("chain_detected",):
            assert dec.counts[k] == v, k
    out = dio.write_out_intermediate(rows, args, ".n12")
    assert os.path.basename(out) == f"dcr_TINY_1_{dec.chainnams[args['chain']]}.n12"


@pytest.mark.parametrize("chain_name", ["alpha", "beta"])
def test_reference_n12_fixture_with_oracle_as_device(chain_name, tmp_path, monkeypatch):
    fx, args = _tiny(chain_name, tmp_path)
    monkeypatch.setattr(nat, "decombine", _oracle_device(fx))
    _check_tiny(fx, args)


@pytest.mark.gpu
@pytest.mark.parametrize("chain_name", ["alpha", "beta"])
def test_reference_n12_fixture_through_hip_path(chain_name, tmp_path):
    fx, args = _tiny(chain_name, tmp_path)
    _check_tiny(fx, args)


# ---- input variants (tests/golden/stage_variants.json, oracle/gen_stage_variants.py) --------------
VARIANTS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "stage_variants.json")))
VARIANT_IDS = [r["name"] for r in VARIANTS["runs"]]


def _run_variant(run, tmp_path, batch_reads, monkeypatch):
    import gzip
    ts = VARIANTS["tagset"]
    t = synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"],
                     v_jumps=ts["v_jumps"], v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"],
                     j_jumps=ts["j_jumps"], j_names=ts["j_names"], j_regions=ts["j_regions"])
    t.write(str(tmp_path / "tags"))
    ext = ".fq.gz" if run["gz"] else ".fq"
    op = gzip.open if run["gz"] else open
    for k, text in (("1", run["fastq_r1"]), ("2", run["fastq_r2"])):
        if text is not None:
            with op(tmp_path / f"VAR_{k}{ext}", "wb") as f:
                f.write(text.encode())
    monkeypatch.setattr(dec, "BATCH_READS", batch_reads)
    dec.counts.clear()
    args = dio.create_args_dict(infile=str(tmp_path / f"VAR_1{ext}"), chain="b", dontgzip=True, dontcount=True,
                                dontcheck=True, suppresssummary=True, tagfastadir=str(tmp_path / "tags"),
                                outpath=str(tmp_path) + os.sep, command="decombine", **run["args"])
    rows = dec.decombinator(args)
    assert rows == run["rows"]
    got = {k: v for k, v in dec.counts.items() if isinstance(v, int) and v}
    exp = {k: v for k, v in run["counts"].items() if v}
    assert got == exp


@pytest.mark.parametrize("batch_reads", [64, 1 << 20])
@pytest.mark.parametrize("k", range(len(VARIANT_IDS)), ids=VARIANT_IDS)
def test_stage_variants_with_oracle_as_device(k, batch_reads, tmp_path, monkeypatch):
    monkeypatch.setattr(nat, "decombine", _oracle_device(VARIANTS))
    _run_variant(VARIANTS["runs"][k], tmp_path, batch_reads, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(VARIANT_IDS)), ids=VARIANT_IDS)
def test_stage_variants_through_hip_path(k, tmp_path, monkeypatch):
    _run_variant(VARIANTS["runs"][k], tmp_path, 100, monkeypatch)


def test_fasta_record_raises_like_the_reference(tmp_path, monkeypatch):
    """A record without quality: the reference fails with TypeError on record[2][...]."""
    ts = VARIANTS["tagset"]
    t = synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"],
                     v_jumps=ts["v_jumps"], v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"],
                     j_jumps=ts["j_jumps"], j_names=ts["j_names"], j_regions=ts["j_regions"])
    t.write(str(tmp_path / "tags"))
    (tmp_path / "F_1.fq").write_text("@a\nACGT\n+\nIIII\n>b\nACGT\n")
    (tmp_path / "F_2.fq").write_text("@a\nACGT\n+\nIIII\n>b\nACGT\n")
    monkeypatch.setattr(nat, "decombine", _oracle_device(VARIANTS))
    for bc_read in ("R1", "R2"):
        args = dio.create_args_dict(infile=str(tmp_path / "F_1.fq"), chain="b", dontgzip=True, dontcount=True,
                                    dontcheck=True, suppresssummary=True, tagfastadir=str(tmp_path / "tags"),
                                    outpath=str(tmp_path) + os.sep, command="decombine", bc_read=bc_read, bclength=2)
        if bc_read == "R1":
            (tmp_path / "F_1.fq").write_text(">b\nACGT\n@a\nACGT\n+\nIIII\n")
        with pytest.raises(TypeError):
            dec.decombinator(args)


def test_n12rows_behaves_like_the_reference_list():
    """decombinator() returns N12Rows: iteration, len, indexing, slicing and == with a list of lists;
    blobs (native assembly) and lists (per-row path) can be mixed; write_text gives the .n12 text."""
    rows = dec.N12Rows()
    assert len(rows) == 0 and list(rows) == [] and rows == []
    rows._add_blob(bytearray(b"1, 2, 0, 3, AC, id1, ACGT, IIII, AAA, III\n4, 5, 1, 0, , id2, GG, II, CCC, III\n"), 2)
    rows.extend([["7", "8", "0", "0", "T", "id3", "TT", "II", "GGG", "III"]])
    rows.append(["9", "1", "2", "3", "", "id4", "A", "I", "TTT", "III"])
    expect = [["1", "2", "0", "3", "AC", "id1", "ACGT", "IIII", "AAA", "III"],
              ["4", "5", "1", "0", "", "id2", "GG", "II", "CCC", "III"],
              ["7", "8", "0", "0", "T", "id3", "TT", "II", "GGG", "III"],
              ["9", "1", "2", "3", "", "id4", "A", "I", "TTT", "III"]]
    assert len(rows) == 4 and rows == expect and expect == rows and list(rows) == expect
    assert rows[1] == expect[1] and rows[-1] == expect[-1] and rows[1:3] == expect[1:3]
    assert rows != expect[:3]
    buf = _io.BytesIO()
    text = _io.TextIOWrapper(buf, encoding="utf-8", newline="")
    rows.write_text(text, ", ")
    text.flush()
    assert buf.getvalue().decode() == "".join(", ".join(r) + "\n" for r in expect)


def test_clip_is_pythons_slice_on_spans():
    """_clip(off, len, lo, hi): the span of s[lo:hi] inside every span of a batch, in the arrays' own types."""
    from decombinator_amd.decombine import _clip
    rng = np.random.default_rng(5)
    off = rng.integers(0, 1 << 40, 500, dtype=np.uint64)
    ln = rng.integers(0, 80, 500).astype(np.uint32)
    for lo, hi in [(0, None), (0, 42), (42, None), (42, 73), (6, 6), (0, 0), (79, 200)]:
        o, l = _clip(off, ln, lo, hi)
        assert o.dtype == np.uint64 and l.dtype == np.uint32
        for k in range(len(ln)):
            want = range(int(ln[k]))[lo:hi]
            assert int(l[k]) == len(want)
            if len(want):
                assert int(o[k]) == int(off[k]) + want[0]


def test_rows_blob_is_bytes_filled_by_the_library():
    """assemble_rows_blob hands back a bytes object (not zeroed first by Python: the library's writers touch its pages) whose
    content is the rows' text, byte for byte what the per-row Python assembly gives."""
    n = 3000
    rng = np.random.default_rng(9)
    rec = np.zeros(n, dtype=nat.RECORD_DTYPE)
    ok = rng.random(n) < 0.5
    rec["status"] = np.where(ok, 0, 3)
    rec["v"], rec["j"] = rng.integers(0, 60, n), rng.integers(0, 13, n)
    rec["v_start"] = rng.integers(0, 30, n)
    rec["j_end"] = rec["v_start"] + rng.integers(20, 60, n)
    rec["ins_start"] = rec["v_start"] + 5
    rec["ins_len"] = rng.integers(0, 10, n)
    rec["vdel"], rec["jdel"], rec["frame"] = 2, 3, 1

    def spans(length):
        text = bytes(rng.integers(65, 90, n * length, dtype=np.uint8))
        return text, np.arange(n, dtype=np.uint64) * length, np.full(n, length, dtype=np.uint32)
    vdj, qual, ident, bc, bcq = spans(100), spans(100), spans(20), spans(12), spans(12)
    blob, rows = nat.assemble_rows_blob(rec, vdj, qual, ident, bc, bcq)
    assert isinstance(blob, bytes) and rows == int(ok.sum())
    want = []
    for k in np.nonzero(ok)[0]:
        s = vdj[0][int(vdj[1][k]):int(vdj[1][k]) + 100].decode()
        q = qual[0][int(qual[1][k]):int(qual[1][k]) + 100].decode()
        r = rec[k]
        ins = s[int(r["ins_start"]):int(r["ins_start"]) + int(r["ins_len"])]
        a, b = int(r["v_start"]), int(r["j_end"])
        fields = [str(int(r["v"])), str(int(r["j"])), str(int(r["vdel"])), str(int(r["jdel"])), ins,
                  ident[0][int(ident[1][k]):int(ident[1][k]) + 20].decode(), s[a:b], q[a:b],
                  bc[0][int(bc[1][k]):int(bc[1][k]) + 12].decode(), bcq[0][int(bcq[1][k]):int(bcq[1][k]) + 12].decode()]
        want.append(", ".join(fields) + "\n")
    assert blob.decode() == "".join(want)
