"""translate.get_cdr3 (decombinator_amd/translate.py) against tests/golden/translate_cdr3.json, generated from
the imported reference (oracle/gen_translate_golden.py): 800 DCRs over synthetic gene tables, both `command`
values, every output field."""
import json
import os

import pytest

from decombinator_amd import translate

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "translate_cdr3.json")


def test_get_cdr3_matches_the_reference_on_every_field():
    fx = json.load(open(GOLDEN))
    translate.set_gene_information(translate.GeneInfo(**fx["genes"]))
    assert fx["headers"] == translate.out_headers
    n_prod = 0
    for cs in fx["cases"]:
        if cs["expect"] == "IndexError":
            with pytest.raises(IndexError):
                translate.get_cdr3(cs["dcr"], translate.out_headers, {"command": cs["command"]})
            continue
        got = dict(translate.get_cdr3(cs["dcr"], translate.out_headers, {"command": cs["command"]}))
        assert got == cs["expect"], cs
        n_prod += got["productive"] == "T"
    assert n_prod > 50


def test_translate_nt_standard_table_and_partial_codon():
    assert translate.translate_nt("ATGGCCTAA") == "MA*"
    assert translate.translate_nt("ATGGCCTA") == "MA"          # trailing partial codon dropped
    assert translate.translate_nt("ATGNNNTGG") == "MXW"


def test_ambiguous_codons_as_biopython_resolves_them():
    """Bio.Seq.translate resolves a codon with ambiguity codes when the residue is determined (hand-checked against
    Bio.Data.CodonTable's AmbiguousForwardTable rules and the standard table)."""
    t = translate.translate_nt
    assert t("GCN") == "A" and t("CTN") == "L" and t("GGN") == "G" and t("ACN") == "T"       # four-fold degenerate third base
    assert t("TAR") == "*" and t("TRA") == "*"                                                 # every codon it stands for is a stop
    assert t("RAY") == "B" and t("SAR") == "Z" and t("MTY") == "J"                             # D+N, E+Q, I+L (ATC, ATT, CTC, CTT)
    assert t("YTA") == "L" and t("MGR") == "R" and t("ATH") == "I"                             # several codons, one residue
    assert t("TGN") == "X" and t("NNN") == "X" and t("ANA") == "X"                             # stops mixed with residues / several residues
    assert t("ATGGCNTAR") == "MA*"
    with pytest.raises(ValueError):
        t("AT!")


def _write_gene_files(fx, d):
    g = fx["genes"]
    for gene, regs, names, pos, res, fun in (("V", g["v_regions"], g["v_names"], g["v_translate_position"], g["v_translate_residue"], g["v_functionality"]),
                                             ("J", g["j_regions"], g["j_names"], g["j_translate_position"], g["j_translate_residue"], g["j_functionality"])):
        with open(os.path.join(d, f"human_extended_TRB{gene}.fasta"), "w") as fh:
            for n, r in zip(names, regs):
                fh.write(f">X0|{n.lower()}|Homo sapiens\n{r.lower()}\n")           # (upper-cased on import, names from field 1: translate.py:186-191)
        with open(os.path.join(d, f"human_extended_TRB{gene}.translate"), "w") as fh:
            for n, p, r, f in zip(names, pos, res, fun):
                fh.write(f"{n},{p},{r},{f}\n")
    with open(os.path.join(d, "human_extended_TRBV.cdrs"), "w") as fh:
        for n, a, b in zip(g["v_names"], g["v_cdr1"], g["v_cdr2"]):
            fh.write(f"{n} {a} {b}\n")


def test_translate_stage_from_files_and_the_cli(tmp_path):
    """import_gene_information from `.fasta` / `.translate` / `.cdrs` files + cdr3translator over a `.freq` file + the `translate`
    sub-command: every row equals get_cdr3's reference-generated expectation, with the stage's own columns filled in."""
    from decombinator_amd import pipeline
    fx = json.load(open(GOLDEN))
    _write_gene_files(fx, str(tmp_path))
    cases = [c for c in fx["cases"] if c["command"] == "translate" and c["expect"] != "IndexError"][:200]
    with open(tmp_path / "x_beta.freq", "w") as fh:
        for k, c in enumerate(cases):
            fh.write(",".join(c["dcr"]) + f",{k + 1},{k % 5 + 1}\n")
    args = {"command": "translate", "infile": str(tmp_path / "x_beta.freq"), "chain": None, "species": "human", "tags": "extended",
            "tagfastadir": str(tmp_path), "nobarcoding": False, "nonproductivefilter": False}
    rows = translate.cdr3translator(args)
    assert args["chain"] == "b" and len(rows) == len(cases)
    for k, (row, c) in enumerate(zip(rows, cases)):
        want = dict(c["expect"], sequence_id=str(k + 1), duplicate_count=k + 1, av_UMI_cluster_size=k % 5 + 1)
        assert dict(zip(translate.out_headers, row)) == want
    assert translate.counts["prod_recomb"] + translate.counts["NP_count"] == len(cases)
    args2 = dict(args, nonproductivefilter=True, chain="TRB")
    assert len(translate.cdr3translator(args2)) == translate.counts["prod_recomb"]
    # the reference's write_out_translated (io.py:516-548): `.tsv.gz` by default, plain `.tsv` with -dz, mode 666
    import gzip
    import stat
    pipeline.main(["translate", "-in", str(tmp_path / "x_beta.freq"), "-tfdir", str(tmp_path), "-op", str(tmp_path) + os.sep])
    assert not os.path.exists(tmp_path / "x_beta.tsv")
    lines = gzip.open(tmp_path / "x_beta.tsv.gz", "rt").read().splitlines()
    assert lines[0].split("\t") == translate.out_headers and len(lines) == 1 + len(cases)
    assert lines[1].split("\t")[1] == rows[0][1]
    assert stat.S_IMODE(os.stat(tmp_path / "x_beta.tsv.gz").st_mode) == 0o666
    pipeline.main(["translate", "-in", str(tmp_path / "x_beta.freq"), "-tfdir", str(tmp_path), "-op", str(tmp_path) + os.sep, "-dz"])
    assert open(tmp_path / "x_beta.tsv").read().splitlines() == lines
    from decombinator_amd.io import write_out_translated
    name = write_out_translated([["a", None, 3]], ["x", "y", "z"], {"command": "translate", "infile": str(tmp_path / "n.freq"), "outpath": str(tmp_path) + os.sep,
                                                                     "dontgzip": True, "chain": "b"})
    assert open(name).read() == "x\ty\tz\na\t\t3\n"           # a missing value is an empty field, as DataFrame.to_csv writes it


STAGE_FX = os.path.join(os.path.dirname(GOLDEN), "translate_stage.json")


CODING_FX = os.path.join(os.path.dirname(GOLDEN), "translate_stage_coding.json")


def _fastq_to_cdr3(tmp_path, fixture=None):
    """FASTQ files -> decombinator() -> rows -> their unique DCRs -> cdr3translator (gene tables imported from files), against
    the reference's get_cdr3 on the same DCRs (tests/golden/translate_stage.json, oracle/gen_translate_golden.py)."""
    from decombinator_amd import decombine as dec, io as dio, synth
    fx = json.load(open(fixture or STAGE_FX))
    stage = fx if "stage" not in fx else json.load(open(os.path.join(os.path.dirname(GOLDEN), fx["stage"])))      # (the coding fixture holds its own files)
    ts = stage["tagset"]
    tags = tmp_path / "tags"
    synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"], v_jumps=ts["v_jumps"],
                 v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"], j_jumps=ts["j_jumps"],
                 j_names=ts["j_names"], j_regions=ts["j_regions"]).write(str(tags))
    g = fx["genes"]
    stem = f"{ts['species']}_{ts['tags']}_TR{ts['chain'].upper()}"
    for gene, names, pos, res, fun in (("V", g["v_names"], g["v_translate_position"], g["v_translate_residue"], g["v_functionality"]),
                                       ("J", g["j_names"], g["j_translate_position"], g["j_translate_residue"], g["j_functionality"])):
        with open(tags / f"{stem}{gene}.translate", "w") as fh:
            for n, p, r, f in zip(names, pos, res, fun):
                fh.write(f"{n},{p},{r},{f}\n")
    with open(tags / f"{stem}V.cdrs", "w") as fh:
        for n, a, b in zip(g["v_names"], g["v_cdr1"], g["v_cdr2"]):
            fh.write(f"{n} {a} {b}\n")
    (tmp_path / "SYNTH_1.fq").write_text(stage["fastq_r1"])
    (tmp_path / "SYNTH_2.fq").write_text(stage["fastq_r2"])
    args = dio.create_args_dict(infile=str(tmp_path / "SYNTH_1.fq"), chain=ts["chain"], bc_read="R2", dontgzip=True, dontcount=True,
                                orientation="reverse", allowNs=False, tagfastadir=str(tags), suppresssummary=True, dontcheck=True,
                                tags=ts["tags"], species=ts["species"], outpath=str(tmp_path) + os.sep, command="pipeline")
    rows = dec.decombinator(args)
    if "rows" in fx:
        assert [list(r) for r in rows] == fx["rows"]              # what the reference's decombinator() returned for these files
    seen, uniq = set(), []
    for r in rows:
        d = tuple(r[:5])
        if d not in seen:
            seen.add(d); uniq.append(list(d))
    assert uniq == fx["dcrs"]
    targs = dict(args, nobarcoding=True, nonproductivefilter=False)
    out = translate.cdr3translator(targs, data=uniq)
    got_genes = translate._genes
    assert got_genes.v_regions == g["v_regions"] and got_genes.j_names == g["j_names"] and got_genes.v_cdr2 == g["v_cdr2"]
    assert len(out) == len(fx["expect"])
    for k, (row, want) in enumerate(zip(out, fx["expect"])):
        assert want != "IndexError"
        want = dict(want, sequence_id=str(k + 1), duplicate_count=1, av_UMI_cluster_size="")
        assert dict(zip(translate.out_headers, row)) == want
    return sum(1 for e in fx["expect"] if e["productive"] == "T"), len(fx["expect"])


def test_fastq_to_cdr3_with_oracle_as_device(tmp_path, monkeypatch):
    from decombinator_amd import _native as nat
    from tests import test_host_stage as ths
    fx = json.load(open(STAGE_FX))
    stage = json.load(open(os.path.join(os.path.dirname(GOLDEN), fx["stage"])))
    monkeypatch.setattr(nat, "decombine", ths._oracle_device(stage))
    _fastq_to_cdr3(tmp_path)


@pytest.mark.gpu
def test_fastq_to_cdr3_through_hip_path(tmp_path):
    _fastq_to_cdr3(tmp_path)


def _coding_oracle_device(fx):
    from tests import golden_util as gu, parity_util as pu
    from decombinator_amd import _native as nat
    ot = gu.oracle_tables(fx["tagset"])

    def device(tables, batch, orientation="reverse", allow_ns=False, lenthreshold=130, flags=0):
        return pu.oracle_records(ot, nat.unpack_reads(batch), orientation, allow_ns, lenthreshold)
    return device


def test_fastq_to_productive_cdr3_with_oracle_as_device(tmp_path, monkeypatch):
    """The coding fixture (oracle/gen_translate_golden.py coding_stage_fixture: germlines that code, in-frame rearrangements):
    FASTQ -> decombine -> rows (= the reference's) -> get_cdr3 (= the reference's), most rows through the productive branch of
    translate.py:312-350."""
    from decombinator_amd import _native as nat
    fx = json.load(open(CODING_FX))
    monkeypatch.setattr(nat, "decombine", _coding_oracle_device(fx))
    prod, total = _fastq_to_cdr3(tmp_path, CODING_FX)
    assert prod >= 0.3 * total and total > 400


@pytest.mark.gpu
def test_fastq_to_productive_cdr3_through_hip_path(tmp_path):
    prod, total = _fastq_to_cdr3(tmp_path, CODING_FX)
    assert prod >= 0.3 * total and total > 400


def test_fixture_cases_with_ambiguity_codes_cover_every_kind_of_answer():
    """The ambiguous-codon rules of translate_nt are pinned by generated cases (the reference's get_cdr3 through the shim's
    translate), not by hand: the fixture holds junctions with X, with B / Z / J, and with a shared residue behind an ambiguous
    codon (those cases run in test_get_cdr3_matches_the_reference_on_every_field with all the others)."""
    fx = json.load(open(GOLDEN))
    amb = [c for c in fx["cases"] if c.get("ambiguous") and c["expect"] != "IndexError"]
    assert len(amb) >= 350
    jun = "".join(c["expect"]["junction_aa"] + c["expect"]["sequence_aa"] for c in amb)
    assert all(x in jun for x in "XBZJ")
    assert any(c["expect"]["productive"] == "T" for c in amb)
