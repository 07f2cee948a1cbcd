"""`translate.get_cdr3` of the reference (src/decombinator/translate.py:257-357): the
productivity call and CDR3 extraction for one five-field DCR.  Pure table look-ups and a codon translation,
run once per UNIQUE DCR after `collapse`: host code (BASELINE.json keeps everything after `decombine` on the
host), kept here so that the AIRR fields of the reference's `.tsv` can be produced from this build's output.

Same name, arguments and output fields as the reference.  The gene tables the reference keeps in module
globals (translate.py:163-254: regions, names, conserved-residue positions and motifs, functionality, germline
CDR1/2) are a `GeneInfo` here, set with `set_gene_information` (or passed as `genes=`).
"""
from __future__ import annotations

import collections as coll
import re
from dataclasses import dataclass, field
from typing import List

out_headers = [
    "sequence_id", "v_call", "d_call", "j_call", "junction_aa", "duplicate_count", "sequence", "junction",
    "decombinator_id", "rev_comp", "productive", "sequence_aa", "cdr1_aa", "cdr2_aa", "vj_in_frame", "stop_codon",
    "conserved_c", "conserved_f", "sequence_alignment", "germline_alignment", "v_cigar", "d_cigar", "j_cigar",
    "av_UMI_cluster_size",
]  # translate.py:360-385

_BASES = "TCAG"
_AAS = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"      # NCBI table 1 (Bio.Seq.translate's default)
_CODON = {a + b + c: _AAS[16 * i + 4 * j + k] for i, a in enumerate(_BASES) for j, b in enumerate(_BASES)
          for k, c in enumerate(_BASES)}


_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "M": "AC", "R": "AG", "W": "AT", "S": "CG", "Y": "CT", "K": "GT",
          "V": "ACG", "H": "ACT", "D": "AGT", "B": "CGT", "X": "ACGT", "N": "ACGT"}     # Bio.Data.IUPACData.ambiguous_dna_values
_AMBIGUOUS_AA = {"B": set("DN"), "Z": set("EQ"), "J": set("IL")}                      # ... .extended_protein_values, the two-residue letters
_AMB_CACHE: dict = {}


def _ambiguous_codon(codon: str) -> str:
    """Bio.Seq.translate's answer for a codon that holds an ambiguity code (Bio.Data.CodonTable.AmbiguousForwardTable and
    the fall-backs of Bio.Seq._translate_str, biopython 1.84): the residue every concrete codon it stands for gives (GCN ->
    A, YTA -> L), B / Z / J when they give exactly D+N / E+Q / I+L (RAY -> B), '*' when all of them are stops (TAR, TRA), and X
    otherwise (a mix of residues, or of stops and residues).  A letter that is no nucleotide code at all raises, as
    Biopython does."""
    hit = _AMB_CACHE.get(codon)
    if hit is not None:
        return hit
    try:
        opts = [_IUPAC[c] for c in codon]
    except KeyError:
        raise ValueError(f"Codon '{codon}' is invalid") from None          # Bio.Data.CodonTable.TranslationError is a ValueError
    aas = {_CODON[a + b + c] for a in opts[0] for b in opts[1] for c in opts[2]}
    if aas == {"*"}:
        out = "*"
    elif "*" in aas:
        out = "X"
    elif len(aas) == 1:
        out = next(iter(aas))
    else:
        out = next((k for k, v in _AMBIGUOUS_AA.items() if aas == v), "X")
    _AMB_CACHE[codon] = out
    return out


def translate_nt(seq: str) -> str:
    """str(Seq(seq).translate()) (translate.py:307-309): standard table, '*' for stops, a trailing partial codon dropped;
    codons with IUPAC ambiguity codes as Biopython resolves them (_ambiguous_codon)."""
    s = seq.upper().replace("U", "T")
    out = []
    for i in range(0, len(s) - len(s) % 3, 3):
        c = s[i:i + 3]
        aa = _CODON.get(c)
        out.append(aa if aa is not None else _ambiguous_codon(c))
    return "".join(out)


@dataclass
class GeneInfo:
    """What import_gene_information() returns (translate.py:163-254)."""
    v_regions: List[str] = field(default_factory=list)
    j_regions: List[str] = field(default_factory=list)
    v_names: List[str] = field(default_factory=list)
    j_names: List[str] = field(default_factory=list)
    v_translate_position: List[int] = field(default_factory=list)
    v_translate_residue: List[str] = field(default_factory=list)
    j_translate_position: List[int] = field(default_factory=list)
    j_translate_residue: List[str] = field(default_factory=list)
    v_functionality: List[str] = field(default_factory=list)
    j_functionality: List[str] = field(default_factory=list)
    v_cdr1: List[str] = field(default_factory=list)
    v_cdr2: List[str] = field(default_factory=list)


_genes: GeneInfo | None = None


def set_gene_information(genes: GeneInfo) -> None:
    global _genes
    _genes = genes


def _native_genes(G: GeneInfo):
    """The tables of dcrx_cdr3_batch for a GeneInfo, built once per GeneInfo object."""
    from . import _native as nat
    cached = getattr(G, "_native", None)
    if cached is None:
        cached = nat.Cdr3Genes(G.v_regions, G.j_regions, G.v_translate_position, G.v_translate_residue,
                               G.j_translate_position, G.j_translate_residue)
        object.__setattr__(G, "_native", cached)
    return cached


def cdr3_batch(dcrs, headers, inputargs, genes: GeneInfo | None = None):
    """get_cdr3 for many DCRs at once: one call into libdcrx (dcrx_cdr3_batch, include/dcrx.h — the reference's translate.py:257-357
    restated natively) and the rows' dictionaries filled from what it returns.  Raises what the reference raises for the first
    row that would: IndexError (a gene index outside its table, a translation shorter than the V gene's residue position),
    ValueError (a letter that is no nucleotide code)."""
    from . import _native as nat
    G = genes if genes is not None else _genes
    if G is None:
        raise RuntimeError("set_gene_information() first (the reference's import_gene_information)")
    from_file = inputargs["command"] == "translate"       # rows of a `.freq` file: fields joined by "," and the insert behind a blank (:271-274, :287-290)
    ints = [[int(d[k]) for d in dcrs] for k in range(4)]
    inserts = [(d[4][1:] if from_file else d[4]) for d in dcrs]
    rows, text = nat.cdr3_batch(_native_genes(G), ints[0], ints[1], ints[2], ints[3], inserts)
    out = []
    flag = ("F", "T")
    for d, r in zip(dcrs, rows):
        if r["status"] == nat.CDR3_INDEX_ERROR:
            raise IndexError("string index out of range")
        if r["status"] == nat.CDR3_BAD_CODON:
            at = int(r["bad_codon_at"])
            v, j, vdel, jdel = (int(x) for x in d[:4])
            seq = "".join([G.v_regions[v] if vdel == 0 else G.v_regions[v][:-vdel], d[4][1:] if from_file else d[4], G.j_regions[j][jdel:]])
            raise ValueError(f"Codon '{seq[at:at + 3].upper().replace('U', 'T')}' is invalid")
        rec = coll.defaultdict()
        for f in headers:
            rec[f] = ""
        so, sl, ao, al = int(r["seq_off"]), int(r["seq_len"]), int(r["aa_off"]), int(r["aa_len"])
        rec["decombinator_id"] = (",".join(d) if from_file else ", ".join(d))
        rec["rev_comp"] = "F"
        rec["v_call"] = G.v_names[int(d[0])].split("*")[0]
        rec["j_call"] = G.j_names[int(d[1])].split("*")[0]
        rec["sequence"] = text[so:so + sl].decode("latin-1")
        rec["sequence_aa"] = text[ao:ao + al].decode("latin-1")
        productive, conserved_f = bool(r["productive"]), bool(r["conserved_f"])
        ja = (int(r["junction_aa_off"]), int(r["junction_aa_len"]))
        jn = (int(r["junction_off"]), int(r["junction_len"]))
        if r["status"] == nat.CDR3_MOTIF_LEFT:
            # the J gene's motif uses regular-expression syntax the library's matcher does not serve (alternation, groups,
            # repetition): the row's last step with Python's engine, as the reference runs it (translate.py:338-355)
            aa, start = rec["sequence_aa"], int(r["start_cdr3"])
            j = int(d[1])
            conserved_f = bool(re.findall(G.j_translate_residue[j], aa[ja[0]:ja[0] + ja[1]]))
            if conserved_f:
                end = len(aa[start:]) + G.j_translate_position[j] + start + 1
                ja = tuple(_span(len(aa), start, end))
                jn = tuple(_span(len(rec["sequence"]), start * 3, 3 * end))
            else:
                productive = False
        rec["productive"] = flag[productive]
        rec["vj_in_frame"] = flag[r["in_frame"]]
        rec["stop_codon"] = flag[r["stop"]]
        rec["conserved_c"] = flag[r["conserved_c"]]
        rec["conserved_f"] = flag[conserved_f]
        if productive:
            rec["junction_aa"] = rec["sequence_aa"][ja[0]:ja[0] + ja[1]]
            rec["junction"] = rec["sequence"][jn[0]:jn[0] + jn[1]]
            rec["cdr1_aa"] = G.v_cdr1[int(d[0])]
            rec["cdr2_aa"] = G.v_cdr2[int(d[0])]
        out.append(rec)
    return out


def _span(n, start, stop):
    """(offset, length) of Python's s[start:stop] on a string of n characters."""
    lo, hi, _ = slice(start, stop).indices(n)
    return lo, max(0, hi - lo)


def get_cdr3(dcr, headers, inputargs, genes: GeneInfo | None = None):
    """The reference's translate.get_cdr3 (translate.py:257-357) for one DCR: same name, arguments and output fields; the work is
    dcrx_cdr3_batch's (cdr3_batch above, a batch of one)."""
    return cdr3_batch([dcr], headers, inputargs, genes)[0]


# ---- the stage around get_cdr3 (translate.py:163-254 and :388-533), for rows this build or the reference's collapse produced ----
counts = coll.Counter()


def _read_fasta(path):
    """(id, sequence) per record, in file order (the reference: SeqIO.parse(..., "fasta"), translate.py:182)."""
    out, name, seq = [], None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(seq)))
                name, seq = line[1:].split()[0] if line[1:].split() else "", []
            elif name is not None:
                seq.append(line.strip())
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def import_gene_information(inputargs) -> GeneInfo:
    """translate.py:163-254: regions and names from the FASTA files, conserved-residue positions, motifs and functionality from
    the `.translate` files, germline CDR1 / CDR2 from the `.cdrs` file (human only).  Files are looked for in the working
    directory, then in inputargs["tagfastadir"] (this build is offline: no download)."""
    from .decombine import read_tcr_file
    if inputargs["species"] not in ("human", "mouse"):
        print("Species not recognised. Please select either 'human' (default) or 'mouse'.\n"
              "If mouse is required by default, consider changing the default value in the script.")
        raise SystemExit
    g = GeneInfo()
    chain = inputargs["chain"]
    for gene in ("v", "j"):
        recs = _read_fasta(read_tcr_file(inputargs["species"], inputargs["tags"], gene, "fasta", inputargs["tagfastadir"], chain))
        setattr(g, gene + "_regions", [sq.upper() for _, sq in recs])
        setattr(g, gene + "_names", [rid.upper().split("|")[1] for rid, _ in recs])
        with open(read_tcr_file(inputargs["species"], inputargs["tags"], gene, "translate", inputargs["tagfastadir"], chain)) as fh:
            rows = [x.rstrip() for x in fh]
        setattr(g, gene + "_translate_position", [int(x.split(",")[1]) for x in rows])
        setattr(g, gene + "_translate_residue", [x.split(",")[2] for x in rows])
        setattr(g, gene + "_functionality", [x.split(",")[3] for x in rows])
    if inputargs["species"] == "human":
        with open(read_tcr_file(inputargs["species"], inputargs["tags"], "v", "cdrs", inputargs["tagfastadir"], chain)) as fh:
            cdr = [x.rstrip() for x in fh]
        g.v_cdr1, g.v_cdr2 = [x.split(" ")[1] for x in cdr], [x.split(" ")[2] for x in cdr]
    else:
        g.v_cdr1, g.v_cdr2 = [""] * len(g.v_regions), [""] * len(g.v_regions)
    return g


_CHAINS = {"A": "a", "ALPHA": "a", "TRA": "a", "TCRA": "a", "B": "b", "BETA": "b", "TRB": "b", "TCRB": "b",
           "G": "g", "GAMMA": "g", "TRG": "g", "TCRG": "g", "D": "d", "DELTA": "d", "TRD": "d", "TCRD": "d"}


def cdr3translator(inputargs: dict, data=None) -> list:
    """translate.py:388-533 without the file writing: one row of `out_headers` fields per input DCR (non-productive ones too
    unless inputargs["nonproductivefilter"]).  `data`: rows whose first five fields are the DCR, then the frequency and the
    average UMI cluster size (what the reference's collapse returns); with inputargs["command"] == "translate" the rows are
    read from the comma-separated file inputargs["infile"] (a `.freq`).  inputargs["nobarcoding"]: every row counts once."""
    import gzip
    global counts
    counts = coll.Counter()
    if not inputargs.get("chain"):
        found = [x for x in ("alpha", "beta", "gamma", "delta") if x in inputargs["infile"].lower()]      # :394-407
        if len(found) != 1:
            print("TCR chain not recognised. Please choose from a/b/g/d (case-insensitive).")
            raise SystemExit
        chain = found[0][0]
    else:
        chain = _CHAINS.get(str(inputargs["chain"]).upper())
        if chain is None:
            print("TCR chain not recognised. Please choose from a/b/g/d (case-insensitive).")
            raise SystemExit
    inputargs["chain"] = chain
    set_gene_information(import_gene_information(inputargs))
    if inputargs["command"] == "translate":
        fh = (gzip.open if inputargs["infile"].endswith(".gz") else open)(inputargs["infile"], "rt")
        rows = list(fh)
        fh.close()
    else:
        rows = data
    # The rows' own checks come row by row in the reference, each in front of that row's get_cdr3 (:430-470): the rows up to
    # the first one that fails a check are translated in ONE call of the batch entry (an error inside get_cdr3 of an earlier
    # row is then raised first, as there), and only then the check's own exit.
    pending, meta, stop = [], [], None
    for line in rows:
        counts["line_count"] += 1
        if inputargs["command"] == "translate":
            tcr = line.rstrip().split(",")
            tcr[5], tcr[6] = int(tcr[5]), int(tcr[6])
        else:
            tcr = line
        if inputargs.get("nobarcoding"):
            frequency, cluster = 1, ""
        else:
            if not isinstance(tcr[5], int):
                stop = ("TCR frequency could not be detected. If using non-barcoded data, please include the additional '-nbc' "
                        "argument when running CDR3translator.")
                break
            frequency = tcr[5]
            cluster = tcr[6] if len(tcr) > 6 and isinstance(tcr[6], (int, float)) else ""
        pending.append(tcr[:5])
        meta.append((str(counts["line_count"]), frequency, cluster))
    out = []
    recs = cdr3_batch(pending, out_headers, inputargs) if pending else []
    for rec, (sid, frequency, cluster) in zip(recs, meta):
        rec["sequence_id"] = sid
        rec["duplicate_count"] = frequency
        rec["av_UMI_cluster_size"] = cluster
        if rec["productive"] == "T":
            counts["prod_recomb"] += 1
            out.append([rec[x] for x in out_headers])
        else:
            counts["NP_count"] += 1
            if not inputargs.get("nonproductivefilter"):
                out.append([rec[x] for x in out_headers])
    if stop is not None:
        print(stop)
        raise SystemExit
    return out
