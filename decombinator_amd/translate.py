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


def translate_nt(seq: str) -> str:
    """str(Seq(seq).translate()) (translate.py:307-309): standard table, '*' for stops, 'X' for a codon holding
    anything but ACGT/U, a trailing partial codon dropped."""
    s = seq.upper().replace("U", "T")
    return "".join(_CODON.get(s[i:i + 3], "X") for i in range(0, len(s) - len(s) % 3, 3))


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


def get_cdr3(dcr, headers, inputargs, genes: GeneInfo | None = None):
    """translate.py:257-357, step for step; returns the dict of output fields."""
    G = genes if genes is not None else _genes
    if G is None:
        raise RuntimeError("set_gene_information() first (the reference's import_gene_information)")
    out_data = coll.defaultdict()
    for f in headers:
        out_data[f] = ""
    out_data["decombinator_id"] = (",".join(dcr) if inputargs["command"] == "translate" else ", ".join(dcr))   # :271-274
    out_data["rev_comp"] = "F"
    start_cdr3 = 0
    end_cdr3 = 0
    v, j, vdel, jdel = int(dcr[0]), int(dcr[1]), int(dcr[2]), int(dcr[3])
    ins_nt = dcr[4][1:] if inputargs["command"] == "translate" else dcr[4]                                     # :287-290
    out_data["v_call"] = G.v_names[v].split("*")[0]
    out_data["j_call"] = G.j_names[j].split("*")[0]
    v_used = G.v_regions[v] if vdel == 0 else G.v_regions[v][:-vdel]                                            # :296-299
    j_used = G.j_regions[j][jdel:]
    out_data["sequence"] = "".join([v_used, ins_nt, j_used])
    out_data["sequence_aa"] = translate_nt(out_data["sequence"])
    if (len(out_data["sequence"]) - 1) % 3 == 0:                                                               # :312-317 (the reference's frame test, literally)
        out_data["productive"] = "T"
        out_data["vj_in_frame"] = "T"
    else:
        out_data["productive"] = "F"
        out_data["vj_in_frame"] = "F"
    if "*" in out_data["sequence_aa"]:                                                                         # :320-324
        out_data["productive"] = "F"
        out_data["stop_codon"] = "T"
    else:
        out_data["stop_codon"] = "F"
    if out_data["sequence_aa"][G.v_translate_position[v] - 1] == G.v_translate_residue[v]:                      # :327-335 (IndexError like the reference when the sequence is too short)
        start_cdr3 = G.v_translate_position[v] - 1
        out_data["conserved_c"] = "T"
    else:
        out_data["productive"] = "F"
        out_data["conserved_c"] = "F"
    downstream_c = out_data["sequence_aa"][start_cdr3:]
    site = downstream_c[G.j_translate_position[j]:G.j_translate_position[j] + 4]                                # :341
    if re.findall(G.j_translate_residue[j], site):
        end_cdr3 = len(downstream_c) + G.j_translate_position[j] + start_cdr3 + 1
        out_data["conserved_f"] = "T"
    else:
        out_data["productive"] = "F"
        out_data["conserved_f"] = "F"
    if out_data["productive"] == "T":                                                                          # :350-355
        out_data["junction_aa"] = out_data["sequence_aa"][start_cdr3:end_cdr3]
        out_data["junction"] = out_data["sequence"][start_cdr3 * 3:3 * end_cdr3]
        out_data["cdr1_aa"] = G.v_cdr1[v]
        out_data["cdr2_aa"] = G.v_cdr2[v]
    return out_data
