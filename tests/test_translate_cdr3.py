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
