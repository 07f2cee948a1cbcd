"""Helpers shared by the parity tests: load tests/golden fixtures, build oracle
tables from them, convert counters."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from oracle import oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORIENT = {"reverse": 0, "forward": 1, "both": 2}


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "dcr_*.json")))


def load(path):
    with open(path) as f:
        return json.load(f)


def oracle_tables(ts: dict) -> orc.OracleTables:
    return orc.OracleTables(ts["v_tags"], ts["v_jumps"], [r.upper() for r in ts["v_regions"]],
                            ts["j_tags"], ts["j_jumps"], [r.upper() for r in ts["j_regions"]],
                            ts["v_half_split"], ts["j_half_split"])


def counts_dict(arr) -> dict:
    """uint64[32] -> {reference Counter key: value} without zeros and without the
    build's own frame_forward tally."""
    return {n: int(arr[i]) for i, n in enumerate(orc.COUNTER_NAMES)
            if int(arr[i]) and n != "frame_forward"}


def expect_from_result(read_in_frame: str, res) -> list | None:
    if int(res["status"] if isinstance(res, np.void) else res.status) != 0:
        return None
    g = (lambda k: int(res[k])) if isinstance(res, np.void) else (lambda k: int(getattr(res, k)))
    return [g("v"), g("j"), g("vdel"), g("jdel"),
            read_in_frame[g("ins_start"):g("ins_start") + g("ins_len")], g("v_start"), g("j_end")]
