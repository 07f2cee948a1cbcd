"""The intermediate files' gzip step (reference io.py:497-506): libdcrx's threaded multi-member writer behind
write_out_intermediate() — the decompressed bytes are what the dontgzip path writes, any gzip reader reads them."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from decombinator_amd import _native as nat, decombine as dcr, io as dio


def _rows_text(n, seed):
    rng = np.random.default_rng(seed)
    lines = []
    for k in range(n):
        ins = "".join("ACGT"[i] for i in rng.integers(0, 4, size=int(rng.integers(0, 12))))
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, size=60))
        q = "".join(chr(33 + int(i)) for i in rng.integers(0, 41, size=60))
        lines.append(", ".join([str(int(rng.integers(0, 60))), str(int(rng.integers(0, 13))), str(int(rng.integers(0, 9))),
                                str(int(rng.integers(0, 9))), ins, f"read{k}:é", seq, q, "ACGTACGTACGT", "IIIIIIIIIIII"]))
    return lines


@pytest.mark.parametrize("level", [1, 6, 9])
def test_writer_round_trip_across_piece_boundaries(tmp_path, level):
    rng = np.random.default_rng(level)
    data = bytes(rng.integers(65, 70, size=9_500_000, dtype=np.uint8))       # more than two 4 MB pieces
    p = str(tmp_path / "x.gz")
    with nat.GzipWriter(p, level=level, n_threads=3) as g:
        g.write(data[:7])
        g.write(b"")
        g.write(memoryview(data)[7:5_000_001])
        g.write(bytearray(data[5_000_001:]))
    assert gzip.open(p, "rb").read() == data
    assert subprocess.run(["gzip", "-dc", p], capture_output=True, check=True).stdout == data


def test_empty_file_is_a_gzip_stream(tmp_path):
    p = str(tmp_path / "e.gz")
    nat.GzipWriter(p).close()
    assert os.path.getsize(p) > 0 and gzip.open(p, "rb").read() == b""


def test_errors(tmp_path):
    with pytest.raises(nat.DcrxError):
        nat.GzipWriter(str(tmp_path / "no" / "such" / "dir" / "x.gz"))
    with pytest.raises(nat.DcrxError):
        nat.GzipWriter(str(tmp_path / "x.gz"), level=0)


@pytest.mark.parametrize("kind", ["n12rows", "lists"])
def test_write_out_intermediate_gz_equals_plain(tmp_path, kind):
    lines = _rows_text(3000, 5)
    if kind == "n12rows":          # the decombine stage's rows: text blobs as libdcrx assembled them, plus a chunk of lists
        data = dcr.N12Rows()
        blob = ("\n".join(l.replace(", ", dcr._FIELD_SEP) for l in lines[:2000]) + "\n").encode("utf-8")
        data._add_blob(blob, 2000)
        data.extend([l.split(", ") for l in lines[2000:]])
    else:
        data = [l.split(", ") for l in lines]
    outs = {}
    for dz in (True, False):
        d = tmp_path / ("plain" if dz else "gz")
        d.mkdir()
        args = {"infile": "SAMPLE_1.fq.gz", "command": "decombine", "outpath": str(d) + os.sep, "prefix": "dcr_", "chain": "b",
                "dontgzip": dz}
        outs[dz] = dio.write_out_intermediate(data, args, ".n12")
    assert outs[True].endswith(".n12") and outs[False].endswith(".n12.gz")
    assert not os.path.exists(outs[False][:-3])                       # no plain file is left beside the .gz
    want = open(outs[True], "rb").read()
    assert want.decode("utf-8").splitlines() == lines
    assert gzip.open(outs[False], "rb").read() == want
    assert [l.rstrip("\n") for l in gzip.open(outs[False], "rt")] == lines      # as collapse's opener reads it back
    assert (os.stat(outs[False]).st_mode & 0o777) == (os.stat(outs[True]).st_mode & 0o777)
