"""Arguments and writers of the `decombine` stage, with the names, defaults and file
formats of the reference's src/decombinator/io.py:

  create_args_dict        reference io.py:391-465 (same keys, same defaults)
  create_parser/cli_args  reference io.py:41-92, 95-383 (the common + decombine flags; the
                          collapse / translate flags as far as the front half of collapse and
                          the translate sub-command use them, the rest accepted so that existing
                          command lines parse)
  write_out_intermediate  reference io.py:480-513 (", "-joined rows, optional gzip, chmod 666)
"""
from __future__ import annotations

import argparse
import gzip
import os

from .decombine import __version__, sort_permissions


def create_args_dict(
    infile: str, chain: str, bc_read: str, suppresssummary: bool = False, dontgzip: bool = False,
    dontcheck: bool = False, dontcount: bool = False, extension: str = "n12", prefix: str = "dcr_",
    orientation: str = "reverse", tags: str = "extended", species: str = "human", allowNs: bool = False,
    lenthreshold: int = 130, tagfastadir: str = "Decombinator-Tags-FASTAs", nobarcoding: bool = False,
    bclength: int = 42, minbcQ: int = 20, bcQbelowmin: int = 1, avgQthreshold: int = 30,
    percentlevdist: int = 10, bcthreshold: int = 2, dontcheckinput: bool = False,
    barcodeduplication: bool = False, positionalbarcodes: bool = False, oligo: str = "M13",
    writeclusters: bool = False, UMIhistogram: bool = False, nonproductivefilter: bool = False,
    outpath: str = None, dontsave: bool = False, command: str = None, sampling_analysis: bool = False,
) -> dict:
    """The function-argument dictionary threaded through the stages (33 keys)."""
    return dict(
        infile=infile, chain=chain, bc_read=bc_read, suppresssummary=suppresssummary, dontgzip=dontgzip,
        dontcheck=dontcheck, dontcount=dontcount, extension=extension, prefix=prefix, orientation=orientation,
        tags=tags, species=species, allowNs=allowNs, lenthreshold=lenthreshold, tagfastadir=tagfastadir,
        nobarcoding=nobarcoding, bclength=bclength, minbcQ=minbcQ, bcQbelowmin=bcQbelowmin,
        avgQthreshold=avgQthreshold, percentlevdist=percentlevdist, bcthreshold=bcthreshold,
        dontcheckinput=dontcheckinput, barcodeduplication=barcodeduplication,
        positionalbarcodes=positionalbarcodes, oligo=oligo, writeclusters=writeclusters,
        UMIhistogram=UMIhistogram, nonproductivefilter=nonproductivefilter, outpath=outpath,
        dontsave=dontsave, command=command, sampling_analysis=sampling_analysis)


def _common(p: argparse.ArgumentParser):
    p.add_argument("-s", "--suppresssummary", action="store_true", help="Suppress the summary log")
    p.add_argument("-dz", "--dontgzip", action="store_true", help="Do not gzip the output files")
    p.add_argument("-dc", "--dontcount", action="store_true", help="Do not print the running count")
    p.add_argument("-op", "--outpath", type=str, default="", help="Output directory (default: cwd)")
    p.add_argument("-c", "--chain", type=str, help="TCR chain (a/b/g/d)")
    p.add_argument("-pf", "--prefix", type=str, default="dcr_", help='Output file prefix. Default "dcr_"')
    p.add_argument("-ds", "--dontsave", action="store_true", help="Do not save output files")
    p.add_argument("-sa", "--sampling_analysis", action="store_true", help="Keep the R2 V-gene tail per row")


def _decombine(p: argparse.ArgumentParser):
    p.add_argument("-in", "--infile", type=str, required=True, help="FASTQ file with the TCR reads")
    p.add_argument("-br", "--bc_read", type=str, required=True, help="Which read holds the barcode (R1/R2)")
    p.add_argument("-dk", "--dontcheck", action="store_true", help="Skip the FASTQ check")
    p.add_argument("-ex", "--extension", type=str, default="n12", help='Output extension. Default "n12"')
    p.add_argument("-or", "--orientation", type=str, default="reverse", help="forward/reverse/both")
    p.add_argument("-tg", "--tags", type=str, default="extended", help="Tag set: extended or original")
    p.add_argument("-sp", "--species", type=str, default="human", help="human or mouse")
    p.add_argument("-N", "--allowNs", action="store_true", help="Allow rearrangements containing N")
    p.add_argument("-ln", "--lenthreshold", type=int, default=130, help="Inter-tag length threshold")
    p.add_argument("-tfdir", "--tagfastadir", type=str, default="Decombinator-Tags-FASTAs",
                   help="Folder with the tag and FASTA files")
    p.add_argument("-nbc", "--nobarcoding", action="store_true", help="Run without barcoding")
    p.add_argument("-bl", "--bclength", type=int, default=42, help="Barcode length. Default 42")


def _later_stage_flags(p: argparse.ArgumentParser):
    # collapse / translate flags of the reference (io.py:230-383): -ol, -mq, -bm, -aq steer the front half of collapse; the rest are parsed and carried
    p.add_argument("-mq", "--minbcQ", type=int, default=20)
    p.add_argument("-bm", "--bcQbelowmin", type=int, default=1)
    p.add_argument("-aq", "--avgQthreshold", type=int, default=30)
    p.add_argument("-lv", "--percentlevdist", type=int, default=10)
    p.add_argument("-bc", "--bcthreshold", type=int, default=2)
    p.add_argument("-di", "--dontcheckinput", action="store_true")
    p.add_argument("-bd", "--barcodeduplication", action="store_true")
    p.add_argument("-pb", "--positionalbarcodes", action="store_true")
    p.add_argument("-ol", "--oligo", type=str, default="M13")
    p.add_argument("-wc", "--writeclusters", action="store_true")
    p.add_argument("-uh", "--UMIhistogram", action="store_true")
    p.add_argument("-npf", "--nonproductivefilter", action="store_true")


def create_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(
        prog="decombinator",
        description="Decombinator `decombine` stage on MI355X (HIP), the per-row front half of `collapse` and `translate` (CDR3 per DCR).  "
                    "Sub-commands as in the reference; the grouping / clustering half of `collapse` is not part of this build.")
    parser.add_argument("-v", "--version", action="version", version=__version__)
    sub = parser.add_subparsers(dest="command", help="Available commands")
    sub.required = False
    pipe = sub.add_parser("pipeline", help="decombine, then the front half of collapse (its grouping half: not in this build)")
    _common(pipe); _decombine(pipe); _later_stage_flags(pipe)
    dec = sub.add_parser("decombine", help="Decombine TCR reads")
    _common(dec); _decombine(dec)
    col = sub.add_parser("collapse", help="front half of collapse over an .n12 file: barcode extraction and the row filters")
    _common(col); _later_stage_flags(col)
    col.add_argument("-in", "--infile", type=str, required=True, help=".n12 file of the decombine stage (optionally gzipped)")
    col.add_argument("-N", "--allowNs", action="store_true", help="Allow barcodes containing N")
    col.add_argument("-ln", "--lenthreshold", type=int, default=130, help="Inter-tag length threshold")
    tr = sub.add_parser("translate", help="CDR3 extraction (translate.get_cdr3) for the DCRs of a .freq file; writes the AIRR .tsv")
    _common(tr); _later_stage_flags(tr)
    tr.add_argument("-in", "--infile", type=str, required=True, help=".freq file (v,j,vdel,jdel,insert,frequency,cluster size per line)")
    tr.add_argument("-tg", "--tags", type=str, default="extended")
    tr.add_argument("-sp", "--species", type=str, default="human")
    tr.add_argument("-tfdir", "--tagfastadir", type=str, default="Decombinator-Tags-FASTAs")
    tr.add_argument("-nbc", "--nobarcoding", action="store_true")
    return parser


def cli_args(argv=None) -> dict:
    return vars(create_parser().parse_args(argv))


class _GzText:
    """What write_text / write need of a text file object, in front of a GzipWriter: str is buffered and encoded as the
    reference's text-mode file does (UTF-8, "\n" kept), bytes go through `buffer` as they are."""

    def __init__(self, gz):
        self._gz, self._pending, self._n = gz, [], 0
        self.buffer = self

    def write(self, x):
        if isinstance(x, str):
            x = x.encode("utf-8")
            self._pending.append(x)
            self._n += len(x)
            if self._n >= (8 << 20):
                self.flush()
        else:
            self.flush()
            self._gz.write(x)

    def flush(self):
        if self._pending:
            self._gz.write(b"".join(self._pending))
            self._pending, self._n = [], 0


def write_out_intermediate(data: list, inputargs: dict, suffix: str):
    """`<outpath><prefix><file id>_<chain name><suffix>`: one row per line, fields joined by
    ", "; gzipped unless dontgzip; mode 666 (reference io.py:480-513)."""
    chainnams = {"a": "alpha", "b": "beta", "g": "gamma", "d": "delta"}
    filename_id = os.path.basename(inputargs["infile"]).split(".")[0]
    if inputargs["command"] in ["collapse", "translate"]:
        outfilename = inputargs["outpath"] + f"{filename_id}" + suffix
    else:
        outfilename = (inputargs["outpath"] + inputargs["prefix"] + f"{filename_id}"
                       + f"_{chainnams[inputargs['chain'].lower()]}" + suffix)
    if not inputargs["dontgzip"]:
        # the reference writes the text, re-reads it through gzip.open (one thread, level 9) and unlinks it; what is left
        # is the .gz, written here at once by libdcrx's threaded gzip writer (same decompressed bytes)
        from . import _native as nat
        print("Compressing intermediate output file to", outfilename + ".gz")
        with nat.GzipWriter(outfilename + ".gz", level=int(os.environ.get("DCRX_GZIP_LEVEL", "6"))) as gz:
            out = _GzText(gz)
            if hasattr(data, "write_text"):      # decombine.N12Rows: the rows are text already
                data.write_text(out, ", ")
            else:
                for line in data:
                    out.write(", ".join(map(str, line)) + "\n")
            out.flush()
        outfilename += ".gz"
    else:
        with open(outfilename, "w") as outfile:
            if hasattr(data, "write_text"):
                data.write_text(outfile, ", ")
            else:
                for line in data:
                    outfile.write(", ".join(map(str, line)) + "\n")
    sort_permissions(outfilename)
    return outfilename


def write_out_translated(rows, headers, inputargs: dict):
    """The AIRR table of the translate stage (reference io.py:516-548: `DataFrame.to_csv(sep="\t", index=False)`, then the
    gzip step unless dontgzip, then mode 666): `<outpath><file id>.tsv[.gz]` for the translate / collapse commands, the
    pipeline's `<prefix><file id>_<chain name>.tsv[.gz]` otherwise.  A missing value (None) is an empty field, as to_csv
    writes it."""
    chainnams = {"a": "alpha", "b": "beta", "g": "gamma", "d": "delta"}
    filename_id = os.path.basename(inputargs["infile"]).split(".")[0]
    if inputargs["command"] in ["collapse", "translate"]:
        outfilename = inputargs["outpath"] + f"{filename_id}" + ".tsv"
    else:
        outfilename = (inputargs["outpath"] + inputargs["prefix"] + f"{filename_id}"
                       + f"_{chainnams[inputargs['chain'].lower()]}" + ".tsv")

    def lines():
        yield "\t".join(headers) + "\n"
        for r in rows:
            yield "\t".join("" if x is None else str(x) for x in r) + "\n"

    if not inputargs["dontgzip"]:
        from . import _native as nat
        print("Compressing pipeline output file to", outfilename + ".gz")
        with nat.GzipWriter(outfilename + ".gz", level=int(os.environ.get("DCRX_GZIP_LEVEL", "6"))) as gz:
            out = _GzText(gz)
            for ln in lines():
                out.write(ln)
            out.flush()
        outfilename += ".gz"
    else:
        with open(outfilename, "w") as fh:
            fh.writelines(lines())
    sort_permissions(outfilename)
    return outfilename
