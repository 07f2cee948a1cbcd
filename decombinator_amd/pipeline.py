"""Stage sequencing (reference src/decombinator/pipeline.py:10-58).  This build implements stage 1 (`decombine`, on the
GPU) and the per-row front half of stage 2 (`collapse`: spacer search, UMI extraction, barcode quality and length filters —
collapse.read_in_data up to where it starts grouping rows, collapse.py:482-565).  The grouping / clustering rest of
`collapse` and the `translate` stage are host stages of the reference that BASELINE.json's north_star leaves on the host
(`translate.get_cdr3` is available as a function: decombinator_amd/translate.py).  `pipeline` therefore writes the `.n12`,
runs the front half over the rows and reports its counters."""
from __future__ import annotations

from datetime import datetime
from typing import Any, Optional

from . import collapse
from .decombine import decombinator
from .io import cli_args, write_out_intermediate


def collapse_front(data, inp):
    """The rows' front half of `collapse` with the stage's own flags (io.py: -ol, -mq, -bm, -aq, -ln, -N).  Returns the
    FrontRows; prints the reference's counters."""
    collapse.counts.clear()
    params = [inp.get("minbcQ", 20), inp.get("bcQbelowmin", 1), inp.get("avgQthreshold", 30)]       # barcode_quality_parameters (collapse.py:917-921)
    front = collapse.read_in_data(data, inp, params, inp.get("percentlevdist", 10) / 100.0, True, _opener(data) if inp["command"] == "collapse" else None)
    kept = len(front.kept())
    print(f"Collapse front half: {len(front):,} rows in, {kept:,} with a barcode of sufficient quality and an inter-tag sequence within the length threshold")
    for k in sorted(collapse.counts):
        print(f"\t{k},{collapse.counts[k]}")
    return front


def _opener(path):
    import gzip
    return gzip.open if str(path).endswith(".gz") else open


def run(args: Optional[dict[str, Any]] = None, cli_args: Optional[dict[str, Any]] = None):
    start = datetime.now()
    inp = cli_args if cli_args else args
    data = decombinator(inp)
    if not inp["dontsave"]:
        write_out_intermediate(data, inp, ".n12")
    print("Decombinator complete...")
    if len(data) and inp.get("oligo") and not inp.get("nobarcoding"):
        collapse_front(data, inp)
    print("The grouping / clustering half of `collapse` is not part of this build: feed the .n12 to the reference's "
          "`decombinator collapse`; `decombinator translate` of this build takes its .freq.")
    print(f"Pipeline complete in {datetime.now() - start}")
    return data


def main(argv=None):
    inp = cli_args(argv)
    if inp["command"] == "decombine":
        data = decombinator(inp)
        write_out_intermediate(data, inp, ".n12")
    elif inp["command"] == "pipeline":
        run(cli_args=inp)
    elif inp["command"] == "collapse":
        # the front half over an `.n12` file: the rows that pass, each with its barcode and barcode quality appended
        # (this build's intermediate: the reference goes on to group them in memory)
        front = collapse_front(inp["infile"], inp)
        out = [list(front[k][2]) + [front[k][5], front[k][3], front[k][4], front[k][0], front[k][1]] for k in front.kept().tolist()]
        write_out_intermediate(out, inp, ".n12u")
    elif inp["command"] == "translate":
        # translate.py:388-560 for a `.freq` file: the AIRR `.tsv` (tab-separated, a header line, no index), gzipped unless
        # -dz, mode 666 — the reference's write_out_translated (io.py:516-548)
        from . import translate
        from .io import write_out_translated
        rows = translate.cdr3translator(inp)
        name = write_out_translated(rows, translate.out_headers, inp)
        print("Translated", translate.counts["line_count"], "DCRs,", translate.counts["prod_recomb"], "productive ->", name)
    else:
        from .io import create_parser
        create_parser().print_help()


if __name__ == "__main__":
    main()
