"""Stage sequencing (reference src/decombinator/pipeline.py:10-58).  This build implements
stage 1 (`decombine`); `collapse` (UMI error correction) and `translate` (CDR3 extraction)
are host stages of the reference that BASELINE.json's north_star leaves on the host and
SURVEY.md §8(f) lists as later rows: `pipeline` therefore stops after writing the `.n12`."""
from __future__ import annotations

from datetime import datetime
from typing import Any, Optional

from .decombine import decombinator
from .io import cli_args, write_out_intermediate


def run(args: Optional[dict[str, Any]] = None, cli_args: Optional[dict[str, Any]] = None):
    start = datetime.now()
    inp = cli_args if cli_args else args
    data = decombinator(inp)
    if not inp["dontsave"]:
        write_out_intermediate(data, inp, ".n12")
    print("Decombinator complete...")
    print("collapse / translate are not part of this build: feed the .n12 to the reference's "
          "`decombinator collapse` and `decombinator translate`.")
    print(f"Pipeline complete in {datetime.now() - start}")
    return data


def main(argv=None):
    inp = cli_args(argv)
    if inp["command"] == "decombine":
        data = decombinator(inp)
        write_out_intermediate(data, inp, ".n12")
    elif inp["command"] == "pipeline":
        run(cli_args=inp)
    else:
        from .io import create_parser
        create_parser().print_help()


if __name__ == "__main__":
    main()
