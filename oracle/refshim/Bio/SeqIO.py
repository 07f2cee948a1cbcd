"""Bio.SeqIO stand-in: parse(handle_or_path, "fasta") yielding records with
.id (header up to the first whitespace), .description (whole header) and .seq."""
from .Seq import Seq


class SeqRecord:
    def __init__(self, seq, id="", description=""):
        self.seq = seq
        self.id = id
        self.name = id
        self.description = description


def parse(handle, fmt):
    if fmt != "fasta":
        raise ValueError("stand-in supports fasta only")
    close = False
    if isinstance(handle, (str, bytes)) or hasattr(handle, "__fspath__"):
        handle = open(handle, "rt")
        close = True
    try:
        header, chunks = None, []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if header is not None:
                    yield SeqRecord(Seq("".join(chunks)), header.split()[0] if header.split() else "", header)
                header, chunks = line[1:], []
            elif header is not None:
                chunks.append(line.strip())
        if header is not None:
            yield SeqRecord(Seq("".join(chunks)), header.split()[0] if header.split() else "", header)
    finally:
        if close:
            handle.close()
