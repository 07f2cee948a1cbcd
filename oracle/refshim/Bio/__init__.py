"""Stand-in for biopython==1.84 (reference pyproject.toml:11).  TEST
INFRASTRUCTURE for oracle/gen_golden.py only: Seq.reverse_complement / upper /
str / len / slicing and SeqIO.parse(path, "fasta")."""


class BiopythonWarning(Warning):
    pass
