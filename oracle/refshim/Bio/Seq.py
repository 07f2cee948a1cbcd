"""Bio.Seq stand-in: the ambiguous-DNA complement table of Biopython's
Bio.Data.IUPACData (both cases, U complemented like T); other characters are
left alone, as bytes.translate does."""

_K = "ACGTMRWSYKVHDBXNU"
_V = "TGCAKYWSRMBDHVXNA"
_TABLE = str.maketrans(_K + _K.lower(), _V + _V.lower())


class Seq:
    def __init__(self, data=""):
        self._data = str(data)

    def __str__(self):
        return self._data

    def __repr__(self):
        return "Seq(%r)" % self._data

    def __len__(self):
        return len(self._data)

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(self._data)

    def __getitem__(self, idx):
        r = self._data[idx]
        return Seq(r) if isinstance(idx, slice) else r

    def __add__(self, other):
        return Seq(self._data + str(other))

    def upper(self):
        return Seq(self._data.upper())

    def lower(self):
        return Seq(self._data.lower())

    def complement(self):
        return Seq(self._data.translate(_TABLE))

    def reverse_complement(self):
        return Seq(self._data.translate(_TABLE)[::-1])

    def translate(self):
        """Standard table (NCBI 1), as Bio.Seq.translate() without arguments (biopython 1.84, Bio.Seq._translate_str over
        Bio.Data.CodonTable's ambiguous standard DNA table): '*' for stop codons; a codon that holds IUPAC ambiguity codes
        gives the residue all the codons it stands for share, 'B' / 'Z' / 'J' when those give exactly {D, N} / {E, Q} / {I, L},
        '*' when every one of them is a stop, and 'X' otherwise (several residues, or stops beside residues); a letter that
        is no nucleotide code raises (CodonTable.TranslationError is a ValueError); a trailing partial codon is dropped
        (Biopython warns and does the same)."""
        s = self._data.upper().replace("U", "T")
        return Seq("".join(_codon_residue(s[i:i + 3]) for i in range(0, len(s) - len(s) % 3, 3)))


_BASES = "TCAG"
_RESIDUES = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_STANDARD = {a + b + c: _RESIDUES[16 * i + 4 * j + k] for i, a in enumerate(_BASES) for j, b in enumerate(_BASES) for k, c in enumerate(_BASES)}
# Bio.Data.IUPACData.ambiguous_dna_values
_MEANS = {"A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "R": "AG", "W": "AT", "S": "CG", "Y": "CT", "K": "GT", "V": "ACG", "H": "ACT",
          "D": "AGT", "B": "CGT", "X": "GATC", "N": "GATC"}
# the two-residue letters of Bio.Data.IUPACData.extended_protein_values
_PAIRS = {frozenset("DN"): "B", frozenset("EQ"): "Z", frozenset("IL"): "J"}
_seen = {}


def _codon_residue(codon):
    got = _STANDARD.get(codon)
    if got is not None:
        return got
    if codon in _seen:
        return _seen[codon]
    for ch in codon:
        if ch not in _MEANS:
            raise ValueError("Codon '%s' is invalid" % codon)
    stops = residues = 0
    kinds = set()
    for x in _MEANS[codon[0]]:
        for y in _MEANS[codon[1]]:
            for z in _MEANS[codon[2]]:
                r = _STANDARD[x + y + z]
                if r == "*":
                    stops += 1
                else:
                    residues += 1
                    kinds.add(r)
    if residues == 0:
        out = "*"
    elif stops:
        out = "X"
    elif len(kinds) == 1:
        out = kinds.pop()
    else:
        out = _PAIRS.get(frozenset(kinds), "X")
    _seen[codon] = out
    return out
