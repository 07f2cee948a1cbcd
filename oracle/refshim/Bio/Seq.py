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
        """Standard table (NCBI 1), as Bio.Seq.translate() without arguments: '*' for stop codons, 'X' for a codon
        with an ambiguous base that does not determine the residue, a trailing partial codon dropped (Biopython warns
        and does the same)."""
        s = self._data.upper().replace("U", "T")
        bases = "TCAG"
        aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
        table = {a + b + c: aas[16 * i + 4 * j + k] for i, a in enumerate(bases) for j, b in enumerate(bases) for k, c in enumerate(bases)}
        out = []
        for i in range(0, len(s) - len(s) % 3, 3):
            out.append(table.get(s[i:i + 3], "X"))
        return Seq("".join(out))
