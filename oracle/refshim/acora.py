"""Stand-in for acora==2.4 (reference pyproject.toml:10), which is not
installable in the build container.  TEST INFRASTRUCTURE: used only by
oracle/gen_golden.py to import the reference's decombine.py unmodified.

Contract restated from acora's published behaviour (AcoraBuilder.add/.build,
matcher.findall), as relied upon at reference decombine.py:722-746 and
:275,:294,:339,:399,:422,:473:
  * keywords are kept as a set (duplicates collapse);
  * findall(text) returns [(keyword, start)] for every occurrence, overlaps
    included, in order of END position; for equal end, longest first;
  * a character that occurs in no keyword resets the machine.
Deliberately written differently from oracle/dcr_oracle.c (dict trie + explicit
failure walk, no dense table) so that the two cross-check each other.
"""


class _Matcher:
    def __init__(self, keywords):
        self._kw = sorted(set(k for k in keywords if k))
        self._children = [{}]
        self._word = [None]
        for w in self._kw:
            s = 0
            for ch in w:
                nxt = self._children[s].get(ch)
                if nxt is None:
                    nxt = len(self._children)
                    self._children.append({})
                    self._word.append(None)
                    self._children[s][ch] = nxt
                s = nxt
            self._word[s] = w
        n = len(self._children)
        self._fail = [0] * n
        order = list(self._children[0].values())
        i = 0
        while i < len(order):
            s = order[i]
            i += 1
            for ch, t in self._children[s].items():
                f = self._fail[s]
                while f and ch not in self._children[f]:
                    f = self._fail[f]
                cand = self._children[f].get(ch, 0)
                self._fail[t] = cand if cand != t else 0
                order.append(t)

    def findall(self, text):
        out = []
        s = 0
        for i, ch in enumerate(text):
            while s and ch not in self._children[s]:
                s = self._fail[s]
            s = self._children[s].get(ch, 0)
            t = s
            while t:
                w = self._word[t]
                if w is not None:
                    out.append((w, i + 1 - len(w)))
                t = self._fail[t]
        return out

    def finditer(self, text):
        return iter(self.findall(text))


class AcoraBuilder:
    def __init__(self, *keywords, **kwargs):
        self.keywords = set()
        self.add(*keywords)

    def add(self, *keywords):
        self.keywords.update(keywords)

    def build(self, ignore_case=None, acora=None):
        return _Matcher(self.keywords)
