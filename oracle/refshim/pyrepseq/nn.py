def symdel(*a, **k):
    raise NotImplementedError("pyrepseq is not available offline; the row front half of collapse does not use it")
