"""Stand-in package for pyrepseq (not installable offline): the reference's collapse.py imports
pyrepseq.nn at module level; the fixtures generated here never reach the functions that use it."""
