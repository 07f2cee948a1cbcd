#!/usr/bin/env python3
"""Times the reference's OWN Python read loop body — dcr(revcomp(read)) per read, decombine.py:998-1001 — in the build
container on reads of the benchmark workload.  Container-only (needs /root/reference); the
reference is imported unmodified, with oracle/refshim standing in for the three wheels that are absent offline
(acora, Bio, Levenshtein: the matcher timed here is a pure-Python stand-in, not Cython acora).  One core: the
reference has no parallelism.  usage: tools/ref_python_loop.py [n_reads]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decombinator_amd import _native as nat, synth  # noqa: E402
from oracle import ref_driver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ts = synth.config_tagset(2)
d = tempfile.mkdtemp()
ts.write(d)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
reads = nat.unpack_reads(nat.synth_reads_host(t, nat.synth_cfg(seed=2, read_len=150), 0, n))
rc = ref_driver.RefChain(d, "human", "original", "b")
m = rc.m
t0 = time.perf_counter()
ok = 0
for r in reads:
    m.counts["read_count"] += 1
    if m.dcr(m.revcomp(r), rc.args):
        ok += 1
dt = time.perf_counter() - t0
print(f"reference decombine.dcr(revcomp(read)) loop: {n} reads of the config-2 workload in {dt:.2f} s = {n / dt:.0f} reads/s "
      f"on one core ({ok} decombined); stand-in matcher for acora")
