import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from decombinator_amd import _native as nat, synth
ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = 10_000_000
hb = nat.synth_reads_host(t, nat.synth_cfg(seed=2, n_rate=0.002), 0, n)
ref, cref = nat.decombine(t, hb)
bad = 0
for k in range(30):
    rec, cnt = nat.decombine(t, hb)
    if rec.tobytes() != ref.tobytes() or (cnt != cref).any(): bad += 1
print("PCIE_REPEAT 30 runs, differing:", bad, "hits", int(cref[19]))
