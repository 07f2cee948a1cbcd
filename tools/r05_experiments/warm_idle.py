"""Is the slow start of a run the device's clocks or a one-time warm-up of the library?  Bursts of 60 steps (config 2, 10 M reads)
with the device left idle for 0 / 0.05 / 0.5 / 3 s between them: the wall time of the first 10, the next 20 and the last 30 steps
of every burst.  usage (GPU box): python tools/r05_experiments/warm_idle.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from decombinator_amd import _native as nat, synth

ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = 10_000_000
db = nat.synth_reads_device(t, nat.synth_cfg(seed=2), 0, n)
d_rec = nat.DeviceBuffer(n * 16)
d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)


def burst(k):
    nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        nat.decombine_device(t, db, d_rec, d_cnt)
    nat.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for idle in (0.0, 0.0, 0.05, 0.5, 3.0, 3.0):
    time.sleep(idle)
    a, b, c = burst(10), burst(20), burst(30)
    print(f"WARM idle {idle:4.2f} s before: steps 1-10 {a:.4f} ms, 11-30 {b:.4f}, 31-60 {c:.4f}")
