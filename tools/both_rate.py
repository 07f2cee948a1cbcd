"""Rate of orientation `both` (two v2 passes) against `reverse` on the bench workload, device-resident (for DESIGN.md)."""
import os as _os, sys as _sys, time
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np
from decombinator_amd import _native as nat, synth
ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = 10_000_000
db = nat.synth_reads_device(t, nat.synth_cfg(seed=2), 0, n)
d_rec, d_cnt = nat.DeviceBuffer(n * 16), nat.DeviceBuffer(nat.N_COUNTERS * 8)
for o in ("reverse", "both", "forward"):
    for _ in range(3):
        nat.decombine_device(t, db, d_rec, d_cnt, orientation=o)
    nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        nat.decombine_device(t, db, d_rec, d_cnt, orientation=o)
    nat.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"ORIENTATION {o}: {dt * 1e3:.3f} ms per 10 M reads = {n / dt / 1e9:.2f} G reads/s, decombined {int(d_cnt.to_host(np.uint64, nat.N_COUNTERS)[19])}")
