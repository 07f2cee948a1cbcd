"""PCIe-inclusive rate of the host-buffer entry point (dcrx_decombine: H2D + kernels + D2H),
for DESIGN.md; never the benchmark's `value`."""
import time
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root
import numpy as np
from decombinator_amd import _native as nat, synth

ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = 10_000_000
hb = nat.synth_reads_host(t, nat.synth_cfg(seed=2), 0, n)
nat.decombine(t, hb)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); rec, cnt = nat.decombine(t, hb); best = min(best, time.perf_counter() - t0)
print(f"PCIE_INCLUSIVE pageable reads={n} seconds={best:.4f} Mreads/s={n / best / 1e6:.1f} hits={int(cnt[19])}")
# the same from buffers the caller has pinned (dcrx_malloc_host): no staging copies
hp = nat.synth_reads_host(t, nat.synth_cfg(seed=2), 0, n, pinned=True)
out = nat.pinned_empty(n, nat.RECORD_DTYPE)
nat.decombine(t, hp, out=out)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); rec2, cnt2 = nat.decombine(t, hp, out=out); best = min(best, time.perf_counter() - t0)
assert (cnt2 == cnt).all() and (rec2 == rec).all()
print(f"PCIE_INCLUSIVE pinned reads={n} seconds={best:.4f} Mreads/s={n / best / 1e6:.1f} hits={int(cnt2[19])}")
