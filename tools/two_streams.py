"""Experiment: the 10 M-read step as two halves on two handles and two streams (the scan of one half beside the finishing
kernels of the other), against one handle; scan blocks of fewer threads leave room for the other half's kernels."""
import os as _os, sys as _sys, time
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import torch
import numpy as np
from decombinator_amd import _native as nat, synth
ts = synth.config_tagset(2)
mk = lambda: nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = 10_000_000
parts = int(_os.environ.get("PARTS", "2"))
tabs = [mk() for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
cfgs = nat.synth_cfg(seed=2)
dbs, recs, cnts = [], [], []
for k in range(parts):
    lo, hi = n * k // parts, n * (k + 1) // parts
    dbs.append(nat.synth_reads_device(tabs[k], cfgs, lo, hi - lo))
    recs.append(nat.DeviceBuffer((hi - lo) * 16)); cnts.append(nat.DeviceBuffer(nat.N_COUNTERS * 8))
def step():
    for k in range(parts):
        nat.decombine_device(tabs[k], dbs[k], recs[k], cnts[k], stream=streams[k].cuda_stream)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
hits = sum(int(c.to_host(np.uint64, nat.N_COUNTERS)[19]) for c in cnts)
print(f"PARTS {parts} scan_threads {_os.environ.get('DCRX_SCAN_THREADS', '1024')}: {dt * 1e3:.4f} ms per 10 M reads, decombined {hits}")
