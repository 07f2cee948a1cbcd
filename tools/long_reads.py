"""Throughput of the hot path on reads longer than the bench's 150 nt (device-resident, like bench.py): the 250 and
300 nt of 2x250 / 2x300 libraries take the kernels' second register shape (20 words per read), 321-511 nt the third
(32 words, one read per lane); 512 nt and more the long form (one read per lane from memory, a two-pass scan, plain-integer positions: fewer
reads per launch here).  usage (GPU box): python tools/long_reads.py [read_len ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from decombinator_amd import _native as nat, synth

ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
FLAGS = int(os.environ.get("DCRX_LONG_READS_FLAGS", "0"))      # (profiling aids of the kernels: nat.F_PROFILE_*)
for L in [int(x) for x in sys.argv[1:]] or [150, 250, 300, 400, 500, 600, 2000]:
    n = 4_000_000 if L <= 511 else (2_000_000 if L <= 1000 else 500_000)      # (the long form: every lane of the chip a read, and a few rounds of them)
    db = nat.synth_reads_device(t, nat.synth_cfg(seed=2, read_len=L), 0, n)
    d_rec = nat.DeviceBuffer(n * 16)
    d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)
    for _ in range(3):
        nat.decombine_device(t, db, d_rec, d_cnt, flags=FLAGS)
    nat.synchronize()
    t0 = time.perf_counter()
    k = 10
    for _ in range(k):
        nat.decombine_device(t, db, d_rec, d_cnt, flags=FLAGS)
    nat.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"LONG read_len={L} reads={n} ms_per_step={dt * 1e3:.3f} Mreads/s={n / dt / 1e6:.0f} Gbases/s={n * L / dt / 1e9:.0f}")
