"""usage: in_flight_worker.py N_READS N_BATCHES — N_BATCHES batches of N_READS reads alternate between two handles of config 2's tag set,
each handle on a stream of its own, nothing waited for until all are issued; every record and counter of every batch against the
oracle.  Prints IN_FLIGHT_OK and the handles' last launch forms.  (A process of its own so that the caller can force a launch
form through the library's debug environment: tests/test_gpu_parity.py.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decombinator_amd import _native as nat, synth      # noqa: E402
from oracle import oracle as orc      # noqa: E402
from tests import parity_util as pu      # noqa: E402

n, n_batches = int(sys.argv[1]), int(sys.argv[2])
ts = synth.config_tagset(2)


def tables():
    return nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)


ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps, [r.upper() for r in ts.j_regions],
                      *ts.half_splits)
handles, streams = (tables(), tables()), (nat.Stream(), nat.Stream())
cfg = nat.synth_cfg(seed=66, sub_rate=0.01)
batches = [nat.synth_reads_device(handles[0], cfg, k * n, n) for k in range(n_batches)]
recs = [nat.DeviceBuffer(n * 16) for _ in range(n_batches)]
cnts = [nat.DeviceBuffer(nat.N_COUNTERS * 8) for _ in range(n_batches)]
nat.synchronize()
for k in range(n_batches):
    nat.decombine_device(handles[k % 2], batches[k], recs[k], cnts[k], stream=streams[k % 2].ptr)
nat.synchronize()
forms = [h.tune_state(n)["launch_form"] for h in handles]
assert all(f.startswith("v2, tail") for f in forms), forms
for k in range(n_batches):
    hb = nat.synth_reads_host(handles[0], cfg, k * n, n)
    buf, offsets = nat.unpack_reads_raw(hb)
    ores, ocnt = ot.decombine_batch_mt(buf, offsets)
    orec = pu.oracle_to_records(ores)
    rec = recs[k].to_host(nat.RECORD_DTYPE, n)
    if rec.tobytes() != orec.tobytes():
        pu.assert_records_equal(rec, orec, nat.unpack_reads(hb), f"batch {k}")
    pu.assert_counters_equal(cnts[k].to_host(np.uint64, nat.N_COUNTERS), ocnt, f"batch {k}")
print("IN_FLIGHT_OK", forms)
