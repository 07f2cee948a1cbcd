"""Worker of tests/test_gpu_parity.py::test_timed_launch_shapes_at_size: one process per launch shape, because the library
reads its A/B switches (DCRX_DEBUG_RESCUE_WAVES, DCRX_DEBUG_TAIL_WAVES, DCRX_DEBUG_NO_TUNE) once per process.  Decombines
N device-resident reads of a BASELINE config in one launch (the fused form at a batch size the handle tunes itself on:
>= 2^20 reads), CALLS times on one handle — the handle's own choice of rescue waves settles on the sixth call — and compares
every record and every counter of every call with the threaded oracle.  usage: forced_shape_worker.py CONFIG N CALLS"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decombinator_amd import _native as nat          # noqa: E402
from decombinator_amd import synth                   # noqa: E402
from oracle import oracle as orc                     # noqa: E402
from tests import parity_util as pu                  # noqa: E402


def main():
    config, n, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ts = synth.config_tagset(2) if config == 2 else list(synth.config5_tagsets())[0]
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    cfg = nat.synth_cfg(seed=config, sub_rate=0.02 if config == 5 else 0.005)
    hb = nat.synth_reads_host(t, cfg, 0, n)
    buf, offsets = nat.unpack_reads_raw(hb)
    ores, ocnt = ot.decombine_batch_mt(buf, offsets)
    orec = pu.oracle_to_records(ores)
    db = nat.synth_reads_device(t, cfg, 0, n)
    d_rec = nat.DeviceBuffer(n * 16)
    d_cnt = nat.DeviceBuffer(nat.N_COUNTERS * 8)
    for k in range(calls):
        nat.check(nat.lib().dcrx_memset_device(d_rec.ptr, 0xEE, n * 16))      # (a record no kernel writes would show)
        nat.decombine_device(t, db, d_rec, d_cnt)
        nat.synchronize()
        rec = d_rec.to_host(nat.RECORD_DTYPE, n)
        cnt = d_cnt.to_host(np.uint64, nat.N_COUNTERS)
        if rec.tobytes() != orec.tobytes():
            pu.assert_records_equal(rec, orec, nat.unpack_reads(hb), f"call {k}")
        pu.assert_counters_equal(cnt, ocnt, f"call {k}")
    st = t.tune_state(n)
    print("SHAPE_OK", config, n, calls, st)


if __name__ == "__main__":
    main()
