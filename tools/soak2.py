"""Second soak run on the GPU box: 65 M more reads against the oracle over the knobs tools/soak.py leaves fixed — allowNs
on and off, inter-tag length thresholds 130 / 60 / 20, orientations reverse, forward and both (on either strand), reads with
1-5 % exception bytes.  usage: python tools/soak2.py   (a little over two minutes on one MI355X box)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from decombinator_amd import synth, _native as nat
from oracle import oracle as orc
from tests import parity_util as pu
t0 = time.time(); tot = 0
for name, ts in (("beta", synth.config_tagset(2)), ("alphaX", synth.config3_tagsets()[0]), ("delta", synth.config5_tagsets()[1])):
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps, [r.upper() for r in ts.j_regions], vs, js)
    for k, (sub, nrate, length) in enumerate([(0.01, 0.02, 150), (0.04, 0.05, 121), (0.02, 0.01, 200)]):
        hb = nat.synth_reads_host(t, nat.synth_cfg(seed=700 + k, sub_rate=sub, n_rate=nrate, read_len=length), 0, 300_000)
        reads = nat.unpack_reads(hb)
        fwd = [orc.revcomp(r) for r in reads]
        hbf = nat.pack_reads(fwd)
        for allow_ns in (False, True):
            for lenthr in (130, 60, 20):
                for orient, batch, rr in (("reverse", hb, reads), ("forward", hbf, fwd), ("both", hbf, fwd), ("both", hb, reads)):
                    rec, cnt = nat.decombine(t, batch, orient, allow_ns, lenthr)
                    orec, ocnt = pu.oracle_records(ot, rr, orient, allow_ns, lenthr)
                    pu.assert_records_equal(rec, orec, rr, f"{name} {orient} allowNs={allow_ns} len={lenthr}")
                    pu.assert_counters_equal(cnt, ocnt, f"{name} {orient}")
                    tot += len(rr)
    print("SOAK2", name, "ok", tot, f"{time.time() - t0:.0f}s", flush=True)
print("SOAK2_DONE", tot)
