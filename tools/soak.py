"""Soak run on the GPU box: the device path against the oracle on 66 M synthetic reads — five tag sets x eleven shapes of
workload (rescue-heavy, exception-heavy, odd, short, long, 31 and 33 nt) x both strands x the default and the in-series launch
order; records and counters must match for every batch (tests/test_emul_parity.py::_synthetic_vs_oracle raises otherwise).
usage: python tools/soak.py   (about three minutes on one MI355X box; the oracle is the slow side)"""
import sys, time
sys.path.insert(0, ".")
from decombinator_amd import synth, _native as nat
from tests import test_emul_parity as tep
t0 = time.time(); total = 0
for name, ts in (("beta", synth.config_tagset(2)), ("alphaX", synth.config3_tagsets()[0]), ("betaX", synth.config3_tagsets()[1]), ("gamma", synth.config5_tagsets()[0]), ("delta", synth.config5_tagsets()[1])):
    for k, (sub, nrate, length) in enumerate(tep._LEAN_CASES + [(0.005, 0.0005, 150), (0.02, 0.001, 250), (0.01, 0.0, 320), (0.02, 0.01, 31), (0.02, 0.0, 33)]):
        for flags in (0, nat.F_V2_LEAN_SERIAL):
            total += tep._synthetic_vs_oracle("hip", ts, 300_000, "reverse", flags, seed=900 + k, sub_rate=sub, n_rate=nrate, read_len=length)
            total += tep._synthetic_vs_oracle("hip", ts, 300_000, "forward", flags, forward_strand=True, seed=950 + k, sub_rate=sub, n_rate=nrate, read_len=length)
    print("SOAK", name, "ok, decombined so far", total, f"{time.time() - t0:.0f}s", flush=True)
print("SOAK_DONE", total)
