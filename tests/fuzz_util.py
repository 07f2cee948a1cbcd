"""Randomised differential runs: random tag sets (tag length, half splits, family structure),
read lengths, error rates, ragged cuts, orientation and filter settings — one backend ("emul" on
CPU, "hip" on the GPU) against the oracle, bit-exact."""
import numpy as np

from decombinator_amd import _native as nat, synth
from oracle import oracle as orc
from tests import parity_util as pu


def run(kind: str, n_configs: int, n_reads: int, seed: int) -> int:
    rng = np.random.default_rng(seed)
    total_ok = 0
    for c in range(n_configs):
        tagsname = ["original", "extended"][int(rng.integers(0, 2))]
        tag_len = int(rng.integers(14, 25))
        n_v, n_j = int(rng.integers(5, 71)), int(rng.integers(3, 16))
        ts = synth.make_tagset("human", tagsname, "b", n_v=n_v, n_j=n_j, seed=int(rng.integers(1, 1 << 30)), tag_len=tag_len,
                               n_shared_groups=int(rng.integers(0, 5)))
        d = dict(v_tags=ts.v_tags, v_jumps=ts.v_jumps, v_regions=ts.v_regions, j_tags=ts.j_tags, j_jumps=ts.j_jumps,
                 j_regions=ts.j_regions, v_half_split=ts.half_splits[0], j_half_split=ts.half_splits[1])
        t = pu.native_tables(d)
        ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                              [r.upper() for r in ts.j_regions], *ts.half_splits)
        be = pu.Backend(kind, d)
        read_len = int(rng.integers(60, 301))
        stride = nat.stride_for(read_len)
        cfg = nat.synth_cfg(seed=int(rng.integers(1, 1 << 30)), read_len=read_len, p_rearranged=float(rng.uniform(0.2, 0.9)),
                            sub_rate=float(rng.choice([0.0, 0.005, 0.02, 0.05])), n_rate=float(rng.choice([0.0, 0.001, 0.02])))
        hb = nat.synth_reads_host(t, cfg, 0, n_reads, stride=stride)
        reads = nat.unpack_reads(hb)
        mode = int(rng.integers(0, 3))
        if mode == 1:                                     # ragged: random cuts from either end
            cut = rng.integers(0, read_len + 1, size=len(reads))
            reads = [r[:k] if i % 2 else r[len(r) - k:] for i, (r, k) in enumerate(zip(reads, cut))]
        elif mode == 2:                                   # another uniform length (odd ones included)
            k = int(rng.integers(max(1, read_len - 40), read_len + 1))
            reads = [r[:k] for r in reads]
        orientation = ["reverse", "forward", "both"][int(rng.integers(0, 3))]
        if orientation != "reverse":
            reads = [orc.revcomp(r) if i % 2 else r for i, r in enumerate(reads)]
        allow_ns = bool(rng.integers(0, 2))
        lenthreshold = int(rng.choice([130, 60, 20]))
        flags = int(rng.choice([0, 0, 0, nat.F_LIST_RESCUE, nat.F_ONE_BASE_SCAN]))
        b = nat.pack_reads(reads, stride=stride)
        rec, cnt = be.run(b, orientation, allow_ns, lenthreshold, flags=flags)
        orec, ocnt = pu.oracle_records(ot, reads, orientation, allow_ns, lenthreshold)
        label = f"config {c}: tags={tagsname} tag_len={tag_len} nv={n_v} nj={n_j} len={read_len} mode={mode} {orientation} flags={flags}"
        pu.assert_records_equal(rec, orec, reads, label)
        pu.assert_counters_equal(cnt, ocnt, label)
        total_ok += int((orec["status"] == 0).sum())
    return total_ok
