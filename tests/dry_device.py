"""CPU stand-in for the GPU in tests/bench_dry.py (tests only): the rank program of the bench — sharding, the step
loop, the TupleGather protocol, the counters, the JSON line — runs over gloo with the oracle producing the records a
device would.  Nothing here is a measurement; the line is marked "dry_run": true and carries no rate."""
from __future__ import annotations

import numpy as np

from decombinator_amd import sharded
from oracle import oracle as orc
from tests import parity_util as pu


class DryDevice:
    def __init__(self, nat, all_tables, tagsets, cfg_synth, batches, n):
        self.nat, self.batches, self.n = nat, batches, n
        self.all_tables = all_tables
        self.codec = None
        self.records = []          # per batch: the records of the last chain (what the gather carries)
        self.counts = []           # per batch: per chain counters
        for first, cnt in batches:
            recs, cnts = None, []
            for k, (tb, ts) in enumerate(zip(all_tables, tagsets)):
                ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                                      [r.upper() for r in ts.j_regions], *ts.half_splits)
                reads = []
                for c, tbc in enumerate(all_tables):        # a batch is drawn in equal parts from each chain's germlines
                    lo, hi = cnt * c // len(all_tables), cnt * (c + 1) // len(all_tables)
                    if hi > lo:
                        reads += nat.unpack_reads(nat.synth_reads_host(tbc, cfg_synth, first + lo, hi - lo))
                rec, c64 = pu.oracle_records(ot, reads, "reverse", False, 130)
                recs = rec
                cnts.append(c64.astype(np.uint64))
            self.records.append(recs)
            self.counts.append(cnts)
        self.sum = [np.zeros(nat.N_COUNTERS, dtype=np.uint64) for _ in all_tables]
        self.last = None
        self.accumulate = len(batches) > 1

    def compact(self, slot, n_reads):          # stands in for dcrx_compact_hits_packed_device (same layout, made on the host)
        nat = self.nat
        rec = np.frombuffer(slot["rec"].np.tobytes(), dtype=nat.RECORD_DTYPE)[:n_reads]
        # (the bench gathers narrow tuples: TupleGather(tables=...); one message of bitmap | low words | high bytes)
        if self.codec is None:
            self.codec = nat.TupleCodec(self.all_tables[-1], 150)
        m = self.codec.pack(rec, n_slots=self.n)
        slot["bitmap"][:] = 0
        slot["msg"].np[:len(m)] = m
        slot["n"].np.view(np.int64)[0] = int((rec["status"] == 0).sum())

    def name(self):
        return "dry (oracle on the CPU)"

    def kernels_tag(self, info):
        return "dry"

    def kernels(self, info):
        return "none (dry run)"

    def make_events(self, steps, timed):
        return {k: None for k in timed}

    def step(self, k, gather, ev):
        b = k % len(self.batches)
        rec = self.records[b]
        if self.accumulate and b == 0:
            for s in self.sum:
                s[:] = 0
        for c, s in enumerate(self.sum):
            if self.accumulate:
                s += self.counts[b][c]
            else:
                s[:] = self.counts[b][c]
        self.last = b
        if gather is not None:
            gather.before_scan()
            buf = gather.records()
            raw = rec.view(np.uint8).reshape(-1)
            buf.np[:raw.size] = raw
            gather.step(len(rec))

    def event_times(self, events, timed):
        return [0.0], [0.0]

    def totals(self):
        nat = self.nat
        i, j = nat.COUNTER_NAMES.index("vj_count"), nat.COUNTER_NAMES.index("read_count")
        return sum(int(s[i]) for s in self.sum), min(int(s[j]) for s in self.sum)

    def expected_read_count(self):
        return sum(b[1] for b in self.batches) if self.accumulate else self.n

    def last_step_hits(self):
        nat = self.nat
        return int(self.counts[self.last][-1][nat.COUNTER_NAMES.index("vj_count")])
