"""Multi-GPU sharding of the decombine hot path: one process per GPU, reads split into contiguous ranges by rank, a final
gather of the DCR tuples on rank 0 and a sum of the counters — over RCCL / xGMI through the library's own binding
(include/dcrx.h "multi-GPU": dcrx_comm_*, dcrx_decombine_sharded; decombinator_amd._native.Comm).  No other framework is on
this path: nothing here imports torch.

The reference is single-process: nothing here mirrors reference code.  decombinator_sharded() is the product
entry (the whole stage over the ranks); TupleGather is what bench.py times (device-resident shards, tuples gathered per step);
decombine_sharded_step() is one synchronous step through the C entry dcrx_decombine_sharded.  Each read's result depends only
on that read and the replicated tables (reference decombine.py:534-585 has no cross-read state but the additive Counter,
:598), so the only exchange is the final one:

  * contiguous shards, so that concatenating the ranks' outputs in rank order reproduces the reference's input-order `.n12`
    (outdata.append, :1039);
  * DCR tuples = the narrow tuples of the status-OK records + one bit per read, left by the decombine call itself (the
    handle's tuple sink) or compacted from the records;
  * counters: all-reduce (sum) of the uint64[32] block.

Two small interfaces keep the protocol testable without a GPU (tests/gloo_backend.py implements both over gloo on host
memory; the product's implementations are _native.Comm and RcclBackend below):

  communicator   world, rank, allgather_host(array), allreduce_host_u64(values), allgather_bytes(blob),
                 allgather_object(obj), gather_bytes(blob, dst), barrier()            — host memory, synchronous
  backend        what TupleGather needs of a device: buffers, a side stream's ordering, the count exchange and the
                 exact-size transfers (class RcclBackend documents the methods)
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the reads rank `rank` owns."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def reduce_counters(comm, counters) -> np.ndarray:
    """Sum of every rank's uint64[32] counter block, on every rank."""
    out = np.ascontiguousarray(counters, dtype=np.uint64).copy()
    if comm is not None and comm.world > 1:
        out = comm.allreduce_host_u64(out)
    return out


def gather_bytes(comm, blob: bytes, dst: int = 0):
    """Every rank's bytes on `dst`, in rank order (a list of bytes objects there, None elsewhere).  What the sharded stage
    sends to rank 0: the rows' text, assembled on the rank that holds the reads."""
    return comm.gather_bytes(blob, dst)


def gather_exact(comm, hits, index, dst: int = 0):
    """Gathers each rank's (k_r, 16) uint8 tuple block and (k_r,) int64 index block on `dst`, concatenated in rank order.
    Returns (hits, index) on dst and (None, None) elsewhere."""
    hits = np.ascontiguousarray(hits, dtype=np.uint8)
    index = np.ascontiguousarray(index, dtype=np.int64)
    assert hits.ndim == 2 and hits.shape[1] == 16 and index.shape[0] == hits.shape[0]
    if comm is None or comm.world == 1:
        return hits, index
    parts = comm.gather_bytes(hits.tobytes() + index.tobytes(), dst)
    if comm.rank != dst:
        return None, None
    hs, ix = [], []
    for p in parts:
        k = len(p) // 24
        hs.append(np.frombuffer(p[:16 * k], dtype=np.uint8).reshape(-1, 16))
        ix.append(np.frombuffer(p[16 * k:], dtype=np.int64))
    return np.concatenate(hs), np.concatenate(ix)


def pack_tuples12(rec):
    """Host-side twin of dcrx_compact_hits_packed_device: the status-OK records of `rec` (RECORD_DTYPE) as
    (k, 3) uint32 tuples in read order, and the bitmap of the reads they belong to ((n + 63) // 64 uint64)."""
    ok = rec["status"] == 0
    r = rec[ok]
    w = np.zeros((len(r), 3), dtype=np.uint32)
    w[:, 0] = r["v"].astype(np.uint32) | (r["j"].astype(np.uint32) << 12) | (r["vdel"].astype(np.uint32) << 24)
    w[:, 1] = r["v_start"].astype(np.uint32) | (r["j_end"].astype(np.uint32) << 9) | (r["ins_start"].astype(np.uint32) << 18)
    w[:, 2] = r["ins_len"].astype(np.uint32) | (r["jdel"].astype(np.uint32) << 9) | (r["frame"].astype(np.uint32) << 17)
    bits = np.zeros(((len(rec) + 63) // 64) * 64, dtype=np.uint8)
    bits[:len(rec)] = ok
    bitmap = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1, 8).view(np.uint64).reshape(-1)
    return w, bitmap


def bitmap_indices(bitmap, n_reads: int):
    """Read indices (ascending) whose bit is set."""
    bits = np.unpackbits(np.ascontiguousarray(bitmap, dtype=np.uint64).view(np.uint8), bitorder="little")[:n_reads]
    return np.nonzero(bits)[0]


def pack_tuples8(rec):
    """Host-side twin of dcrx_compact_hits_packed8_device: (k, 2) uint32 tuples in read order and the bitmap."""
    from . import _native as nat
    _, bitmap = pack_tuples12(rec)
    return nat.pack_tuples8(rec), bitmap


class _Event:
    """An event of the side or the main stream, re-recorded in place (no allocation per step)."""

    def __init__(self, nat):
        self.ev = nat.Event(timing=False)

    def record(self, stream_ptr):
        self.ev.record(stream_ptr)
        return self

    def synchronize(self):
        self.ev.synchronize()


class RcclBackend:
    """What TupleGather needs of a device, on a GPU: device buffers, a side stream ordered against the stream the decombine
    calls run on (`main_stream`: a hipStream_t pointer, None = the default stream), the count exchange and the exact-size
    transfers over RCCL (nat.Comm; comm None = one rank, nothing travels).

      buffer(nbytes) -> object with .ptr (.np on host backends: the same bytes as an array)
      host_counts(world) -> uint64 array the counts are copied into (pinned here)
      side_wait_main() / main_wait_side()        stream ordering
      new_event() -> an event the calls below record; main_wait(event) / side_wait(event); event.synchronize() for the host
      side_ptr                                   the pointer compaction launches take
      zero(buf, offset, nbytes)                  on the side stream
      exchange_counts(n_buf, counts_buf, host)   all-gather of the ranks' counts and their copy to the host -> event
      post(slot_bufs, counts)                    the exact-size transfers of one slot (rank 0: one grouped receive) -> event
    """
    cuda = True

    def __init__(self, nat, comm=None, main_stream=None):
        self.nat, self.comm = nat, comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0
        self.main = main_stream
        self.side = nat.Stream()
        self.side_ptr = self.side.ptr
        self._tmp = _Event(nat)

    def buffer(self, nbytes: int):
        buf = self.nat.DeviceBuffer(max(int(nbytes), 16))
        self.nat.check(self.nat.lib().dcrx_memset_device(buf.ptr, 0, buf.nbytes))
        return buf

    def host_counts(self, world: int):
        return self.nat.pinned_empty((world,), np.uint64)

    def side_wait_main(self):
        self.side.wait_event(self._tmp.record(self.main).ev)

    def main_wait_side(self):
        self.main_wait(self._tmp.record(self.side_ptr))

    def new_event(self):
        return _Event(self.nat)

    def main_wait(self, event):
        self.nat.check(self.nat.lib().dcrx_stream_wait_event(self.main, event.ev.ptr))

    def side_wait(self, event):
        pass      # (the transfers ran on the side stream: what follows on it is behind them)

    def zero(self, buf, offset: int, nbytes: int):
        self.nat.check(self.nat.lib().dcrx_memset_device_async(buf.ptr + offset, 0, nbytes, self.side_ptr))

    def exchange_counts(self, n_buf, counts_buf, host, event):
        nat = self.nat
        if self.world > 1:
            self.comm.allgather(n_buf.ptr, counts_buf.ptr, 8, self.side_ptr)
        else:
            nat.check(nat.lib().dcrx_memcpy_d2d_async(counts_buf.ptr, n_buf.ptr, 8, self.side_ptr))
        nat.check(nat.lib().dcrx_memcpy_d2h_async(host.ctypes.data, counts_buf.ptr, 8 * self.world, self.side_ptr))
        return event.record(self.side_ptr)

    def post(self, own_msg, peer_msgs, nbytes, event):
        """nbytes[r]: the exact size of rank r's message.  Rank 0 receives every peer's into peer_msgs[r]; the others send."""
        if self.world > 1:
            self.comm.gather_v(own_msg.ptr, nbytes[self.rank], [m.ptr if m is not None else None for m in peer_msgs] if self.rank == 0 else None,
                               nbytes if self.rank == 0 else None, 0, self.side_ptr)
        return event.record(self.side_ptr)

    def to_host(self, buf, nbytes: int) -> np.ndarray:
        return buf.to_host(np.uint8, nbytes)

    def synchronize(self):
        self.nat.synchronize()


class TupleGather:
    """Per-step gather of the DCR tuples on rank 0, exact sizes, nothing truncated.

    A tuple travels as the tag set's narrow tuple when `tables` is given (include/dcrx.h dcrx_tuple_layout: 5 bytes for
    human beta at 150 nt; the receiver holds the same tables), as 8 bytes when only the V tags' jumps are given (`v_jumps`;
    dcrx_compact_hits_packed8_device) and as 12 bytes otherwise, in read order, plus one bit per read saying which reads
    decombined: at eight ranks and 20 G reads/s per rank the tuples are what the xGMI links into rank 0 carry.  Per step, on a
    side stream beside the scan of the following step:

      1. the step's message: left by the decombine call itself (tuple sink) or compacted from its records -> tuples, bitmap, count;
      2. count exchange: all-gather of the ranks' counts, copied to pinned host memory;
      3. one step later, when the counts have arrived: exact-size point-to-point transfers
         (rank r sends one message: its bitmap, then count_r tuples; rank 0 posts the receives from all peers as one group).

    The caller alternates between `depth` record buffers (`records()`), the scan of step k + depth waits for the compaction
    of step k (`before_scan`), and a buffer set is reused only after its transfers completed.  `compact` replaces step 1
    (tests feed tuples made on the host).  `backend`: RcclBackend on a GPU; tests/gloo_backend.GlooBackend on host memory."""

    TUPLE_BYTES = 12      # (the class default; an instance with v_jumps carries 8)

    def __init__(self, n_reads: int, backend, depth: int = 2, compact=None, v_jumps=None,
                 n_v: int = None, n_j: int = None, tables=None, max_read_len: int = None, use_sink: bool = True):
        from . import _native as nat
        self.nat, self.be = nat, backend
        world, rank = backend.world, backend.rank
        # `tables` (+ max_read_len): the narrow tuple of include/dcrx.h — widths from the tag tables, neither ins_start nor
        # ins_len on the wire (5 bytes for human beta at 150 nt); the receiver holds the same tables (nat.TupleCodec)
        self.tables, self.codec = tables, None
        if tables is not None:
            try:
                self.codec = nat.TupleCodec(tables, int(max_read_len))
            except nat.DcrxError:
                self.codec = None             # wider than 64 bits: the fixed forms below
        self.v_jumps = None if v_jumps is None else list(v_jumps)
        # the 8-byte tuple holds v in 11 bits and j in 9 (include/dcrx.h, dcrx_compact_hits_packed8_device) and re-derives
        # ins_start from the V tag's jump: a table it does not describe (n_v / n_j from dcrx_tables_info, when the caller gives
        # them) or a tag set too large for those fields travels as 12-byte tuples instead — never as masked garbage
        if self.v_jumps is not None:
            if n_v is not None and n_v != len(self.v_jumps):
                raise ValueError(f"TupleGather: {len(self.v_jumps)} V jumps for a table of {n_v} V tags")
            if len(self.v_jumps) >= 2048 or (n_j is not None and n_j >= 512):
                self.v_jumps = None
        self.TUPLE_BYTES = self.codec.bytes if self.codec is not None else (8 if self.v_jumps is not None else 12)
        self.world, self.rank, self.n_reads = world, rank, n_reads
        self.cuda = bool(backend.cuda)
        self.words = (n_reads + 63) // 64
        self.k = 0
        self.compact = compact
        # the tuple sink (dcrx_set_tuple_sink): the decombine call of a step leaves the step's message itself, on its own
        # stream — no compaction pass beside the next step's scan; the side stream carries the count exchange and the transfers
        self.sink = bool(use_sink and self.codec is not None and self.cuda and compact is None)
        self.slots = []
        # a rank's message of a step: its bitmap, then its tuples — one buffer, one transfer per peer and step
        self.bm_bytes = self.words * 8
        msg_bytes = self.bm_bytes + n_reads * self.TUPLE_BYTES
        for _ in range(depth):
            msg = backend.buffer(msg_bytes)
            slot = {
                "rec": backend.buffer(n_reads * 16),
                "msg": msg,
                "n": backend.buffer(8),
                "counts": backend.buffer(8 * world),
                "counts_host": backend.host_counts(world),
                "counted": backend.new_event(), "has_counts": False,      # event: counts_host is filled
                "compacted": backend.new_event(), "has_compacted": False,  # event: the compaction that read this slot's records is done
                "moved": backend.new_event(), "in_flight": False,          # event: the slot's last transfers are done
                "posted": True,        # the transfers of the slot's last use have been posted
                "step": -1,
            }
            if not self.cuda:           # host memory: the hooks of the tests write the message through these views
                slot["bitmap"] = msg.np[:self.bm_bytes].view(np.int64)
                slot["hits"] = msg.np[self.bm_bytes:]
            if rank == 0:
                slot["g_msg"] = [msg if r == 0 else backend.buffer(msg_bytes) for r in range(world)]
            self.slots.append(slot)

    def records(self):
        """The record buffer the next scan writes (call before_scan() first): an object with .ptr (and .np on host memory)."""
        return self.slots[self.k % len(self.slots)]["rec"]

    def before_scan(self) -> None:
        """The stream of the decombine calls waits until the slot's previous compaction has read its records."""
        s = self.slots[self.k % len(self.slots)]
        if self.sink:
            # the call about to be queued writes the slot's message: its last transfers must be done (they were posted a step ago)
            self._post(s)
            if s["in_flight"]:
                self.be.main_wait(s["moved"])
                s["in_flight"] = False
            self.nat.set_tuple_sink(self.tables, self.codec, s["msg"].ptr, self.n_reads, s["n"].ptr)
            return
        if s["has_compacted"] and self.cuda:
            self.be.main_wait(s["compacted"])

    def _sizes(self, counts):
        return [self.bm_bytes + int(c) * self.TUPLE_BYTES for c in counts]

    def _post(self, s) -> None:
        """Exact-size transfers of a slot whose counts have arrived."""
        if s["posted"]:
            return
        if s["has_counts"]:
            s["counted"].synchronize()       # the host waits for the counts (a few bytes, one step old)
        counts = [int(x) for x in s["counts_host"].tolist()]
        s["posted"] = True
        self.be.post(s["msg"], s.get("g_msg"), self._sizes(counts), s["moved"])
        s["in_flight"] = True

    def step(self, n_reads: int) -> None:
        """After the scan of this step has been queued on the main stream."""
        nat, be = self.nat, self.be
        s = self.slots[self.k % len(self.slots)]
        prev = self.slots[(self.k - 1) % len(self.slots)] if self.k else None
        # The previous step's transfers go out FIRST — its counts are nearly a step old: the host waits for them here, with this
        # step's scan already queued — so that on the side stream they stand in front of the wait for this step's scan and move
        # beside it.  (Posted behind this step's count exchange, as rounds 3-5 had it, they — and the event the slot's next user
        # waits for — stood behind this step's scan: a bubble of 40 us per step at the start of the scan after next, seen at one
        # rank once the native backend recorded that event there too: profiles/r06/gather_one_rank.log.)
        if prev is not None and prev is not s:
            self._post(prev)
        be.side_wait_main()
        self._post(s)                # (a slot is reused only when its last transfers were posted ...
        if s["in_flight"]:
            be.side_wait(s["moved"])     # ... and are done: on a device the side stream's own order, on host memory a wait)
        if n_reads < self.n_reads and not self.sink:
            be.zero(s["msg"], 0, self.bm_bytes)      # a short batch (the end of a shard): no stale bits beyond its reads
        if self.sink:
            pass                     # (the call has left the message and the count on its own stream: nothing to compact)
        elif self.compact is not None:
            self.compact(s, n_reads)
        elif self.codec is not None:
            nat.compact_hits_narrow_device(self.tables, self.codec, s["rec"].ptr, n_reads, s["msg"].ptr, s["n"].ptr, be.side_ptr, n_slots=self.n_reads)
        else:
            fn = nat.lib().dcrx_compact_hits_packed8_device if self.TUPLE_BYTES == 8 else nat.lib().dcrx_compact_hits_packed_device
            nat.check(fn(s["rec"].ptr, n_reads, s["msg"].ptr + self.bm_bytes, s["msg"].ptr, s["n"].ptr, be.side_ptr))
        if self.cuda:
            s["compacted"].record(be.side_ptr)
            s["has_compacted"] = True
        be.exchange_counts(s["n"], s["counts"], s["counts_host"], s["counted"])
        s["has_counts"] = True
        s["posted"] = False
        s["step"] = self.k
        self.k += 1

    def finish(self) -> None:
        """Posts what is still to be posted and makes the main stream wait for every transfer."""
        if self.sink:
            self.nat.set_tuple_sink(self.tables, None)      # (calls outside the gather's steps leave no message)
        for s in self.slots:
            self._post(s)
        self.be.main_wait_side()
        for s in self.slots:
            s["in_flight"] = False

    def gathered(self, step: int):
        """On rank 0, after finish(): the tuples of `step` (one of the last `depth` steps) as (records, read index
        within the rank, rank) per rank, re-expanded from the wire form and the bitmaps."""
        s = next(x for x in self.slots if x["step"] == step)
        self.be.synchronize()
        counts = [int(x) for x in s["counts_host"].tolist()]
        sizes = self._sizes(counts)
        out = []
        for r in range(self.world):
            raw = self.be.to_host(s["g_msg"][r], sizes[r])
            if self.codec is not None:
                rec, idx = self.codec.unpack(raw, self.n_reads, counts[r])
                out.append((rec, idx, r))
                continue
            w = raw[self.bm_bytes:].view(np.uint32).reshape(-1, self.TUPLE_BYTES // 4)
            idx = bitmap_indices(raw[:self.bm_bytes].view(np.uint64), self.n_reads)
            rec = self.nat.unpack_tuples8(w, self.v_jumps) if self.TUPLE_BYTES == 8 else self.nat.unpack_tuples12(w)
            out.append((rec, idx, r))
        return out

    def check(self, n_hits_local: int) -> None:
        """After the run: the last step's counts and bitmaps are consistent on rank 0."""
        self.finish()
        self.be.synchronize()
        last = self.slots[(self.k - 1) % len(self.slots)]
        counts = [int(x) for x in last["counts_host"].tolist()]
        if counts[self.rank] != n_hits_local:
            raise RuntimeError(f"rank {self.rank}: exchanged count {counts[self.rank]} != {n_hits_local} decombined reads")
        if self.rank == 0:
            for r, (rec, idx, _) in enumerate(self.gathered(self.k - 1)):
                if len(rec) != counts[r] or len(idx) != counts[r]:
                    raise RuntimeError(f"rank {r}: {len(rec)} tuples, {len(idx)} bitmap bits, {counts[r]} announced")


def decombine_sharded_step(tables, comm, codec, batch_c, d_records_ptr: int, d_counters_ptr: int, d_message_ptr: int, n_slots: int,
                           d_gathered_ptrs=None, orientation="reverse", allow_ns=False, lenthreshold=130, stream=None):
    """One step of a sharded run through the C entry dcrx_decombine_sharded (include/dcrx.h): the hot path on this rank's
    shard, the counts all-gathered, the tuple messages on rank 0 in exact sizes, the counters summed — all inside the library.
    Returns the ranks' counts of decombined reads; the transfers and the all-reduce are on `stream`, not yet waited for."""
    from . import _native as nat
    C = nat.C
    cfg = nat.make_cfg(orientation, allow_ns, lenthreshold, 0)
    counts = (C.c_uint64 * comm.world)()
    ptrs = None
    if comm.rank == 0 and comm.world > 1:
        ptrs = (C.c_void_p * comm.world)(*[int(p) if p else None for p in d_gathered_ptrs])
    nat.check(nat.lib().dcrx_decombine_sharded(tables.handle, comm.handle, C.byref(cfg), C.byref(batch_c), d_records_ptr, d_counters_ptr,
                                               C.byref(codec.layout), d_message_ptr, int(n_slots), ptrs, counts, stream))
    return [int(x) for x in counts]


def plan_fastq_shards(comm, paths, records_per_unit: int = 1):
    """Byte ranges [(begin, end) per file] of the records this rank reads — contiguous shards in rank order, whole records, the
    files of a pair cut at the same record — or None when the files cannot be read in shards (gzipped, carriage returns, not
    whole four-line records, pairs of different length): the caller then reads unsharded.

    No rank reads another rank's bytes to find the cuts: each counts the newlines of its own S/W bytes of every file (memory
    speed: dcrx_fastq_lines), the counts are all-gathered, and record k of a four-line file starts at line 4 k — a rank owns the
    records whose first line starts behind a newline of its range of the FIRST file (line 0: rank 0); the offsets of the cut
    lines are looked up by the ranks whose ranges hold them (a second pass that stops at the line) and all-gathered.
    records_per_unit = 2 keeps the record pairs of bc_read R1 together (reference decombine.py:956-961: zip over one generator
    consumes two records per iteration).  Every rank must call this (four small collectives)."""
    import os
    from . import _native as nat
    world, rank = comm.world, comm.rank
    if any(str(p).endswith(".gz") for p in paths):
        return None
    # (whatever goes wrong on one rank — a file it cannot open — is part of what is exchanged: no rank leaves before the
    # collectives below, and all of them then read unsharded, where the readers report the file in their own words)
    mine, sizes = "ERR", None
    try:
        sizes = [os.path.getsize(p) for p in paths]
        mine = []
        for p, size in zip(paths, sizes):
            b, e = rank * size // world, (rank + 1) * size // world
            n, _, cr, _ = nat.fastq_lines(p, b, e)
            last_nl = True
            if rank == world - 1 and size:
                with open(p, "rb") as fh:
                    fh.seek(size - 1)
                    last_nl = fh.read(1) == b"\n"
            mine.append((n, cr, last_nl, size))
    except Exception:
        mine = "ERR"
    every = comm.allgather_object(mine)
    if any(x == "ERR" for x in every):
        return None
    counts = [[every[r][f][0] for r in range(world)] for f in range(len(paths))]
    totals = [sum(c) for c in counts]
    if any(every[r][f][1] or not every[r][f][2] or every[r][f][3] != sizes[f] for r in range(world) for f in range(len(paths))):
        return None
    if any(t % 4 for t in totals) or len(set(totals)) != 1:
        return None
    n_records = totals[0] // 4
    unit = 4 * records_per_unit
    # the first record (a multiple of records_per_unit) of each rank: ownership by the first file's newlines
    prefix = [0] * (world + 1)
    for r in range(world):
        prefix[r + 1] = prefix[r] + counts[0][r]
    first = [0] * (world + 1)
    for r in range(1, world):
        first[r] = min(-(-(prefix[r] + 1) // unit) * records_per_unit, n_records)      # smallest unit start line >= prefix + 1
        first[r] = max(first[r], first[r - 1])
    first[world] = n_records
    # offsets of the cut lines, each looked up by the rank whose byte range of that file holds the line's newline
    found = {}
    for f, (p, size) in enumerate(zip(paths, sizes)):
        pre = [0] * (world + 1)
        for r in range(world):
            pre[r + 1] = pre[r] + counts[f][r]
        b, e = rank * size // world, (rank + 1) * size // world
        for r in range(1, world):
            line = 4 * first[r]
            if line == 0:
                found[(f, r)] = 0
            elif line >= totals[f]:
                if rank == world - 1:
                    found[(f, r)] = size
            elif pre[rank] < line <= pre[rank + 1]:
                try:
                    _, off, _, _ = nat.fastq_lines(p, b, e, nth=line - pre[rank], count=False)
                except Exception:
                    off = None
                found[(f, r)] = off
    allfound = comm.allgather_object(found)
    cuts = {}
    for d in allfound:
        cuts.update(d)
    out = []
    for f, size in enumerate(sizes):
        begin = 0 if rank == 0 else cuts.get((f, rank))
        end = size if rank == world - 1 else cuts.get((f, rank + 1))
        if begin is None or end is None:
            out = None                    # (cannot happen for consistent counts; be safe — and stay for the collective below)
            break
        out.append((begin, max(begin, end)))
    # The cuts assume four-line records (record k starts at line 4 k).  A file whose sequences run over several lines can have a
    # multiple of four lines all the same: every rank runs its own shard through the strict reader once (dcrx_fastq_open_range:
    # parsed by several threads from the page cache, nothing kept) and the verdicts are exchanged; one rank that meets anything
    # but plain four-line records sends every rank back to the unsharded readers (readfq, which is the definition: reference
    # decombine.py:228-265) — before a row exists, instead of an error in the middle of one rank's loop.
    ok = out is not None
    if ok:
        try:
            for p, (begin, end) in zip(paths, out):
                if end <= begin:
                    continue
                with nat.FastqReader(p, False, byte_range=(begin, end)) as rd:
                    while rd.next(1 << 20).n:
                        pass
        except Exception:
            ok = False
    oks = comm.allgather_object(ok)
    return out if all(oks) else None


def decombinator_sharded(inputargs: dict, comm, device_index: int | None = None):
    """The decombine stage over all ranks of `comm` (one process per GPU; a communicator as the module's head describes:
    _native.Comm over RCCL): the multi-GPU form of decombinator_amd.decombine.decombinator().

    The input is read in shards: rank r reads the records of its contiguous share of the FASTQ bytes and nothing else of the
    files (plan_fastq_shards: plain four-line files; the R1 / R2 files of a pair are cut at the same record), decombines them
    on its own GPU and assembles their rows; the rows' bytes are gathered on rank 0 (gather_bytes) and concatenated in rank
    order — contiguous shards, so that this is the input order the reference's outdata.append keeps (decombine.py:1039); the
    counters are summed over the ranks before rank 0 prints the totals and writes the summary log.  Files that cannot be cut
    (gzipped, multi-line records, carriage returns) are read whole by every rank, batches dealt round-robin, as before.
    Returns the rows (an N12Rows, as decombinator() does) on rank 0 and None on the other ranks.  No data-path collective:
    the gather of the rows at the end and one all-reduce of 64 integers."""
    from decombinator_amd import _native as nat
    from decombinator_amd import decombine as dec

    if comm is None:
        raise RuntimeError("decombinator_sharded needs a communicator (decombinator_amd._native.comm_from_env())")
    rank, world = comm.rank, comm.world
    if device_index is not None:
        nat.check(nat.lib().dcrx_set_device(int(device_index)))

    def reduce_counts(counts):
        # the Counter's keys differ between ranks (a key appears with its first increment): agree on the union first
        mine = sorted(k for k in counts if k not in ("start_time", "end_time"))
        every = comm.allgather_object(mine)
        keys = sorted(set(k for ks in every for k in ks))
        if keys:
            sums = comm.allreduce_host_u64(np.array([int(counts.get(k, 0)) for k in keys], dtype=np.uint64))
            for k, v in zip(keys, sums.tolist()):
                counts[k] = int(v)

    def exchange_error(err):
        # every rank reports whether its part raised; if any did, every rank raises here, before the first collective of
        # the results (a rank that stopped alone would leave the others waiting for ever)
        flag = comm.allreduce_host_u64(np.array([1 if err is not None else 0], dtype=np.uint64))
        if int(flag[0]) == 0:
            return
        said = comm.allgather_object(None if err is None else f"{type(err).__name__}: {err}")
        if err is not None:
            raise err
        raise RuntimeError("decombinator_sharded: " + "; ".join(f"rank {r}: {m}" for r, m in enumerate(said) if m))

    rows = dec.decombinator(inputargs, shard=(rank, world), reduce_counts=reduce_counts, exchange_error=exchange_error,
                            plan_shards=lambda paths, w, r, unit: plan_fastq_shards(comm, paths, unit))
    chunks = rows._tagged_chunks()
    # a rank's message: per chunk (tag, rows, bytes) — three int64 — then the chunks' text
    head = np.array([[t, n, len(b)] for t, b, n in chunks], dtype=np.int64).reshape(-1, 3)
    msg = np.int64(len(chunks)).tobytes() + head.tobytes() + b"".join(b for _, b, _ in chunks)
    parts = gather_bytes(comm, msg, dst=0)
    if rank != 0:
        return None
    tagged = []
    for r, part in enumerate(parts):
        k = int(np.frombuffer(part[:8], dtype=np.int64)[0])
        hd = np.frombuffer(part[8:8 + 24 * k], dtype=np.int64).reshape(-1, 3)
        at = 8 + 24 * k
        for t, n, nb in hd.tolist():
            tagged.append(((t, r), part[at:at + nb], n))
            at += nb
    merged = dec.N12Rows()
    for (t, r), blob, n in sorted(tagged, key=lambda c: c[0]):      # (sharded input: every tag is its rank; round-robin: the batch index)
        merged._tag = t
        merged._add_blob(blob, n)
    return merged
