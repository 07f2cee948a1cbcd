"""Multi-GPU sharding of the decombine hot path: one process per GPU, reads split
into contiguous ranges by rank, a final gather of the DCR tuples on rank 0 and a
sum of the counters (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).

The reference is single-process: nothing here mirrors reference
code.  decombinator_sharded() is the product entry (the whole stage over the ranks);
TupleGather is what bench.py times (device-resident shards, tuples gathered per step).  Each read's result depends only on that read and the replicated tables
(reference decombine.py:534-585 has no cross-read state but the additive
Counter, :598), so the only exchange is the final one:

  * contiguous shards, so that concatenating the ranks' outputs in rank order
    reproduces the reference's input-order `.n12` (outdata.append, :1039);
  * DCR tuples = the status-OK 16-byte records + their global read indices,
    compacted on the device (dcrx_compact_hits_device);
  * counters: all_reduce(sum) of the uint64[32] block.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of the reads rank `rank` owns."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def reduce_counters(counters: torch.Tensor) -> torch.Tensor:
    """Sum of every rank's int64[32] counter block, on every rank."""
    out = counters.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out


def gather_exact(hits: torch.Tensor, index: torch.Tensor, dst: int = 0):
    """Gathers each rank's (k_r, 16) uint8 tuple block and (k_r,) int64 index block on
    `dst`, concatenated in rank order.  Sizes are exchanged first (one small
    all_gather), then one padded gather per array.  Returns (hits, index) on dst and
    (None, None) elsewhere."""
    assert hits.dtype == torch.uint8 and hits.dim() == 2 and hits.shape[1] == 16
    assert index.dtype == torch.int64 and index.shape[0] == hits.shape[0]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return hits, index
    world, rank = dist.get_world_size(), dist.get_rank()
    k = torch.tensor([hits.shape[0]], dtype=torch.int64, device=hits.device)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    counts = [int(x.item()) for x in ks]
    kmax = max(counts) if counts else 0
    pad_h = torch.zeros((kmax, 16), dtype=torch.uint8, device=hits.device)
    pad_i = torch.zeros((kmax,), dtype=torch.int64, device=hits.device)
    pad_h[:hits.shape[0]] = hits
    pad_i[:index.shape[0]] = index
    if rank == dst:
        gh = [torch.empty_like(pad_h) for _ in range(world)]
        gi = [torch.empty_like(pad_i) for _ in range(world)]
        dist.gather(pad_h, gh, dst=dst)
        dist.gather(pad_i, gi, dst=dst)
        return (torch.cat([g[:c] for g, c in zip(gh, counts)]),
                torch.cat([g[:c] for g, c in zip(gi, counts)]))
    dist.gather(pad_h, None, dst=dst)
    dist.gather(pad_i, None, dst=dst)
    return None, None


def pack_tuples12(rec):
    """Host-side twin of dcrx_compact_hits_packed_device: the status-OK records of `rec` (RECORD_DTYPE) as
    (k, 3) uint32 tuples in read order, and the bitmap of the reads they belong to ((n + 63) // 64 uint64)."""
    import numpy as np
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
    import numpy as np
    bits = np.unpackbits(np.ascontiguousarray(bitmap, dtype=np.uint64).view(np.uint8), bitorder="little")[:n_reads]
    return np.nonzero(bits)[0]


def pack_tuples8(rec):
    """Host-side twin of dcrx_compact_hits_packed8_device: (k, 2) uint32 tuples in read order and the bitmap."""
    from . import _native as nat
    _, bitmap = pack_tuples12(rec)
    return nat.pack_tuples8(rec), bitmap


class TupleGather:
    """Per-step gather of the DCR tuples on rank 0, exact sizes, nothing truncated.

    A tuple travels as 8 bytes when the V tags' jumps are given (`v_jumps`; dcrx_compact_hits_packed8_device: every field of
    the record but ins_start, which rank 0 re-derives from the tag file's jump — tag sets of < 2048 V and < 512 J tags) and
    as 12 bytes otherwise (dcrx_compact_hits_packed_device), in read order, plus one bit per read saying which reads
    decombined: at eight ranks and 20 G reads/s per rank the tuples are what the xGMI links into rank 0 carry.  Per step, on a
    side stream beside the scan of the following step:

      1. compaction of the step's records -> tuples, bitmap, count (on the device);
      2. count exchange: all_gather of the ranks' counts, copied to pinned host memory;
      3. one step later, when the counts have arrived: exact-size point-to-point transfers
         (rank r sends one message: its bitmap, then count_r tuples; rank 0 posts the receives from all peers as one group).

    The caller alternates between `depth` record buffers (`records()`), the scan of step k + depth waits for
    the compaction of step k (`before_scan`), and a buffer set is reused only after its transfers completed.
    `compact` replaces step 1 (tests feed tuples made on the host); device None or CPU runs without streams
    (gloo), else on a CUDA side stream (RCCL)."""

    TUPLE_BYTES = 12      # (the class default; an instance with v_jumps carries 8)

    def __init__(self, n_reads: int, world: int, rank: int, device: torch.device, depth: int = 2, compact=None, v_jumps=None,
                 n_v: int = None, n_j: int = None, tables=None, max_read_len: int = None, use_sink: bool = True):
        from . import _native as nat
        self.nat = nat
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
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.device = device if self.cuda else torch.device("cpu")
        self.words = (n_reads + 63) // 64
        self.k = 0
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.compact = compact
        # the tuple sink (dcrx_set_tuple_sink): the decombine call of a step leaves the step's message itself, on its own
        # stream — no compaction pass beside the next step's scan; the side stream carries the count exchange and the transfers
        self.sink = bool(use_sink and self.codec is not None and self.cuda and compact is None)
        self.slots = []
        dev = self.device
        # a rank's message of a step: its bitmap, then its tuples — one buffer, one transfer per peer and step
        bm_bytes = self.words * 8
        msg_bytes = bm_bytes + n_reads * self.TUPLE_BYTES

        def message():
            m = torch.zeros(msg_bytes, dtype=torch.uint8, device=dev)
            return m, m[:bm_bytes].view(torch.int64), m[bm_bytes:]

        self.bm_bytes = bm_bytes
        for _ in range(depth):
            msg, bitmap, hits = message()
            slot = {
                "rec": torch.empty(n_reads * 16, dtype=torch.uint8, device=dev),
                "msg": msg,
                "hits": hits,
                "bitmap": bitmap,
                "n": torch.zeros(1, dtype=torch.int64, device=dev),
                "counts": [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)],
                "counts_host": torch.zeros(world, dtype=torch.int64, pin_memory=self.cuda),
                "counted": None,       # event: counts_host is filled
                "compacted": None,     # event: the compaction that read this slot's records is done
                "posted": True,        # the transfers of the slot's last use have been posted
                "work": [],
                "step": -1,
            }
            if rank == 0:
                peers = [(slot["msg"], slot["bitmap"], slot["hits"]) if r == 0 else message() for r in range(world)]
                slot["g_msg"] = [p[0] for p in peers]
                slot["g_bitmap"] = [p[1] for p in peers]
                slot["g_hits"] = [p[2] for p in peers]
            self.slots.append(slot)

    def records(self) -> torch.Tensor:
        """The record buffer the next scan writes (call before_scan() first)."""
        return self.slots[self.k % len(self.slots)]["rec"]

    def before_scan(self) -> None:
        """The current stream waits until the slot's previous compaction has read its records."""
        s = self.slots[self.k % len(self.slots)]
        if self.sink:
            # the call about to be queued writes the slot's message: its last transfers must be done (they were posted a step ago)
            self._post(s)
            for w in s["work"]:
                w.wait()
            s["work"] = []
            self.nat.set_tuple_sink(self.tables, self.codec, s["msg"].data_ptr(), self.n_reads, s["n"].data_ptr())
            return
        ev = s["compacted"]
        if ev is not None and self.cuda:
            torch.cuda.current_stream().wait_event(ev)

    def _side(self):
        return torch.cuda.stream(self.side) if self.cuda else _NullCtx()

    def _post(self, s) -> None:
        """Exact-size transfers of a slot whose counts have arrived."""
        if s["posted"]:
            return
        if s["counted"] is not None:
            s["counted"].synchronize()       # the host waits for the counts (a few bytes, one step old)
        counts = [int(x) for x in s["counts_host"].tolist()]
        s["posted"] = True
        with self._side():
            if self.rank == 0:
                # one grouped call for the receives from every peer (one launch on RCCL, not one per peer)
                ops = [dist.P2POp(dist.irecv, s["g_msg"][r][:self.bm_bytes + counts[r] * self.TUPLE_BYTES], r) for r in range(1, self.world)]
                if ops:
                    s["work"].extend(dist.batch_isend_irecv(ops))
            else:
                s["work"].append(dist.isend(s["msg"][:self.bm_bytes + counts[self.rank] * self.TUPLE_BYTES], dst=0))

    def step(self, n_reads: int) -> None:
        """After the scan of this step has been queued on the current stream."""
        nat = self.nat
        s = self.slots[self.k % len(self.slots)]
        prev = self.slots[(self.k - 1) % len(self.slots)] if self.k else None
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream())
        with self._side():
            self._post(s)                # (a slot is reused only when its last transfers were posted ...)
            for w in s["work"]:          # ... and are done
                w.wait()
            s["work"] = []
            if n_reads < self.n_reads and not self.sink:
                s["bitmap"].zero_()      # a short batch (the end of a shard): no stale bits beyond its reads
            if self.sink:
                pass                     # (the call has left the message and the count on its own stream: nothing to compact)
            elif self.compact is not None:
                self.compact(s, n_reads)
            elif self.codec is not None:
                nat.compact_hits_narrow_device(self.tables, self.codec, s["rec"].data_ptr(), n_reads, s["msg"].data_ptr(),
                                               s["n"].data_ptr(), self.side.cuda_stream, n_slots=self.n_reads)
            else:
                fn = nat.lib().dcrx_compact_hits_packed8_device if self.TUPLE_BYTES == 8 else nat.lib().dcrx_compact_hits_packed_device
                nat.check(fn(s["rec"].data_ptr(), n_reads, s["hits"].data_ptr(), s["bitmap"].data_ptr(), s["n"].data_ptr(),
                             self.side.cuda_stream))
            if self.cuda:
                s["compacted"] = self.side.record_event()
            if self.world > 1:
                dist.all_gather(s["counts"], s["n"])
            else:
                s["counts"][0].copy_(s["n"])
            s["counts_host"].copy_(torch.cat(s["counts"]), non_blocking=True)
            s["counted"] = self.side.record_event() if self.cuda else None
            s["posted"] = False
            s["step"] = self.k
        self.k += 1
        if prev is not None and prev is not s:
            self._post(prev)             # the previous step's counts are one step old by now

    def finish(self) -> None:
        """Posts what is still to be posted and makes the current stream wait for every transfer."""
        if self.sink:
            self.nat.set_tuple_sink(self.tables, None)      # (calls outside the gather's steps leave no message)
        for s in self.slots:
            self._post(s)
        with self._side():
            for s in self.slots:
                for w in s["work"]:
                    w.wait()
                s["work"] = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)

    def gathered(self, step: int):
        """On rank 0, after finish(): the tuples of `step` (one of the last `depth` steps) as (records, read index
        within the rank, rank) per rank, re-expanded from the 12-byte form and the bitmaps."""
        import numpy as np
        s = next(x for x in self.slots if x["step"] == step)
        if self.cuda:
            torch.cuda.synchronize()
        counts = [int(x) for x in s["counts_host"].tolist()]
        out = []
        for r in range(self.world):
            if self.codec is not None:
                rec, idx = self.codec.unpack(s["g_msg"][r][:self.bm_bytes + counts[r] * self.TUPLE_BYTES].cpu().numpy(), self.n_reads, counts[r])
                out.append((rec, idx, r))
                continue
            w = s["g_hits"][r][:counts[r] * self.TUPLE_BYTES].cpu().numpy().view(np.uint32).reshape(-1, self.TUPLE_BYTES // 4)
            idx = bitmap_indices(s["g_bitmap"][r].cpu().numpy().view(np.uint64), self.n_reads)
            rec = self.nat.unpack_tuples8(w, self.v_jumps) if self.TUPLE_BYTES == 8 else self.nat.unpack_tuples12(w)
            out.append((rec, idx, r))
        return out

    def check(self, n_hits_local: int) -> None:
        """After the run: the last step's counts and bitmaps are consistent on rank 0."""
        self.finish()
        if self.cuda:
            torch.cuda.synchronize()
        last = self.slots[(self.k - 1) % len(self.slots)]
        counts = [int(x) for x in last["counts_host"].tolist()]
        if counts[self.rank] != n_hits_local:
            raise RuntimeError(f"rank {self.rank}: exchanged count {counts[self.rank]} != {n_hits_local} decombined reads")
        if self.rank == 0:
            for r, (rec, idx, _) in enumerate(self.gathered(self.k - 1)):
                if len(rec) != counts[r] or len(idx) != counts[r]:
                    raise RuntimeError(f"rank {r}: {len(rec)} tuples, {len(idx)} bitmap bits, {counts[r]} announced")


def _backend_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_bytes(blob: bytes, dst: int = 0):
    """Every rank's bytes on `dst`, in rank order (a list of bytes objects there, None elsewhere): the sizes first (one small
    all_gather), then one padded gather of uint8 tensors — over RCCL from device memory on the GPU box, over gloo in the CPU
    tests.  What the sharded stage sends to rank 0: the rows' text, assembled on the rank that holds the reads."""
    import numpy as np
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _backend_device()
    k = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    sizes = [int(x.item()) for x in ks]
    kmax = max(max(sizes), 1)
    pad = torch.zeros(kmax, dtype=torch.uint8, device=dev)
    if len(blob):
        pad[:len(blob)] = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    if rank == dst:
        got = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, got, dst=dst)
        return [g[:n].cpu().numpy().tobytes() for g, n in zip(got, sizes)]
    dist.gather(pad, None, dst=dst)
    return None


def plan_fastq_shards(paths, world: int, rank: int, records_per_unit: int = 1):
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
    every = [None] * world
    dist.all_gather_object(every, mine)
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
    allfound = [None] * world
    dist.all_gather_object(allfound, found)
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
    oks = [None] * world
    dist.all_gather_object(oks, ok)
    return out if all(oks) else None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def decombinator_sharded(inputargs: dict, device_index: int | None = None):
    """The decombine stage over all ranks of the initialised process group (one process per GPU): the multi-GPU form
    of decombinator_amd.decombine.decombinator().

    The input is read in shards: rank r reads the records of its contiguous share of the FASTQ bytes and nothing else of the
    files (plan_fastq_shards: plain four-line files; the R1 / R2 files of a pair are cut at the same record), decombines them
    on its own GPU and assembles their rows; the rows' bytes are gathered on rank 0 (gather_bytes: one padded gather, RCCL on
    the GPU box) and concatenated in rank order — contiguous shards, so that this is the input order the reference's
    outdata.append keeps (decombine.py:1039); the counters are summed over the ranks before rank 0 prints the totals and
    writes the summary log.  Files that cannot be cut (gzipped, multi-line records, carriage returns) are read whole by
    every rank, batches dealt round-robin, as before.  Returns the rows (an N12Rows, as decombinator() does) on rank 0 and
    None on the other ranks.  No data-path collective: the gather of the rows at the end and one all-reduce of 64 integers."""
    import numpy as np

    from decombinator_amd import _native as nat
    from decombinator_amd import decombine as dec

    if not dist.is_initialized():
        raise RuntimeError("decombinator_sharded needs an initialised torch.distributed process group")
    rank, world = dist.get_rank(), dist.get_world_size()
    if device_index is not None:
        nat.check(nat.lib().dcrx_set_device(int(device_index)))

    keys_holder = {}

    def reduce_counts(counts):
        # the Counter's keys differ between ranks (a key appears with its first increment): agree on the union first
        mine = sorted(k for k in counts if k not in ("start_time", "end_time"))
        every = [None] * world
        dist.all_gather_object(every, mine)
        keys = sorted(set(k for ks in every for k in ks))
        keys_holder["keys"] = keys
        t = torch.tensor([int(counts.get(k, 0)) for k in keys], dtype=torch.int64)
        if t.numel():
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            for k, v in zip(keys, t.cpu().tolist()):
                counts[k] = int(v)

    def exchange_error(err):
        # every rank reports whether its part raised; if any did, every rank raises here, before the first collective of
        # the results (a rank that stopped alone would leave the others waiting in all_gather_object for ever)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        if int(flag.item()) == 0:
            return
        said = [None] * world
        dist.all_gather_object(said, None if err is None else f"{type(err).__name__}: {err}")
        if err is not None:
            raise err
        raise RuntimeError("decombinator_sharded: " + "; ".join(f"rank {r}: {m}" for r, m in enumerate(said) if m))

    rows = dec.decombinator(inputargs, shard=(rank, world), reduce_counts=reduce_counts, exchange_error=exchange_error,
                            plan_shards=plan_fastq_shards)
    chunks = rows._tagged_chunks()
    # a rank's message: per chunk (tag, rows, bytes) — three int64 — then the chunks' text
    head = np.array([[t, n, len(b)] for t, b, n in chunks], dtype=np.int64).reshape(-1, 3)
    msg = np.int64(len(chunks)).tobytes() + head.tobytes() + b"".join(b for _, b, _ in chunks)
    parts = gather_bytes(msg, dst=0)
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
