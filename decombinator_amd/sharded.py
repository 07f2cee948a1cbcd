"""Multi-GPU sharding of the decombine hot path: one process per GPU, reads split
into contiguous ranges by rank, a final gather of the DCR tuples on rank 0 and a
sum of the counters (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).

The reference is single-process: nothing here mirrors reference
code.  Each read's result depends only on that read and the replicated tables
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


class TupleGather:
    """Per-step fixed-capacity gather for the benchmark loop: no host sync inside the timed
    region.  Every rank compacts its step's tuples on the device — each decombined record squeezed
    into 12 bytes, in read order, plus one bit per read saying which reads they belong to (half the
    bytes of 16-byte records with 8-byte indices) — and sends the first `cap` records (cap = reads/2 covers the synthetic
    mixture's ~42 % decombined reads; `check` verifies that after the run) and the bitmap.

    Compaction and gather run on a side stream, beside the scan of the following step: the caller
    alternates between `depth` record buffers (`records(k)`), the scan of step k+depth waits for
    the compaction of step k (`before_scan`), and a buffer set is reused only after its previous
    gather has completed."""

    TUPLE_BYTES = 12     # dcrx_compact_hits_packed_device: the record's fields in three uint32

    def __init__(self, n_reads: int, world: int, rank: int, device: torch.device, cap_fraction: float = 0.5,
                 depth: int = 2):
        from . import _native as nat
        self.nat = nat
        self.world, self.rank = world, rank
        self.cap = int(n_reads * cap_fraction) + 1024
        self.words = (n_reads + 63) // 64
        self.k = 0
        self.side = torch.cuda.Stream(device=device)
        self.slots = []
        for _ in range(depth):
            slot = {
                "rec": torch.empty(n_reads * 16, dtype=torch.uint8, device=device),
                "hits": torch.empty(n_reads * self.TUPLE_BYTES, dtype=torch.uint8, device=device),
                "bitmap": torch.zeros(self.words, dtype=torch.int64, device=device),
                "n": torch.zeros(1, dtype=torch.int64, device=device),
                "work": [],
                "compacted": None,     # event: the compaction that read this slot's records is done
            }
            if rank == 0:
                slot["g_hits"] = [torch.empty(self.cap * self.TUPLE_BYTES, dtype=torch.uint8, device=device) for _ in range(world)]
                slot["g_bitmap"] = [torch.empty(self.words, dtype=torch.int64, device=device) for _ in range(world)]
                slot["g_n"] = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
            self.slots.append(slot)

    def records(self) -> torch.Tensor:
        """The record buffer the next scan writes (call before_scan() first)."""
        return self.slots[self.k % len(self.slots)]["rec"]

    def before_scan(self) -> None:
        """The current stream waits until the slot's previous compaction has read its records."""
        ev = self.slots[self.k % len(self.slots)]["compacted"]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def step(self, n_reads: int) -> None:
        """After the scan of this step has been queued on the current stream."""
        nat = self.nat
        s = self.slots[self.k % len(self.slots)]
        self.k += 1
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            for w in s["work"]:          # this set's previous gather must be done before its buffers are rewritten
                w.wait()
            nat.check(nat.lib().dcrx_compact_hits_packed_device(s["rec"].data_ptr(), n_reads, s["hits"].data_ptr(),
                                                                s["bitmap"].data_ptr(), s["n"].data_ptr(),
                                                                self.side.cuda_stream))
            s["compacted"] = self.side.record_event()
            h = s["hits"][:self.cap * self.TUPLE_BYTES]
            if self.rank == 0:
                s["work"] = [dist.gather(s["n"], s["g_n"], dst=0, async_op=True),
                             dist.gather(h, s["g_hits"], dst=0, async_op=True),
                             dist.gather(s["bitmap"], s["g_bitmap"], dst=0, async_op=True)]
            else:
                s["work"] = [dist.gather(s["n"], None, dst=0, async_op=True),
                             dist.gather(h, None, dst=0, async_op=True),
                             dist.gather(s["bitmap"], None, dst=0, async_op=True)]

    def finish(self) -> None:
        """Makes the current stream wait for every compaction and gather still in flight."""
        with torch.cuda.stream(self.side):
            for s in self.slots:
                for w in s["work"]:
                    w.wait()
                s["work"] = []
        torch.cuda.current_stream().wait_stream(self.side)

    @staticmethod
    def _popcount(words: torch.Tensor) -> int:
        b = words.view(torch.uint8).to(torch.int32)
        total = 0
        for k in range(8):
            total += int(((b >> k) & 1).sum().item())
        return total

    def check(self, n_hits_local: int) -> None:
        self.finish()
        torch.cuda.synchronize()
        if n_hits_local > self.cap:
            raise RuntimeError(f"rank {self.rank}: {n_hits_local} tuples exceed the gather capacity {self.cap}")
        if self.rank == 0:
            last = self.slots[(self.k - 1) % len(self.slots)]
            got = [int(x.item()) for x in last["g_n"]]
            if got[0] != n_hits_local or any(g <= 0 or g > self.cap for g in got):
                raise RuntimeError(f"gathered tuple counts look wrong: {got}")
            # every rank's bitmap marks exactly as many reads as it sent tuples (rank order = read order)
            for r in range(self.world):
                if self._popcount(last["g_bitmap"][r]) != got[r]:
                    raise RuntimeError(f"bitmap of rank {r} does not match its {got[r]} tuples")
