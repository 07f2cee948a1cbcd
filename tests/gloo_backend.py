"""CPU stand-ins for the two interfaces of decombinator_amd/sharded.py (tests only): the communicator and the TupleGather
backend over the gloo backend of torch.distributed on host memory.  The product's own implementations are _native.Comm and
sharded.RcclBackend (RCCL through libdcrx); these let the protocol — count exchange, exact-size transfers, slot reuse, the
sharded stage's control plane — run with world sizes of 2, 3 and 8 on a box without a GPU."""
from __future__ import annotations

import pickle

import numpy as np
import torch
import torch.distributed as dist


class GlooComm:
    """The communicator interface of sharded.py over an initialised gloo process group."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def allgather_host(self, arr) -> np.ndarray:
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        got = [torch.empty_like(t) for _ in range(self.world)]
        if t.numel():
            dist.all_gather(got, t)
        return np.stack([g.numpy().view(a.dtype).reshape(a.shape) for g in got])

    def allreduce_host_u64(self, values, op: int = 0) -> np.ndarray:
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        t = torch.from_numpy(a.view(np.int64))
        if t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        return a

    def allgather_bytes(self, blob: bytes) -> list:
        sizes = self.allgather_host(np.array([len(blob)], dtype=np.uint64)).reshape(-1)
        kmax = int(sizes.max()) if sizes.size else 0
        if kmax == 0:
            return [b"" for _ in range(self.world)]
        pad = np.zeros(kmax, dtype=np.uint8)
        pad[:len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        got = self.allgather_host(pad)
        return [got[r, :int(sizes[r])].tobytes() for r in range(self.world)]

    def allgather_object(self, obj) -> list:
        return [pickle.loads(b) for b in self.allgather_bytes(pickle.dumps(obj))]

    def gather_bytes(self, blob: bytes, dst: int = 0):
        sizes = [int(x) for x in self.allgather_host(np.array([len(blob)], dtype=np.uint64)).reshape(-1)]
        kmax = max(max(sizes), 1)
        pad = torch.zeros(kmax, dtype=torch.uint8)
        if len(blob):
            pad[:len(blob)] = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy())
        if self.rank == dst:
            got = [torch.empty_like(pad) for _ in range(self.world)]
            dist.gather(pad, got, dst=dst)
            return [g[:n].numpy().tobytes() for g, n in zip(got, sizes)]
        dist.gather(pad, None, dst=dst)
        return None

    def barrier(self, stream=None):
        dist.barrier()

    def close(self):
        dist.destroy_process_group()


class _HostBuf:
    def __init__(self, nbytes: int):
        self.np = np.zeros(max(int(nbytes), 16), dtype=np.uint8)
        self.nbytes = self.np.size

    @property
    def ptr(self):
        return self.np.ctypes.data


class _Works:
    """An 'event' of the host backend: the transfers it stands for (waited for where a stream would wait)."""

    def __init__(self):
        self.works = []

    def synchronize(self):
        for w in self.works:
            w.wait()
        self.works = []


class GlooBackend:
    """The TupleGather backend on host memory: no streams (every ordering call is immediate), gloo for the exchange."""
    cuda = False

    def __init__(self, comm: GlooComm = None):
        self.comm = comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0
        self.side_ptr = None
        self._events = []

    def buffer(self, nbytes: int):
        return _HostBuf(nbytes)

    def host_counts(self, world: int):
        return np.zeros(world, dtype=np.uint64)

    def new_event(self):
        e = _Works()
        self._events.append(e)
        return e

    def side_wait_main(self):
        pass

    def main_wait_side(self):
        for e in self._events:
            e.synchronize()

    def main_wait(self, event):
        event.synchronize()

    def side_wait(self, event):
        event.synchronize()

    def zero(self, buf, offset: int, nbytes: int):
        buf.np[offset:offset + nbytes] = 0

    def exchange_counts(self, n_buf, counts_buf, host, event):
        mine = torch.from_numpy(n_buf.np[:8].view(np.int64).copy())
        if self.world > 1:
            got = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(got, mine)
            host[:] = np.array([int(g.item()) for g in got], dtype=np.uint64)
        else:
            host[0] = int(mine.item())
        return event

    def post(self, own_msg, peer_msgs, nbytes, event):
        if self.world > 1:
            if self.rank == 0:
                ops = [dist.P2POp(dist.irecv, torch.from_numpy(peer_msgs[r].np[:nbytes[r]]), r) for r in range(1, self.world)]
                if ops:
                    event.works.extend(dist.batch_isend_irecv(ops))
            else:
                event.works.append(dist.isend(torch.from_numpy(own_msg.np[:nbytes[self.rank]]), dst=0))
        return event

    def to_host(self, buf, nbytes: int) -> np.ndarray:
        return buf.np[:nbytes].copy()

    def synchronize(self):
        pass
