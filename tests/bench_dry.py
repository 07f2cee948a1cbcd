#!/usr/bin/env python3
"""The bench's rank program on CPU ranks (tests only): bench.run_rank() with tests/dry_device.py in place of the GPU and the
gloo backend in place of RCCL.  It exercises what `python bench.py --gpus N` does before and around the device — the spawn
of the ranks, the sharding, the step loop, the TupleGather protocol, the JSON line — and can never yield a rate: the line
says "dry_run": true and `value` is null.  bench.py itself has no such switch (the measurement tool cannot produce a line
from the checker)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    argv = sys.argv[1:]
    args = bench.parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(bench.spawn_ranks(args, argv, script=os.path.abspath(__file__)))
    from tests import dry_device

    def comm_factory(use_dist):      # gloo on host memory in place of RCCL on device memory (tests/gloo_backend.py)
        if not use_dist:
            return None, None
        import torch.distributed as dist
        from tests.gloo_backend import GlooBackend, GlooComm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")))
        comm = GlooComm()
        return comm, GlooBackend(comm)
    bench.run_rank(args, device_factory=dry_device.DryDevice, comm_factory=comm_factory)


if __name__ == "__main__":
    main()
