#!/usr/bin/env python3
"""bench.py — Mreads/s decombined on synthetic 150 bp human-beta reads.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 10 M x 150 bp synthetic human-beta reads
against the original-like synthetic tag set, generated ON the GPU from
(seed, read index) so that the timed region starts with the packed reads
resident in HBM.  A step is one pass of the hot path (dcrx_decombine_device:
DFA scan + rescue + walks + filters -> 16-byte records + counters) over the
rank's 10 M-read batch; with N > 1 every rank takes its own 10 M reads (weak
scaling), compacts its DCR tuples and the tuples are gathered on rank 0 over
RCCL inside the same step.

Rank 0 prints ONE JSON line with the whole-job rate, the HBM roofline of the
dominant kernel (algorithmic 54 B/read divided by that kernel's launch time,
measured with HIP events around the launch on its stream) and a CPU baseline:
the oracle (a C port of the reference's algorithm) timed on a bounded sample of
the same reads on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

READS_PER_GPU = 10_000_000
READ_LEN = 150
SEED = 2
ALGO_BYTES_PER_READ = 54      # 38 B packed 150-mer (rounded up) + 16 B record: SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0         # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
TUPLE_BYTES = 16 + 8          # gathered per decombined read: record + global read index


def cpu_baseline(nat, tables, ts, cfg_synth, sample_reads: int):
    """The oracle on `sample_reads` of the very same reads: one thread, then all cores."""
    import numpy as np
    from oracle import oracle as orc

    vs, js = ts.half_splits
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    hb = nat.synth_reads_host(tables, cfg_synth, 0, sample_reads)
    buf, offsets = nat.unpack_reads_raw(hb)
    t0 = time.perf_counter()
    n1 = min(sample_reads, 4_000_000)
    ot.decombine_batch(buf, offsets[:n1 + 1])
    t1 = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    chunks = np.array_split(np.arange(sample_reads), cores)
    out = [None] * cores

    def work(i):
        lo, hi = int(chunks[i][0]), int(chunks[i][-1]) + 1
        out[i] = ot.decombine_batch(buf, offsets[lo:hi + 1])

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    tn = time.perf_counter() - t0
    return {
        "value": round(sample_reads / tn / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
        "sample": f"first {sample_reads} reads of the same synthetic workload, oracle/dcr_oracle.c "
                  f"(C port of the reference's Python path), {cores} threads",
        "value_1thread": round(n1 / t1 / 1e6, 4), "sample_1thread": f"first {n1} reads, 1 thread",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg-flags", type=int, default=0, help="profiling only: DCRX_F_* bits (results are then not checked)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs the torch.distributed.run launcher (one process per GPU)")
        args.gpus = world

    # torch first: its bundled HIP runtime (same soname as /opt/rocm's) must be the one
    # libdcrx binds to, so that torch/RCCL and the kernels share one runtime.
    import torch
    import torch.distributed as dist
    import numpy as np

    from decombinator_amd import _native as nat
    from decombinator_amd import synth
    from decombinator_amd import sharded

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the decombine hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    nat.check(nat.lib().dcrx_set_device(local_rank))
    # DCRX_BENCH_FORCE_GATHER=1 under torchrun with one rank exercises the RCCL path on one GPU
    use_dist = world > 1 or (os.environ.get("DCRX_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    tables = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    info = tables.info()
    n = args.reads
    stride = nat.stride_for(READ_LEN)
    cfg_synth = nat.synth_cfg(seed=SEED, read_len=READ_LEN)
    if os.environ.get("DCRX_BENCH_N_RATE"):          # experiments only: share of reads with an N (default 0.0005)
        cfg_synth = nat.synth_cfg(seed=SEED, read_len=READ_LEN, n_rate=float(os.environ["DCRX_BENCH_N_RATE"]))
    first = rank * n

    # inputs resident in HBM before the timed region
    d_packed = torch.empty(n * stride + 16, dtype=torch.uint8, device=dev)
    nat.check(nat.lib().dcrx_synth_reads_device(tables.handle, nat.C.byref(cfg_synth), first, n, stride,
                                                d_packed.data_ptr(), sptr))
    er, ep, ec = nat.synth_exceptions_host(tables, cfg_synth, first, n)
    d_er = torch.from_numpy(er.astype(np.int64)).to(dev).to(torch.int32)  # same bits as uint32
    d_ep = torch.from_numpy(ep.astype(np.int32)).to(dev).to(torch.int16)
    d_ec = torch.from_numpy(ec).to(dev)
    d_rec = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nat.N_COUNTERS, dtype=torch.int64, device=dev)
    batch = nat.BatchC()
    batch.n_reads, batch.packed, batch.stride, batch.read_len, batch.lens = n, d_packed.data_ptr(), stride, READ_LEN, None
    batch.n_exc = len(er)
    batch.exc_read, batch.exc_pos, batch.exc_chr = (d_er.data_ptr(), d_ep.data_ptr(), d_ec.data_ptr()) if len(er) else (None, None, None)
    cfg = nat.make_cfg("reverse", False, 130, args.cfg_flags)
    nat.check(nat.lib().dcrx_reserve_device(tables.handle, n))
    gather = sharded.TupleGather(n, world, rank, dev) if use_dist else None
    if world > 1:
        # the persistent scan kernels would fill every compute unit; a few are left to RCCL so that the
        # tuples of step k really move beside the scan of step k+1 (on one GPU, where the "gather" is a
        # local copy, reserving units only costs: 0.85 ms/step with none, 0.89 with 16)
        nat.check(nat.lib().dcrx_set_reserved_cus(tables.handle, int(os.environ.get("DCRX_BENCH_RESERVED_CUS", "16"))))

    def step(ev_pair=None):
        rec = d_rec
        if gather is not None:           # alternating record buffers: the previous step's tuples are still being compacted
            gather.before_scan()
            rec = gather.records()
        if ev_pair is not None:
            nat.check(nat.lib().dcrx_set_timing_events(tables.handle, ev_pair[0].ptr, ev_pair[1].ptr))
        nat.check(nat.lib().dcrx_decombine_device(tables.handle, nat.C.byref(cfg), nat.C.byref(batch),
                                                  rec.data_ptr(), d_cnt.data_ptr(), sptr))
        if ev_pair is not None:
            nat.check(nat.lib().dcrx_set_timing_events(tables.handle, None, None))
        if gather is not None:
            gather.step(n)

    def fence():
        if gather is not None:
            gather.finish()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    events = [(nat.Event(), nat.Event()) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = [a.elapsed_ms(b) for a, b in events]
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    counters = d_cnt.cpu().numpy().astype(np.uint64)
    n_hits = int(counters[nat.COUNTER_NAMES.index("vj_count")])
    assert args.cfg_flags or int(counters[nat.COUNTER_NAMES.index("read_count")]) == n
    if gather is not None:
        gather.check(n_hits)

    if rank == 0:
        total_reads = n * world * args.steps
        value = total_reads / elapsed / 1e6
        achieved = ALGO_BYTES_PER_READ * n / (kern_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("reads_per_launch") == n and tj.get("read_len") == READ_LEN:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mreads/s decombined (150 bp human-beta)",
            "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: synthetic 10M x 150bp human-beta reads, original-like synthetic "
                            "tag set (60 V / 13 J), 45% rearranged, 0.5% substitutions, orientation reverse",
                "reads_per_gpu_per_step": n, "read_len": READ_LEN, "seed": SEED,
                "tagset": "synthetic human_original_TRB (real tag files are not available offline)",
                "dfa_states": info["n_states"], "dfa_bytes_in_lds": info["dfa_bytes"],
                "decombined_fraction": round(n_hits / n, 4),
                "parallelism": f"reads sharded x{world}, RCCL gather of DCR tuples to rank 0" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": "dcrx::decombine_kernel", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ,
                "kernel_ms_avg": round(kern_avg_ms, 5), "kernel_ms_min": round(min(kern_ms), 5),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(nat, tables, ts, cfg_synth, min(args.cpu_sample, n))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
