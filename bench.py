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
hot path (algorithmic 54 B/read divided by the device time of ALL launches of a
step, measured with HIP events around them on their stream; the dominant
kernel's own launch time rides along) and a CPU baseline: the oracle (a C port
of the reference's algorithm) timed on a bounded sample of the same reads on the
host cores, one thread and all of them, threaded inside the C library.

--config 3 / 5 run the other single-GPU workloads of BASELINE.json (human
alpha+beta on extended-like tag sets, both chains per step; mouse gamma+delta);
the default, and the line the driver records, is config 2.
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
    """The oracle on `sample_reads` of the very same reads: one thread, then every host core
    (POSIX threads inside the C library; each thread repeats its slice until it has about a
    second of work, so that thread start-up does not show)."""
    from oracle import oracle as orc

    vs, js = ts.half_splits
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    hb = nat.synth_reads_host(tables, cfg_synth, 0, sample_reads)
    buf, offsets = nat.unpack_reads_raw(hb)
    n1 = min(sample_reads, 1_000_000)
    ot.decombine_batch_mt(buf, offsets[:min(n1, 100_000) + 1], n_threads=1)          # warm the pages
    t0 = time.perf_counter()
    ot.decombine_batch_mt(buf, offsets[:n1 + 1], n_threads=1)
    t1 = time.perf_counter() - t0
    rate1 = n1 / t1
    # the cores this process may run on (a container's share of the host), not every core of the box
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    passes = max(1, int(round(rate1 * 1.0 / max(1, sample_reads // cores))))      # ~1 s of work per thread
    passes = min(passes, 64)
    t0 = time.perf_counter()
    ot.decombine_batch_mt(buf, offsets, n_threads=cores, passes=passes)
    tn = time.perf_counter() - t0
    return {
        "value": round(sample_reads * passes / tn / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
        "sample": f"first {sample_reads} reads of the same synthetic workload x {passes} passes, oracle/dcr_oracle.c "
                  f"(C port of the reference's Python path), {cores} POSIX threads (cores this process may use; the box reports "
                  f"{os.cpu_count()}), {tn:.2f} s",
        "value_1thread": round(rate1 / 1e6, 4), "sample_1thread": f"first {n1} reads, 1 thread, {t1:.2f} s",
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
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 5),
                    help="BASELINE.json workload: 2 (default; the recorded metric), 3 = alpha+beta extended-like sets, 5 = mouse gamma+delta")
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

    # the chains a step resolves: one table set each (the reference resolves one chain per call, decombine.py:593-661)
    if args.config == 2:
        tagsets = [synth.config_tagset(2)]
    elif args.config == 3:
        tagsets = list(synth.config3_tagsets())
    else:
        tagsets = list(synth.config5_tagsets())
    all_tables = [nat.Tables(x.v_tags, x.v_jumps, x.v_regions, x.j_tags, x.j_jumps, x.j_regions, *x.half_splits) for x in tagsets]
    ts, tables = tagsets[-1], all_tables[-1]         # the reads are drawn from the last chain's germlines
    info = tables.info()
    algo_bytes = 38 + 16 * len(all_tables)           # 54 B/read one chain, 70 B/read two
    n = args.reads
    stride = nat.stride_for(READ_LEN)
    cfg_synth = nat.synth_cfg(seed={2: SEED, 3: 3, 5: 5}[args.config], read_len=READ_LEN, sub_rate=0.02 if args.config == 5 else 0.005)
    if os.environ.get("DCRX_BENCH_N_RATE"):          # experiments only: share of reads with an N (default 0.0005)
        cfg_synth = nat.synth_cfg(seed={2: SEED, 3: 3, 5: 5}[args.config], read_len=READ_LEN, sub_rate=0.02 if args.config == 5 else 0.005,
                                  n_rate=float(os.environ["DCRX_BENCH_N_RATE"]))
    first = rank * n

    # inputs resident in HBM before the timed region
    # (several chains: the batch is drawn in equal parts from each chain's germlines, part k from chain k)
    d_packed = torch.empty(n * stride + 16, dtype=torch.uint8, device=dev)
    ers, eps, ecs = [], [], []
    for k, tb in enumerate(all_tables):
        lo, hi = n * k // len(all_tables), n * (k + 1) // len(all_tables)
        nat.check(nat.lib().dcrx_synth_reads_device(tb.handle, nat.C.byref(cfg_synth), first + lo, hi - lo, stride,
                                                    d_packed.data_ptr() + lo * stride, sptr))
        e_r, e_p, e_c = nat.synth_exceptions_host(tb, cfg_synth, first + lo, hi - lo)
        ers.append(e_r.astype(np.int64) + lo); eps.append(e_p); ecs.append(e_c)
    er, ep, ec = np.concatenate(ers), np.concatenate(eps), np.concatenate(ecs)
    d_er = torch.from_numpy(er.astype(np.int64)).to(dev).to(torch.int32)  # same bits as uint32
    d_ep = torch.from_numpy(ep.astype(np.int32)).to(dev).to(torch.int16)
    d_ec = torch.from_numpy(ec).to(dev)
    d_recs = [torch.empty(n * 16, dtype=torch.uint8, device=dev) for _ in all_tables]      # one record plane per chain
    d_cnts = [torch.zeros(nat.N_COUNTERS, dtype=torch.int64, device=dev) for _ in all_tables]
    d_rec, d_cnt = d_recs[-1], d_cnts[-1]
    batch = nat.BatchC()
    batch.n_reads, batch.packed, batch.stride, batch.read_len, batch.lens = n, d_packed.data_ptr(), stride, READ_LEN, None
    batch.n_exc = len(er)
    batch.exc_read, batch.exc_pos, batch.exc_chr = (d_er.data_ptr(), d_ep.data_ptr(), d_ec.data_ptr()) if len(er) else (None, None, None)
    cfg = nat.make_cfg("reverse", False, 130, args.cfg_flags)
    for tb in all_tables:
        nat.check(nat.lib().dcrx_reserve_device(tb.handle, n))
    gather = sharded.TupleGather(n, world, rank, dev) if use_dist else None
    if world > 1:
        # the persistent scan kernels would fill every compute unit; a few are left to RCCL so that the
        # tuples of step k really move beside the scan of step k+1 (on one GPU, where the "gather" is a
        # local copy, reserving units only costs: 0.85 ms/step with none, 0.89 with 16)
        nat.check(nat.lib().dcrx_set_reserved_cus(tables.handle, int(os.environ.get("DCRX_BENCH_RESERVED_CUS", "16"))))

    def step(ev=None):
        rec = d_rec
        if gather is not None:           # alternating record buffers: the previous step's tuples are still being compacted
            gather.before_scan()
            rec = gather.records()
        for k, tb in enumerate(all_tables):
            last = k == len(all_tables) - 1
            if ev is not None:           # events: (step start, step stop, kernel start, kernel stop) per chain
                nat.check(nat.lib().dcrx_set_step_events(tb.handle, ev[k][0].ptr, ev[k][1].ptr))
                nat.check(nat.lib().dcrx_set_timing_events(tb.handle, ev[k][2].ptr, ev[k][3].ptr))
            nat.check(nat.lib().dcrx_decombine_device(tb.handle, nat.C.byref(cfg), nat.C.byref(batch),
                                                      (rec if last else d_recs[k]).data_ptr(), d_cnts[k].data_ptr(), sptr))
            if ev is not None:
                nat.check(nat.lib().dcrx_set_step_events(tb.handle, None, None))
                nat.check(nat.lib().dcrx_set_timing_events(tb.handle, None, None))
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
    events = [[tuple(nat.Event() for _ in range(4)) for _ in all_tables] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    # HIP events around the kernels are not free (a step that carries its four costs ~20 us more: 0.508 against 0.485 ms,
    # whether recorded separately or riding on the dispatches): every EVENT_EVERY-th step of the timed region carries
    # them, and the device-side averages below are over those steps
    every = max(1, int(os.environ.get("DCRX_BENCH_EVENT_EVERY", "5")))
    timed = [k for k in range(args.steps) if k % every == every - 1 or args.steps < every]
    for k in range(args.steps):
        step(events[k] if k in timed else None)
    fence()
    elapsed = time.perf_counter() - t0
    events = [events[k] for k in timed]

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    step_ms = [sum(e[0].elapsed_ms(e[1]) for e in evs) for evs in events]         # all launches of a step, on the device
    kern_ms = [sum(e[2].elapsed_ms(e[3]) for e in evs) for evs in events]         # the dominant kernel(s) alone
    step_avg_ms, kern_avg_ms = sum(step_ms) / len(step_ms), sum(kern_ms) / len(kern_ms)
    counters = d_cnt.cpu().numpy().astype(np.uint64)
    n_hits = sum(int(c.cpu().numpy().astype(np.uint64)[nat.COUNTER_NAMES.index("vj_count")]) for c in d_cnts)
    assert args.cfg_flags or all(int(c.cpu().numpy().astype(np.uint64)[nat.COUNTER_NAMES.index("read_count")]) == n for c in d_cnts)
    if gather is not None:
        gather.check(int(counters[nat.COUNTER_NAMES.index("vj_count")]))

    if rank == 0:
        total_reads = n * world * args.steps
        value = total_reads / elapsed / 1e6
        achieved = algo_bytes * n / (step_avg_ms * 1e-3) / 1e9
        v2 = bool(info.get("v2_tables")) and not (args.cfg_flags & 64)
        dominant = "dcrx::scan2_kernel" if v2 else "dcrx::decombine_kernel"
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == 2:
            try:
                tj = json.load(open(tpath))
                if tj.get("reads_per_launch") == n and tj.get("read_len") == READ_LEN and tj.get("kernels") == ("v2" if v2 else "v1"):
                    traffic = tj.get("hbm_bytes_per_step")
                    traffic_source = "profiles/traffic.json: rocprofv3 --pmc passes of this command on an earlier run (" + tj.get("profile", "?") + "), not measured by this run"
            except Exception:
                traffic = None
        line = {
            "metric": "Mreads/s decombined (150 bp human-beta)" if args.config == 2 else f"Mreads/s decombined (150 bp, BASELINE config {args.config})",
            "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": {2: "BASELINE configs[1]: synthetic 10M x 150bp human-beta reads, original-like synthetic "
                                "tag set (60 V / 13 J), 45% rearranged, 0.5% substitutions, orientation reverse",
                             3: "BASELINE configs[2]: synthetic 150bp reads, human alpha + beta on extended-like synthetic tag sets "
                                "(104 V / 61 J and 88 V / 14 J), both chains resolved per step (two passes over the resident reads), "
                                f"{n} reads per step",
                             5: "BASELINE configs[4]: synthetic 150bp reads, mouse gamma + delta original-like synthetic tag sets, "
                                f"2% substitutions (half-tag rescue path), both chains per step, {n} reads per step"}[args.config],
                "reads_per_gpu_per_step": n, "read_len": READ_LEN, "seed": SEED,
                "tagset": "synthetic " + " + ".join(x.file_stem("v")[:-1] for x in tagsets) + " (real tag files are not available offline)",
                "kernels": "v2 (scan2 / rescue2 + tail2 / events2)" if v2 else "three-launch form",
                "dfa_states": info["n_states"], "dfa_bytes_in_lds": info.get("v2_scan_bytes") if v2 else info["dfa_bytes"],
                "decombined_fraction": round(n_hits / n, 4),
                "parallelism": f"reads sharded x{world}, RCCL gather of DCR tuples to rank 0" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": "all launches of a step (dcrx_decombine_device: prologue, scan, finishing kernels)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_read": algo_bytes,
                "step_device_ms_avg": round(step_avg_ms, 5), "step_device_ms_min": round(min(step_ms), 5),
                "dominant_kernel": dominant, "dominant_kernel_ms_avg": round(kern_avg_ms, 5),
                "dominant_kernel_ms_min": round(min(kern_ms), 5),
                "events": f"HIP events on {len(events)} of the {args.steps} timed steps (every {every}th: a step that carries them runs ~4 % "
                          "longer, so the device-side averages can exceed ms_per_step)",
            },
            # the same algorithmic bytes over the host-side time of a step (launch gaps included)
            "step_frac": round(algo_bytes * n * world / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 5),
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
