#!/usr/bin/env python3
"""bench.py — Mreads/s decombined on synthetic 150 bp human-beta reads.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 10 M x 150 bp synthetic human-beta reads
against the original-like synthetic tag set, generated ON the GPU from
(seed, read index) so that the timed region starts with the packed reads
resident in HBM.  A step is one pass of the hot path (dcrx_decombine_device:
automaton scan + half-tag rescue + walks + filters -> 16-byte records + counters)
over the rank's 10 M-read batch; with N > 1 every rank takes its own 10 M reads
(weak scaling), leaves its DCR tuples in a message and the messages are gathered on rank 0 over
RCCL inside the same step — through libdcrx's own RCCL binding (include/dcrx.h dcrx_comm_*): this
file imports no torch; the ranks find each other through a file of the node's temporary directory
(decombinator_amd/_native.py comm_from_env).

`--gpus N` without a launcher (WORLD_SIZE unset): this process — before it
touches HIP — starts N child processes of this same file, one
per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in
their environment), waits for them, relays rank 0's JSON line and exits non-zero
if any of them did.  Under torch.distributed.run the ranks are the launcher's.

Rank 0 prints ONE JSON line with the whole-job rate, the HBM roofline of the
hot path (algorithmic 54 B/read divided by the device time of ALL launches of a
step, measured with HIP events around them on their stream; the dominant
kernel's own launch time rides along) and a CPU baseline: the oracle (a C port
of the reference's algorithm) timed on a bounded sample of the same reads on the
host cores, threaded inside the C library, at several thread counts.

--config 3 / 5 run the other single-GPU workloads of BASELINE.json (human
alpha+beta on extended-like tag sets, both chains per step; mouse gamma+delta);
--config 4 is BASELINE configs[3]: a FIXED total of 1 B reads (--total-reads)
sharded over the ranks (sharded.shard_range), every rank working through its
shard in 10 M-read steps — the strong-scaling curve at 1/2/4/8 GPUs.  The
default, and the line the driver records, is config 2.

The CPU test of the spawn path, the sharding and the tuple-gather protocol lives
in tests/bench_dry.py (it hands run_rank() a stand-in device and the gloo backend);
nothing in this file can produce a line without a GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

READS_PER_GPU = 10_000_000
READ_LEN = 150
SEED = 2
ALGO_BYTES_PER_READ = 54      # 38 B packed 150-mer (rounded up) + 16 B record: SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0         # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
CONFIG_SEED = {2: SEED, 3: 3, 4: 4, 5: 5}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg-flags", type=int, default=0, help="profiling only: DCRX_F_* bits (results are then not checked)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json workload: 2 (default; the recorded metric), 3 = alpha+beta extended-like sets, "
                         "4 = a fixed total of reads sharded over the ranks (strong scaling), 5 = mouse gamma+delta")
    ap.add_argument("--total-reads", type=int, default=10**9, help="config 4: reads of the whole job")
    ap.add_argument("--in-flight", type=int, default=0, choices=(0, 1, 2),
                    help="batches in flight: 2 = consecutive steps alternate between two handles (streams, record planes, counter blocks) "
                         "so that a step's finishing launches run beside the next step's scan; 1 = one stream, each step behind the last; "
                         "0 (default) = 2 for config 2 on one rank without a gather, else 1 (configs 3 and 5 overlap their two chains instead: a stream per chain)")
    ap.add_argument("--no-gather-ab", action="store_true", help="N > 1: skip the second loop that prices the exposed gather time")
    return ap.parse_args(argv)


def spawn_ranks(args, argv, script=None) -> int:
    """The parent of a plain `bench.py --gpus N`: N children of this file (or of `script`: tests/bench_dry.py), one per
    GPU.  Nothing here imports torch or initialises HIP (a process that has must not exec another program on this pool,
    and need not: the children are fresh processes)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))
    out0 = ""
    failed = None
    # rank 0's stdout is read to its end (the JSON line), then every child is waited for; a child that fails takes the
    # others down (exact PIDs), so that a half-started job cannot hang on a collective
    import threading
    buf = []
    t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    t.start()
    live = set(range(args.gpus))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and failed is None:
                failed = (r, rc)
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    t.join(timeout=10)
    out0 = buf[0] if buf else ""
    sys.stdout.write(out0)
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        return failed[1] if failed[1] > 0 else 1
    return 0


# (what the decombine call never runs: the FASTQ reader, row assembly, the collapse front, CDR3 translation)
HOST_ONLY_SOURCES = ("dcrx_fastq.cpp", "dcrx_rows.cpp", "dcrx_collapse.cpp", "dcrx_translate.cpp")


def csrc_digest() -> str:
    """sha256 over the sources of the library that the decombine call runs — kernels, device headers, tables, launch code
    (tools/prof_summary_r05.py holds the same function)."""
    import hashlib
    d = os.path.join(ROOT, "decombinator_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) and name not in HOST_ONLY_SOURCES:
            h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def host_cpu_report() -> dict:
    """What the CPU baseline ran on: the cores this process may use, the cgroup's CPU quota, the box's count."""
    rep = {"os_cpu_count": os.cpu_count()}
    try:
        rep["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except Exception:
        rep["sched_getaffinity"] = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            rep["cgroup_" + os.path.basename(path)] = open(path).read().strip()
        except Exception:
            pass
    try:
        rep["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except Exception:
        pass
    return rep


def usable_cores() -> int:
    """Cores this process can really use: the cgroup's CPU quota (cpu.max `quota period`, or the v1 pair) when there is
    one, capped by the affinity mask — not the thread count that happened to win."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = min(cores, max(1, int(round(quota / period))))
        except Exception:
            pass
    return cores


# The reference's own path — the unmodified decombine.py loop `dcr(revcomp(read))` (:998-1001) with the stand-in matcher for
# acora — cannot run on the GPU box (the reference does not travel); it was timed in the build container on 100 k of the same
# reads by tools/ref_python_loop.py and rides along as a constant, labelled as such.
REFERENCE_PYTHON = {
    "value": 0.0084, "unit": "Mreads/s", "cores": 1, "where": "build container, 1 core, stand-in matcher for acora",
    "script": "tools/ref_python_loop.py", "sample": "first 100000 reads of the same synthetic workload",
    "measured_by_this_run": False,
}


def cpu_baseline(nat, tables, ts, cfg_synth, sample_reads: int):
    """The oracle on `sample_reads` of the very same reads (POSIX threads inside the C library; each thread owns a
    contiguous slice and repeats it until it has about a second of work, so that thread start-up does not show):
    one thread, then 8 / 32 / 128 / every core this process may use — `value` is the best of them."""
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
    ladder = sorted(set(k for k in (8, 32, 128, cores) if k <= cores))
    by_threads = {1: round(rate1 / 1e6, 4)}
    best, best_k, best_t, best_passes = rate1, 1, t1, 1
    for k in ladder:
        passes = max(1, int(round(rate1 * 1.0 / max(1, sample_reads // k))))          # ~1 s of work per thread
        passes = min(passes, 64)
        t0 = time.perf_counter()
        ot.decombine_batch_mt(buf, offsets, n_threads=k, passes=passes)
        tn = time.perf_counter() - t0
        rate = sample_reads * passes / tn
        by_threads[k] = round(rate / 1e6, 4)
        if rate > best:
            best, best_k, best_t, best_passes = rate, k, tn, passes
    return {
        "value": round(best / 1e6, 4), "unit": "Mreads/s", "cores": usable_cores(), "threads": best_k, "kind": "port",
        "sample": f"first {sample_reads} reads of the same synthetic workload x {best_passes} passes, oracle/dcr_oracle.c "
                  f"(C port of the reference's Python path), {best_k} POSIX threads (the fastest of the thread counts tried) on "
                  f"{usable_cores()} usable cores (cgroup quota / affinity), {best_t:.2f} s",
        "reference_python": REFERENCE_PYTHON,
        "value_1thread": round(rate1 / 1e6, 4), "sample_1thread": f"first {n1} reads, 1 thread, {t1:.2f} s",
        "mreads_per_s_by_threads": by_threads, "host": host_cpu_report(),
    }


def run_rank(args, device_factory=None, comm_factory=None):
    """One rank.  `device_factory` / `comm_factory` are given only by tests/bench_dry.py: a CPU stand-in for the device and a
    gloo communicator + host-memory backend (the line then says "dry_run": true and carries no rate); the command line of this
    file cannot set them."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    dry = device_factory is not None

    import numpy as np

    from decombinator_amd import _native as nat
    from decombinator_amd import synth
    from decombinator_amd import sharded

    if not dry:
        if nat.device_count() < 1:
            sys.exit("bench.py needs a GPU: the decombine hot path has no CPU fallback")
        nat.check(nat.lib().dcrx_set_device(local_rank))
    # DCRX_BENCH_FORCE_GATHER=1 with one rank exercises the gather path on one GPU
    use_dist = world > 1 or (os.environ.get("DCRX_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ)
    comm, backend = None, None
    main_stream = None if dry else nat.Stream()      # the stream of the decombine calls (one handle = one stream)
    sptr = None if dry else main_stream.ptr
    if dry:
        comm, backend = comm_factory(use_dist)
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        comm = nat.comm_from_env()                   # RCCL through libdcrx; rank 0's id travels through a file of this node
        backend = sharded.RcclBackend(nat, comm, sptr)

    # the chains a step resolves: one table set each (the reference resolves one chain per call, decombine.py:593-661)
    if args.config in (2, 4):
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
    seed = CONFIG_SEED[args.config]
    n_rate = float(os.environ["DCRX_BENCH_N_RATE"]) if os.environ.get("DCRX_BENCH_N_RATE") else 0.0005     # experiments only
    sub_rate = float(os.environ["DCRX_BENCH_SUB_RATE"]) if os.environ.get("DCRX_BENCH_SUB_RATE") else (0.02 if args.config == 5 else 0.005)     # experiments only
    p_rearranged = float(os.environ["DCRX_BENCH_P_REARRANGED"]) if os.environ.get("DCRX_BENCH_P_REARRANGED") else 0.45     # experiments only
    cfg_synth = nat.synth_cfg(seed=seed, read_len=READ_LEN, p_rearranged=p_rearranged, sub_rate=sub_rate, n_rate=n_rate)

    # ---- what a rank works through: `batches` = [(first read index, reads)], one per step of a pass ----
    if args.config == 4:
        lo, hi = sharded.shard_range(args.total_reads, world, rank)
        batches = [(a, min(n, hi - a)) for a in range(lo, hi, n)] or [(lo, 0)]
        glo = [sharded.shard_range(args.total_reads, world, r) for r in range(world)]
        n_steps = max((b - a + n - 1) // n for a, b in glo)          # every rank runs the same number of steps (short shards: empty steps)
        while len(batches) < n_steps:
            batches.append((hi, 0))
        args.steps = n_steps
        job_reads_per_pass = args.total_reads
    else:
        batches = [(rank * n, n)]
        job_reads_per_pass = None
    if dry:
        device = device_factory(nat, all_tables, tagsets, cfg_synth, batches, n)
    else:
        device = HipDevice(nat, np, sptr, all_tables, cfg_synth, batches, n, stride, args.cfg_flags)
        # Two batches in flight (round 6): consecutive steps alternate between two handles of the same tag set — each with its stream,
        # workspace, record plane and counter block, as the library's host entry keeps two chunks in flight (dcrx_decombine) — so that
        # a step's finishing launches, latency-bound and alone on the chip otherwise, run beside the next step's scan.  Where tuples are
        # gathered (several ranks, config 4, DCRX_BENCH_FORCE_GATHER) the gather's own alternating buffers set the order: one in flight.
        plain = not use_dist and os.environ.get("DCRX_BENCH_FORCE_GATHER") != "1" and args.config != 4 and not args.cfg_flags
        if use_dist and device.chain_streams is not None:      # (the gather orders its buffers on the rank's one stream: the chains stay on it)
            device.chain_streams = None
            device.slots = [(device.all_tables, device.d_recs, device.d_cnts, None)]
        in_flight = int(os.environ.get("DCRX_BENCH_BATCHES_IN_FLIGHT", "0")) or args.in_flight or (2 if plain and len(all_tables) == 1 else 1)      # (two chains on a stream each already overlap: a second pair of handles adds nothing, measured)
        for _ in range(in_flight - 1 if plain else 0):
            device.second_slot([nat.Tables(x.v_tags, x.v_jumps, x.v_regions, x.j_tags, x.j_jumps, x.j_regions, *x.half_splits) for x in tagsets])
    # what travels: the narrow tuple of the tag set (5 bytes here) left by the decombine call itself (the handle's tuple sink);
    # A/B (DCRX_BENCH_GATHER_MODE): "narrow" = the same tuples compacted from the records on a side stream, "tuple8" = round 3's
    # 8-byte tuples, compacted
    gmode = os.environ.get("DCRX_BENCH_GATHER_MODE", "sink")
    gather = sharded.TupleGather(n, backend, compact=device.compact, v_jumps=ts.v_jumps,
                                 n_v=info["n_v"], n_j=info["n_j"], tables=tables if gmode in ("sink", "narrow") else None,
                                 max_read_len=READ_LEN, use_sink=gmode == "sink") if use_dist else None
    if use_dist and not dry and (world > 1 or os.environ.get("DCRX_BENCH_RESERVED_CUS")):
        # the persistent scan kernels would fill every compute unit; a few are left to RCCL so that the tuples of step k
        # really move beside the scan of step k+1 (one rank alone — DCRX_BENCH_FORCE_GATHER — sends nothing: none reserved
        # unless asked for)
        reserved = int(os.environ.get("DCRX_BENCH_RESERVED_CUS", "16"))
        for tb in all_tables:
            nat.check(nat.lib().dcrx_set_reserved_cus(tb.handle, reserved))

    def fence(g):
        if g is not None:
            g.finish()
        if use_dist:
            comm.barrier(sptr)
        if not dry:
            nat.synchronize()

    def timed_loop(g, steps, events=None, timed=()):
        fence(g)
        t0 = time.perf_counter()
        for k in range(steps):
            device.step(k, g, events[k] if events is not None and k in timed else None)
        fence(g)
        return time.perf_counter() - t0

    # The chip's clocks come up over the first ~60 steps behind an idle gap (profiles/r05_final/step_time_over_200_steps.log: 0.361 ->
    # 0.334 ms), and the set-up above ends in seconds of host work (the exception lists) with the device idle: the device is kept
    # busy with PREROLL_STEPS untimed steps of the same call (320: a tenth of a second) before the W warm-up steps, so that `--steps 20 --warmup 5` times
    # what `--steps 200` times.  The line says how many (`preroll_steps`); the warm-up and the timed region are untouched.
    first_launch_ms = None
    preroll_steps = int(os.environ.get("DCRX_BENCH_PREROLL_STEPS", "320")) if (not dry and args.warmup > 0) else 0      # (every rank the same count: the steps of a sharded run hold collectives)
    for k in range(preroll_steps):
        if k == 0:      # what a cold handle's first step takes (workspace allocation, code load, untuned launch shape)
            nat.synchronize()
            t_first = time.perf_counter()
        device.step(k, gather, None)
        if k == 0:
            nat.synchronize()
            first_launch_ms = (time.perf_counter() - t_first) * 1e3
        elif k % 32 == 31:
            nat.synchronize()      # (the host does not run hundreds of launches ahead of the device)
    for k in range(args.warmup):
        if k == 0 and not dry and first_launch_ms is None:      # what a cold handle's first step takes (workspace allocation, code load, untuned launch shape)
            nat.synchronize()
            t_first = time.perf_counter()
        device.step(k, gather, None)
        if k == 0 and not dry and first_launch_ms is None:
            nat.synchronize()
            first_launch_ms = (time.perf_counter() - t_first) * 1e3
    fence(gather)
    # HIP events around the kernels are not free (a step that carries its four costs ~20 us more): every
    # EVENT_EVERY-th step of the timed region carries them, and the device-side averages below are over those steps
    every = max(1, int(os.environ.get("DCRX_BENCH_EVENT_EVERY", "5")))
    # (the step's pair rides on the step's first and last dispatch, the dominant kernel's pair on that kernel's dispatch: the scan
    # is also a step's first launch, so the two pairs go on different steps — one dispatch carries one start event)
    timed = [k for k in range(args.steps) if k % every == every - 1 or args.steps < every]
    events = device.make_events(args.steps, timed)
    elapsed = timed_loop(gather, args.steps, events, set(timed))
    # the same steps once more, 200 of them, behind the timed region: the steady-state figure beside the headline (an extra key)
    elapsed_steady = None
    steady_steps = int(os.environ.get("DCRX_BENCH_STEADY_STEPS", "200"))
    if not dry and steady_steps > 0 and args.config != 4:
        elapsed_steady = timed_loop(gather, steady_steps)
    # ... and, where two batches are in flight, as many steps as the timed region's on the first handle alone — each step behind the
    # last, as rounds 1-5 timed them: the figure beside the headline (an extra key)
    elapsed_one = None
    if not dry and len(getattr(device, "slots", [0])) > 1:
        all_slots = device.slots
        device.slots = all_slots[:1]
        timed_loop(None, 5)
        elapsed_one = timed_loop(None, max(args.steps, 50))
        device.slots = all_slots
    elapsed_nogather = None
    if gather is not None and not args.no_gather_ab:
        # the same steps without the gather: the difference is the gather time the steps do not hide
        elapsed_nogather = timed_loop(None, args.steps)

    def max_over_ranks(x):      # (seconds, exchanged as whole nanoseconds)
        if not use_dist:
            return x
        return float(comm.allreduce_host_u64(np.array([int(round(x * 1e9))], dtype=np.uint64), 1)[0]) / 1e9

    # every rank's own time (the job's is the slowest rank's): a first multi-GPU run then shows at a glance which rank lags
    per_rank_ms = [round(elapsed / args.steps * 1e3, 4)]
    if use_dist:
        per_rank_ms = comm.allgather_object(per_rank_ms[0])
    elapsed = max_over_ranks(elapsed)
    if elapsed_steady is not None:
        elapsed_steady = max_over_ranks(elapsed_steady)
    if elapsed_nogather is not None:
        elapsed_nogather = max_over_ranks(elapsed_nogather)
    step_ms, kern_ms = device.event_times(events, timed)
    if os.environ.get("DCRX_BENCH_STEP_TRACE") == "1":      # (does the step time drift over the timed region?  tools/r05_e29.sh)
        print("step_trace", [round(x, 4) for x in step_ms], "kernel", [round(x, 4) for x in kern_ms], file=sys.stderr)
    if os.environ.get("DCRX_BENCH_DUMP_COUNTERS") == "1" and not dry:      # (instrumented builds of the library, tools/: the raw counter block of the last step)
        print("counters", [int(x) for x in device.counters_host(-1)], file=sys.stderr)
    n_hits, n_read = device.totals()
    if not dry:
        assert device.device_errors() == 0, "a device-side wait timed out (include/dcrx_codes.h, DCRX_C_DEVICE_ERRORS)"

    # (DCRX_BENCH_NO_CHECK=1: experiment builds of the library that leave work out, tools/)
    assert args.cfg_flags or os.environ.get("DCRX_BENCH_NO_CHECK") == "1" or n_read == device.expected_read_count(), (n_read, device.expected_read_count())
    if gather is not None and os.environ.get("DCRX_BENCH_NO_GATHER_CHECK") != "1":      # (experiment builds of the library: tools/r04_sink_exp.sh)
        gather.check(device.last_step_hits())
    names = comm.allgather_object(device.name()) if use_dist else [device.name()]
    hits_all = n_hits
    if use_dist:
        hits_all = int(comm.allreduce_host_u64(np.array([n_hits], dtype=np.uint64))[0])

    if rank == 0:
        total_reads = args.total_reads if args.config == 4 else n * world * args.steps
        value = total_reads / elapsed / 1e6
        overlapped = 0 if dry else (len(device.slots) if len(getattr(device, 'slots', [0])) > 1 else (1 if getattr(device, 'chain_streams', None) else 0))      # launches of different steps / chains overlap
        line = {
            "metric": "Mreads/s decombined (150 bp human-beta)" if args.config in (2, 4) else f"Mreads/s decombined (150 bp, BASELINE config {args.config})",
            "value": None if dry else round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_steady": None if elapsed_steady is None else round(elapsed_steady / steady_steps * 1e3, 4),
            "ms_per_step_one_batch_in_flight": None if elapsed_one is None else round(elapsed_one / max(args.steps, 50) * 1e3, 4),
            "preroll_steps": preroll_steps,
            "higher_is_better": True, "scaling": "strong" if args.config == 4 else "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic", "world_size": world, "per_rank_ms_per_step": per_rank_ms,
            "config": {
                "workload": {2: "BASELINE configs[1]: synthetic 10M x 150bp human-beta reads, original-like synthetic "
                                "tag set (60 V / 13 J), 45% rearranged, 0.5% substitutions, orientation reverse",
                             3: "BASELINE configs[2]: synthetic 150bp reads, human alpha + beta on extended-like synthetic tag sets "
                                "(104 V / 61 J and 88 V / 14 J), both chains resolved per step (two passes over the resident reads), "
                                f"{n} reads per step",
                             4: f"BASELINE configs[3]: {args.total_reads} synthetic 150bp human-beta reads in all (fixed), sharded over "
                                f"{world} rank(s) in contiguous ranges (sharded.shard_range), {n} reads per rank and step, the whole shard "
                                "resident in HBM before the timed region, DCR tuples gathered on rank 0",
                             5: "BASELINE configs[4]: synthetic 150bp reads, mouse gamma + delta original-like synthetic tag sets, "
                                f"2% substitutions (half-tag rescue path), both chains per step, {n} reads per step"}[args.config],
                "reads_per_gpu_per_step": n, "read_len": READ_LEN, "seed": seed,
                "tagset": "synthetic " + " + ".join(x.file_stem("v")[:-1] for x in tagsets) + " (real tag files are not available offline)",
                "kernels": device.kernels(info),
                "dfa_states": info["n_states"], "dfa_bytes_in_lds": info.get("v2_scan_bytes") if info.get("v2_tables") else info["dfa_bytes"],
                "decombined_fraction": round(hits_all / max(1, device.expected_read_count() * world if args.config != 4 else args.total_reads), 4),
                "parallelism": f"reads sharded x{world}, gather of DCR tuples to rank 0 ({'gloo' if dry else 'RCCL'})" if world > 1 else
                               ("single GPU" + (f", {overlapped} batches in flight (consecutive steps alternate between {overlapped} handles, each with its stream, workspace and record plane)" if overlapped >= 2 else "")
                                + (", a stream per chain" if overlapped and len(all_tables) > 1 else "")),
                "batches_in_flight": overlapped if overlapped >= 2 else 1,
                "world_size": comm.world if use_dist else 1, "devices": names,
                "collectives": ("gloo (dry run)" if dry else "RCCL through libdcrx (dcrx_comm_*: no torch in this process)") if use_dist else None,
            },
        }
        if dry:
            line["dry_run"] = True
            line["note"] = "CPU ranks over gloo with a stand-in device (tests/bench_dry.py): a test of the launch, sharding and gather protocol, not a measurement"
        else:
            step_avg_ms, kern_avg_ms = sum(step_ms) / len(step_ms), sum(kern_ms) / len(kern_ms)
            # (the timed steps that carry events: for config 4 the last, shorter step of a shard may be among them; the reads of
            # an average step are what the device time is set against)
            achieved = algo_bytes * (total_reads / world / args.steps) / (step_avg_ms * 1e-3) / 1e9
            if overlapped:      # the spans of a step's launches on their streams overlap with the next step's (the other chain's): the timed region's own clock
                achieved = algo_bytes * (total_reads / world) / elapsed / 1e9
            traffic, traffic_source = None, None
            handles = [tb for sl in getattr(device, "slots", [(all_tables,)]) for tb in sl[0]]      # (two batches in flight: both handles of a chain)
            forms = [tb.tune_state(n)["launch_form"] for tb in handles]
            # (a handle settles on one of two forms — list E a role of the finishing launch or inside the scan kernel —: each has its own profile)
            v2 = bool(info.get("v2_tables")) and not (args.cfg_flags & 64)
            roofline_lds = None
            if args.config == 2:
                # (constants from a profile, one file per launch form — a handle settles on list E as a role of the finishing launch or inside
                # the scan kernel —: quoted only while the sources they were measured on are the tree's — the digest of decombinator_amd/csrc
                # that tools/prof_summary_r05.py left in the file — else null, not a stale figure; two handles in flight: the mean of theirs)
                got = []
                for f in forms:
                    tpath = os.path.join(ROOT, "profiles", "traffic_list_e_inside_the_scan.json" if "list E inside" in f else "traffic.json")
                    try:
                        tj = json.load(open(tpath))
                        if (tj.get("reads_per_launch") == n and tj.get("read_len") == READ_LEN and tj.get("kernels") == device.kernels_tag(info)
                                and tj.get("csrc_sha16") == csrc_digest()):
                            got.append((os.path.basename(tpath), tj))
                    except Exception:
                        pass
                if forms and len(got) == len(forms):
                    traffic = int(sum(tj.get("hbm_bytes_per_step") or 0 for _, tj in got) / len(got))
                    traffic_source = ("profiles/" + " + ".join(sorted({nm for nm, _ in got})) + ": rocprofv3 --pmc passes of this command (one batch in flight) on an earlier run (" +
                                      got[0][1].get("profile", "?") + ") of the same kernel sources (csrc_sha16 " + got[0][1]["csrc_sha16"] + "), not measured by this run")
                    roofline_lds = got[0][1].get("roofline_lds")
                else:
                    traffic_source = "profiles/traffic*.json were measured on other kernel sources or another workload: not quoted"
            # what the timed steps ran on: the handle's own choice for its finishing launches (dcrx_tune_state: settled inside
            # the warm-up when that has five steps or more; 0 = not settled, the launches ran on 4096) and the scan blocks' choice
            # of tail waves (per block, from its region's share of tail reads in the launch before: no host-side state to report)
            line["tune"] = {
                "rescue_waves": [tb.tune_state(n)["rescue_waves"] for tb in handles],
                "samples_us": [{k: v for k, v in tb.tune_state(n).items() if k.startswith("us_")} for tb in handles],
                "launches_in_size_class": [tb.tune_state(n)["launches"] for tb in handles],
                "launch_form": forms,      # (what really ran: a tag set the v2 kernels do not serve shows here)
                "tail_waves": "per scan block: 2-6 by its region's share of tail reads in the previous launch (3 on a handle's first launch)",
                "first_launch_ms": None if first_launch_ms is None else round(first_launch_ms, 3),
            }
            line["roofline"] = {
                "bound": "hbm", "kernel": "all launches of a step (dcrx_decombine_device)" + (
                    "; launches of consecutive steps" + (" and of the two chains" if len(all_tables) > 1 else "") + " overlap on their streams: `achieved` is the algorithmic bytes of "
                    "the timed steps over the timed region's wall clock (barrier to barrier), step_device_ms_* the span of one step's launches on its own stream "
                    "(beside the neighbouring step's: longer than ms_per_step); --in-flight 1 times one step behind the other" if overlapped else ""),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_read": algo_bytes,
                "step_device_ms_avg": round(step_avg_ms, 5), "step_device_ms_min": round(min(step_ms), 5),
                "dominant_kernel": (("dcrx::scan2_kernel (scan + lean tail + list E's rescue fused: tail and event entries go through rings in LDS)" if "list E inside" in forms[-1]
                                     else "dcrx::scan2_kernel (scan + lean tail fused: the tail entries go through a ring in LDS)") if v2 else "dcrx::decombine_kernel"), "dominant_kernel_ms_avg": round(kern_avg_ms, 5),
                "dominant_kernel_ms_min": round(min(kern_ms), 5),
                "roofline_lds": roofline_lds,
                "events": f"HIP events on {len(timed)} of the {args.steps} timed steps (every {every}th, in turn the step's pair — on its first and last "
                          "dispatch — and the dominant kernel's pair: a step that carries them runs a few % longer, so the device-side averages can exceed ms_per_step)",
            }
            # the same algorithmic bytes over the host-side time of a step (launch gaps included)
            line["step_frac"] = round(algo_bytes * total_reads / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5)
        if gather is not None:
            hits_per_step = hits_all / (args.steps if args.config == 4 else 1)       # config 4: the counters are those of a whole pass
            tuple_mb = hits_per_step * gather.TUPLE_BYTES / 1e6
            line["gather"] = {
                "tuple_bytes": gather.TUPLE_BYTES,
                "mode": "sink (the decombine call leaves the message)" if gather.sink else "compaction of the records on a side stream",
                "mb_per_step_all_ranks": round(tuple_mb + world * ((n + 63) // 64) * 8 / 1e6, 3),
                "ms_per_step_without_gather": None if elapsed_nogather is None else round(elapsed_nogather / args.steps * 1e3, 4),
                "exposed_ms_per_step": None if elapsed_nogather is None else round((elapsed - elapsed_nogather) / args.steps * 1e3, 4),
            }
        if world == 1 and not args.no_cpu_baseline and not dry:
            line["cpu_baseline"] = cpu_baseline(nat, tables, ts, cfg_synth, min(args.cpu_sample, n))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if use_dist:
        comm.barrier(sptr)
        comm.close()


class HipDevice:
    """The rank's GPU: resident reads of every batch, the record planes and one dcrx_decombine_device per chain and step.
    Device memory, streams and events through the library's own entries (nat.DeviceBuffer / Stream / Event): no torch."""

    def __init__(self, nat, np, sptr, all_tables, cfg_synth, batches, n, stride, cfg_flags):
        self.nat, self.np, self.sptr = nat, np, sptr
        self.all_tables, self.batches, self.n, self.stride = all_tables, batches, n, stride
        self.cfg_flags = cfg_flags
        total = sum(b[1] for b in batches)
        # inputs resident in HBM before the timed region
        # (several chains: a batch is drawn in equal parts from each chain's germlines, part k from chain k)
        self.d_packed = nat.DeviceBuffer(max(1, total) * stride + 16)
        self.bc = []
        self._keep = []
        at = 0
        for first, cnt in batches:
            ers, eps, ecs = [], [], []
            for k, tb in enumerate(all_tables):
                lo, hi = cnt * k // len(all_tables), cnt * (k + 1) // len(all_tables)
                if hi > lo:
                    nat.check(nat.lib().dcrx_synth_reads_device(tb.handle, nat.C.byref(cfg_synth), first + lo, hi - lo, stride,
                                                                self.d_packed.ptr + (at + lo) * stride, sptr))
                e_r, e_p, e_c = nat.synth_exceptions_host(tb, cfg_synth, first + lo, hi - lo)
                ers.append(e_r.astype(np.int64) + lo); eps.append(e_p); ecs.append(e_c)
            er, ep, ec = np.concatenate(ers), np.concatenate(eps), np.concatenate(ecs)
            cluster = os.environ.get("DCRX_BENCH_N_CLUSTER")      # experiments only: "all:0.01" / "tail20:0.05" — that share of the reads all N / with a 20-nt N tail
            if cluster and cnt:
                kind, share = cluster.split(":")
                rs = np.random.default_rng(1234 + first)
                pick = np.nonzero(rs.random(cnt) < float(share))[0].astype(np.int64)
                span = np.arange(READ_LEN, dtype=np.int64) if kind == "all" else np.arange(READ_LEN - int(kind[4:]), READ_LEN, dtype=np.int64)
                cr = np.repeat(pick, len(span)); cp = np.tile(span, len(pick))
                key = np.concatenate([er.astype(np.int64) * 65536 + ep.astype(np.int64), cr * 65536 + cp])
                key, idx = np.unique(key, return_index=True)      # sorted by (read, position), one entry per place
                er, ep = key // 65536, key % 65536
                ec = np.concatenate([ec, np.full(len(cr), ord("N"), dtype=ec.dtype)])[idx]
                # ... packed as the packer packs such bytes: zero bits (dcrx_pack_reads; the device generator leaves the drawn base):
                # the batch's rows through the host, once, before anything is timed
                nat.synchronize()
                rows2d = np.zeros((cnt, stride), dtype=np.uint8)
                nat.check(nat.lib().dcrx_memcpy_d2h(rows2d.ctypes.data, self.d_packed.ptr + at * stride, rows2d.nbytes))
                lo_b, hi_b = int(span[0]) // 4, (READ_LEN + 3) // 4
                if int(span[0]) % 4:
                    rows2d[pick, lo_b] &= (1 << (2 * (int(span[0]) % 4))) - 1
                    lo_b += 1
                rows2d[pick, lo_b:hi_b] = 0
                nat.check(nat.lib().dcrx_memcpy_h2d(self.d_packed.ptr + at * stride, rows2d.ctypes.data, rows2d.nbytes))
            d_er = nat.DeviceBuffer.from_host(er.astype(np.uint32))
            d_ep = nat.DeviceBuffer.from_host(ep.astype(np.uint16))
            d_ec = nat.DeviceBuffer.from_host(ec.astype(np.uint8))
            self._keep.append((d_er, d_ep, d_ec))
            b = nat.BatchC()
            b.n_reads, b.packed, b.stride, b.read_len, b.lens = cnt, self.d_packed.ptr + at * stride, stride, READ_LEN, None
            b.n_exc = len(er)
            b.exc_read, b.exc_pos, b.exc_chr = (d_er.ptr, d_ep.ptr, d_ec.ptr) if len(er) else (None, None, None)
            self.bc.append(b)
            at += cnt
        self.d_recs = [nat.DeviceBuffer(n * 16) for _ in all_tables]      # one record plane per chain
        # a counter block per chain and batch of a pass (every call overwrites its own block): config 4's pass is summed on the host
        self.d_cnts = [[nat.DeviceBuffer.from_host(np.zeros(nat.N_COUNTERS, dtype=np.uint64)) for _ in self.bc] for _ in all_tables]
        self.cfg = nat.make_cfg("reverse", False, 130, cfg_flags)
        # a stream per chain (configs 3 and 5: two handles, two record planes): one chain's finishing launches run beside the other's scan
        # (round 6: 17.7 -> 19.5 and 16.3 -> 19.3 G reads/s, profiles/r06/a_stream_per_chain_ab.log; DCRX_BENCH_CHAIN_STREAMS=0: one stream, A/B)
        self.chain_streams = [nat.Stream() for _ in all_tables] if os.environ.get("DCRX_BENCH_CHAIN_STREAMS", "1") == "1" and len(all_tables) > 1 else None
        for tb in all_tables:
            nat.check(nat.lib().dcrx_reserve_device(tb.handle, n))
        self.slots = [(all_tables, self.d_recs, self.d_cnts, self.chain_streams)]      # (handles, record planes, counter blocks, streams) per batch in flight
        self.accumulate = len(batches) > 1
        self.last_batch = 0
        self.event_every = max(1, int(os.environ.get("DCRX_BENCH_EVENT_EVERY", "5")))
        self.compact = None            # TupleGather then compacts with dcrx_compact_hits_packed_device
        nat.synchronize()
        if os.environ.get("DCRX_BENCH_PRINT_PTRS") == "1":      # (experiments: where the buffers lie)
            print("PTRS packed %x records %x counters %x" % (self.d_packed.ptr, self.d_recs[-1].ptr, self.d_cnts[-1][0].ptr), file=sys.stderr)

    def second_slot(self, tables2):
        nat, np = self.nat, self.np
        for tb in tables2:
            nat.check(nat.lib().dcrx_reserve_device(tb.handle, self.n))
        recs = [nat.DeviceBuffer(self.n * 16) for _ in tables2]
        cnts = [[nat.DeviceBuffer.from_host(np.zeros(nat.N_COUNTERS, dtype=np.uint64)) for _ in self.bc] for _ in tables2]
        if self.chain_streams is None:
            self.slots[0] = (self.all_tables, self.d_recs, self.d_cnts, [nat.Stream() for _ in self.all_tables])
        self.slots.append((tables2, recs, cnts, [nat.Stream() for _ in tables2]))
        nat.synchronize()

    def name(self):
        return self.nat.device_name()

    def kernels_tag(self, info):
        return "v2" if bool(info.get("v2_tables")) and not (self.cfg_flags & 64) else "v1"

    def kernels(self, info):
        return ("v2 (scan2 with the lean tail inside — and list E's rescue where the handle's timing of both forms says so: tune.launch_form — / finish2: lean rescue + "
                "general form / list kernel)") if self.kernels_tag(info) == "v2" else "three-launch form"

    def make_events(self, steps, timed):
        nat = self.nat
        # (in turn the step's pair and the dominant kernel's pair; a run with one event-carrying step gets both on it)
        self.event_kind = {k: (i % 2 if len(timed) > 1 else 2) for i, k in enumerate(timed)}
        return {k: [tuple(nat.Event() for _ in range(4)) for _ in self.all_tables] for k in timed}

    def step(self, k, gather, ev):
        nat = self.nat
        kb = k % len(self.bc)
        b = self.bc[kb]
        self.last_batch = kb
        tables, d_recs, d_cnts, streams = self.slots[k % len(self.slots)]
        self.d_recs, self.d_cnts = d_recs, d_cnts      # (what the checks behind the timed region read: the last step's)
        rec = d_recs[-1]
        if gather is not None:           # alternating record buffers: the previous step's tuples are still being compacted
            gather.before_scan()
            rec = gather.records()
        for c, tb in enumerate(tables):
            last = c == len(tables) - 1
            if ev is not None:           # events: (step start, step stop, kernel start, kernel stop) per chain; the step's pair on
                kind = self.event_kind[k]             # every other event-carrying step, the kernel's pair on the ones between
                if kind in (0, 2):
                    nat.check(nat.lib().dcrx_set_step_events(tb.handle, ev[c][0].ptr, ev[c][1].ptr))
                if kind in (1, 2):
                    nat.check(nat.lib().dcrx_set_timing_events(tb.handle, ev[c][2].ptr, ev[c][3].ptr))
            nat.check(nat.lib().dcrx_decombine_device(tb.handle, nat.C.byref(self.cfg), nat.C.byref(b),
                                                      (rec if last else d_recs[c]).ptr, d_cnts[c][kb].ptr,
                                                      streams[c].ptr if streams else self.sptr))
            if ev is not None:
                nat.check(nat.lib().dcrx_set_step_events(tb.handle, None, None))
                nat.check(nat.lib().dcrx_set_timing_events(tb.handle, None, None))
        if gather is not None:
            gather.step(b.n_reads)

    def event_times(self, events, timed):
        st = [k for k in timed if self.event_kind[k] in (0, 2)]
        kt = [k for k in timed if self.event_kind[k] in (1, 2)]
        step_ms = [sum(e[0].elapsed_ms(e[1]) for e in events[k]) for k in st]         # all launches of a step, on the device
        kern_ms = [sum(e[2].elapsed_ms(e[3]) for e in events[k]) for k in kt]         # the dominant kernel(s) alone
        return step_ms, kern_ms

    def counters_host(self, chain):
        """The counters of `chain`: of the last step, or — a pass over several batches (config 4) — summed over the pass."""
        np, nat = self.np, self.nat
        nat.synchronize()
        blocks = self.d_cnts[chain] if self.accumulate else [self.d_cnts[chain][self.last_batch]]
        return sum(blk.to_host(np.uint64, nat.N_COUNTERS) for blk in blocks)

    def _counter(self, name):
        i = self.nat.COUNTER_NAMES.index(name)
        return [int(self.counters_host(c)[i]) for c in range(len(self.all_tables))]

    def totals(self):
        return sum(self._counter("vj_count")), min(self._counter("read_count"))

    def expected_read_count(self):
        return sum(b[1] for b in self.batches) if self.accumulate else self.n

    def device_errors(self):
        nat = self.nat
        return sum(int(self.counters_host(c)[nat.DEVICE_ERRORS]) for c in range(len(self.all_tables)))

    def last_step_hits(self):
        np, nat = self.np, self.nat
        nat.synchronize()
        return int(self.d_cnts[-1][self.last_batch].to_host(np.uint64, nat.N_COUNTERS)[nat.COUNTER_NAMES.index("vj_count")])


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.cfg_flags:
        os.environ.setdefault("DCRX_DEBUG_FLAGS", "1")      # profiling switches are refused by the library without this
    # HIP maps its streams onto 4 hardware queues by default; with the gather's streams (the side stream, RCCL's) beside
    # the three of a decombine call, two of those would share a queue and run their kernels one after the other.  Only then:
    # without the gather the default is right (the two-chain configs, two handles with three streams each, lose a third of
    # their rate on 8 queues: 0.81 -> 1.12 ms per step).
    gathering = args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("DCRX_BENCH_FORCE_GATHER") == "1"
    if gathering and args.config in (2, 4):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
