"""world_size-2 gloo run (CPU) of the multi-GPU exchange: contiguous sharding,
gather of DCR tuples in rank order, counter all-reduce.  The tuples come from the
oracle here (no GPU in this test); on the GPU box the same functions carry the
HIP path's records over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["DCRX_ROOT"])
    from decombinator_amd import sharded, synth, _native as nat
    from oracle import oracle as orc
    from tests import parity_util as pu
    from tests.gloo_backend import GlooComm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = GlooComm()
    N = 6001
    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)
    lo, hi = sharded.shard_range(N, world, rank)
    hb = nat.synth_reads_host(t, nat.synth_cfg(seed=4), lo, hi - lo)   # any shard from (seed, index)
    rec, cnt = pu.oracle_records(ot, nat.unpack_reads(hb), "reverse", False, 130)
    ok = np.nonzero(rec["status"] == 0)[0]
    hits = rec[ok].view(np.uint8).reshape(-1, 16).copy()
    index = (ok + lo).astype(np.int64)
    gh, gi = sharded.gather_exact(comm, hits, index, dst=0)
    total = sharded.reduce_counters(comm, cnt)
    if rank == 0:
        whole = nat.synth_reads_host(t, nat.synth_cfg(seed=4), 0, N)
        wrec, wcnt = pu.oracle_records(ot, nat.unpack_reads(whole), "reverse", False, 130)
        wok = np.nonzero(wrec["status"] == 0)[0]
        assert gh.tobytes() == wrec[wok].tobytes(), "tuples differ from the unsharded run"
        assert (gi == wok).all(), "indices differ"
        assert (total == wcnt).all(), "counters differ"
        print("SHARDED_OK", len(wok))
    else:
        assert gh is None and gi is None
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_gloo_gather_matches_unsharded(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCRX_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARDED_OK" in outs[0]


GATHER_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["DCRX_ROOT"])
    from decombinator_amd import sharded, synth, _native as nat
    from oracle import oracle as orc
    from tests import parity_util as pu
    from tests.gloo_backend import GlooComm, GlooBackend

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, STEPS = 5000, 5
    ts = synth.config_tagset(2)
    vs, js = ts.half_splits
    t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, vs, js)
    ot = orc.OracleTables(ts.v_tags, ts.v_jumps, [r.upper() for r in ts.v_regions], ts.j_tags, ts.j_jumps,
                          [r.upper() for r in ts.j_regions], vs, js)

    def step_reads(step, r):
        """Reads of rank r's batch of `step`: full batches but for the last step of a run of more than two ranks — a shard's
        short last batch on the odd ranks, and no reads at all on rank 3 (its shard ended a step earlier: an empty step)."""
        if world > 2 and step == STEPS - 1:
            return 0 if r == 3 else (N - 1234 if r % 2 else N)
        return N

    def shard_records(step, r):
        """What rank r's device would hold after the scan of `step`: here from the oracle (p_rearranged differs per step
        and rank, so that the counts do: 0 % .. 95 % decombined, above any fixed fraction; rank 5 of a larger run never
        decombines a read: a shard without hits)."""
        if r == 0:
            p = [0.05, 0.95, 0.5, 0.0, 0.7][step]
        elif r == 1:
            p = [0.9, 0.1, 0.45, 0.6, 0.0][step]
        else:
            p = 0.0 if r == 5 else ((0.17 * r + 0.23 * step) % 1.0)
        n = step_reads(step, r)
        if n == 0:
            return np.zeros(0, dtype=nat.RECORD_DTYPE)
        hb = nat.synth_reads_host(t, nat.synth_cfg(seed=100 + step, p_rearranged=p), r * N, n)
        rec, _ = pu.oracle_records(ot, nat.unpack_reads(hb), "reverse", False, 130)
        return rec

    def compact(slot, n_reads):       # stands in for dcrx_compact_hits_packed_device: same layout, made on the host
        rec = np.frombuffer(slot["rec"].np.tobytes(), dtype=nat.RECORD_DTYPE)[:n_reads]
        if os.environ.get("DCRX_TUPLE8") == "narrow":      # dcrx_compact_hits_narrow_device: one message, bitmap | low words | high bytes
            m = g.codec.pack(rec, n_slots=N)
            slot["msg"].np[:len(m)] = m
            slot["n"].np.view(np.int64)[0] = int((rec["status"] == 0).sum())
            return
        w, bm = (sharded.pack_tuples8 if os.environ.get("DCRX_TUPLE8") == "1" else sharded.pack_tuples12)(rec)
        slot["hits"][:w.size * 4] = w.reshape(-1).view(np.uint8)
        slot["bitmap"][:] = 0
        slot["bitmap"][:len(bm)] = bm.view(np.int64)
        slot["n"].np.view(np.int64)[0] = len(w)

    g = sharded.TupleGather(N, GlooBackend(GlooComm()), depth=2, compact=compact, v_jumps=ts.v_jumps if os.environ.get("DCRX_TUPLE8") == "1" else None,
                            tables=t if os.environ.get("DCRX_TUPLE8") == "narrow" else None, max_read_len=150)
    assert g.TUPLE_BYTES == {"0": 12, "1": 8, "narrow": 5}[os.environ.get("DCRX_TUPLE8")]
    checked = 0
    for step in range(STEPS):
        g.before_scan()
        rec = shard_records(step, rank)
        raw = rec.view(np.uint8).reshape(-1)
        g.records().np[:raw.size] = raw                                              # "the scan wrote the records"
        g.step(len(rec))
        if step >= 1 and rank == 0:
            # the previous step is complete on rank 0 once its transfers are waited for: compare it in full
            g.finish()
            for r, (grec, gidx, _) in enumerate(g.gathered(step - 1)):
                want = shard_records(step - 1, r)
                ok = np.nonzero(want["status"] == 0)[0]
                assert (gidx == ok).all(), (step, r, "bitmap")
                w = want[ok].copy()
                assert grec.tobytes() == w.tobytes(), (step, r, "tuples")
                checked += len(ok)
        elif step >= 1:
            g.finish()
    g.finish()
    if rank == 0:
        for r, (grec, gidx, _) in enumerate(g.gathered(STEPS - 1)):
            want = shard_records(STEPS - 1, r)
            ok = np.nonzero(want["status"] == 0)[0]
            assert (gidx == ok).all() and grec.tobytes() == want[ok].tobytes()
        print("GATHER_OK", checked)
    dist.barrier()
    dist.destroy_process_group()
''')


import pytest


@pytest.mark.parametrize("tuple8,world", [("0", 2), ("1", 2), ("narrow", 2), ("narrow", 8)],
                         ids=["12-byte-tuples", "8-byte-tuples", "narrow-tuples", "narrow-tuples-8-ranks"])
def test_two_rank_tuple_gather_protocol_exact_sizes(tmp_path, tuple8, world):
    """The gather bench.py runs between ranks (count exchange, exact-size transfers of 12- or 8-byte tuples + bitmap,
    alternating slots), over gloo with two ranks — and with eight: the world size of BASELINE config 4 — and five steps
    whose decombined fractions range from 0 to 95 %: rank 0 re-expands every rank's tuples, and they are that rank's
    decombined records in read order — concatenated in rank order, the order of the reference's outdata.append
    (decombine.py:1039) over contiguous shards.  The eight-rank run ends as a sharded job does: short last batches, a rank
    with no reads left in the last step (an empty trailing step), and a rank whose shard never decombines a read."""
    script = tmp_path / "gworker.py"
    script.write_text(GATHER_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCRX_ROOT=ROOT, OMP_NUM_THREADS="1", DCRX_TUPLE8=tuple8)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_shard_range_covers_everything():
    from decombinator_amd import sharded
    for n in (0, 1, 7, 1000, 10**9 + 7):
        for w in (1, 2, 3, 8):
            r = [sharded.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


STAGE_WORKER = textwrap.dedent('''
    import json, os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["DCRX_ROOT"])
    from decombinator_amd import sharded, decombine as dec, io as dio, _native as nat
    from tests import golden_util as gu, parity_util as pu
    from tests.gloo_backend import GlooComm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = GlooComm()
    work = os.environ["DCRX_WORK"]
    stage = json.load(open(os.path.join(os.environ["DCRX_ROOT"], "tests", "golden", "stage_human_extended_b.json")))
    ot = gu.oracle_tables(stage["tagset"])

    def oracle_device(tables, batch, orientation="reverse", allow_ns=False, lenthreshold=130, flags=0):
        return pu.oracle_records(ot, nat.unpack_reads(batch), orientation, allow_ns, lenthreshold)   # stands in for the GPU (CPU test)
    nat.decombine = oracle_device
    dec.BATCH_READS = 7                       # many small batches: every rank gets several, the last one is short
    args = json.load(open(os.path.join(work, "args.json")))
    rows = sharded.decombinator_sharded(args, comm)
    info = comm.allgather_object(dict(dec.stage_info))
    if rank == 0:
        json.dump({"rows": [list(r) for r in rows], "counts": {k: int(v) for k, v in dec.counts.items() if k not in ("start_time", "end_time")},
                   "info": info}, open(os.path.join(work, "sharded.json"), "w"))
    else:
        assert rows is None
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("world,run_index,cut", [(2, 0, "shards"), (3, 0, "shards"), (2, 1, "shards"), (3, 2, "shards"), (2, 0, "crlf"),
                                                 (2, 0, "wrapped"), (3, 1, "wrapped")])
def test_sharded_stage_equals_single_process(tmp_path, monkeypatch, world, run_index, cut):
    """decombinator_sharded() on two and three gloo ranks (the oracle standing in for the GPUs) against decombinator() in this
    process on the same files: the same rows in the same order, the same counters, one summary log — with the input read in
    shards (every rank its own byte ranges of the FASTQ files: disjoint, in rank order, covering the files, the R1 / R2 files
    of a pair cut at the same record; bc_read R1 keeps its record pairs together), and, for a file that cannot be cut (CRLF
    line ends; records over more than four lines whose line count is a multiple of four all the same), with every rank reading
    the whole file as before."""
    import json
    from decombinator_amd import decombine as dec, io as dio, _native as nat
    from tests import golden_util as gu, parity_util as pu
    from decombinator_amd import synth
    from tests import test_host_stage as ths
    stage = json.load(open(os.path.join(ROOT, "tests", "golden", "stage_human_extended_b.json")))
    run = stage["runs"][run_index % len(stage["runs"])]
    work = tmp_path / "w"
    work.mkdir()
    ts = stage["tagset"]
    synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"], v_jumps=ts["v_jumps"],
                 v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"], j_jumps=ts["j_jumps"],
                 j_names=ts["j_names"], j_regions=ts["j_regions"]).write(str(work / "tags"))
    eol = "\r\n" if cut == "crlf" else "\n"

    def wrapped(text):
        # two early records with their sequence and quality over two lines each: four lines more, so that the file still
        # has a multiple of four lines and every later record still starts on a line 4 k — what the newline counts cannot see
        lines = text.split("\n")
        out = []
        for k in range(0, len(lines) - 1, 4):
            h, sq, plus, ql = lines[k:k + 4]
            if k in (8, 12):      # (not the first record: fastq_check, as the reference's :126-179, reads the file's first four lines as one)
                m = len(sq) // 2
                out += [h, sq[:m], sq[m:], plus, ql[:m], ql[m:]]
            else:
                out += [h, sq, plus, ql]
        return "\n".join(out) + "\n"
    for name, key in (("SYNTH_1.fq", "fastq_r1"), ("SYNTH_2.fq", "fastq_r2")):
        text = stage[key]
        (work / name).write_bytes((wrapped(text) if cut == "wrapped" else text.replace("\n", eol)).encode())
    (work / "single").mkdir()
    (work / "sharded").mkdir()

    def make_args(out):
        return dio.create_args_dict(infile=str(work / "SYNTH_1.fq"), chain="b", bc_read=run["bc_read"], dontgzip=True, dontcount=True,
                                    orientation=run["orientation"], allowNs=run["allowNs"], tagfastadir=str(work / "tags"),
                                    outpath=str(work / out) + os.sep, command="decombine")
    json.dump(make_args("sharded"), open(work / "args.json", "w"))
    monkeypatch.setattr(nat, "decombine", ths._oracle_device(stage))
    monkeypatch.setattr(dec, "BATCH_READS", 7)
    dec.counts.clear()
    want_rows = [list(r) for r in dec.decombinator(make_args("single"))]
    assert want_rows == run["rows"]
    want_counts = {k: int(v) for k, v in dec.counts.items() if k not in ("start_time", "end_time")}
    script = tmp_path / "sworker.py"
    script.write_text(STAGE_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCRX_ROOT=ROOT, DCRX_WORK=str(work), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = json.load(open(work / "sharded.json"))
    assert len(want_rows) > (20 if run["bc_read"] == "R2" else 10)
    assert got["rows"] == want_rows
    assert got["counts"] == want_counts
    # how the ranks read: their own byte ranges only — or, for the file that cannot be cut, the whole file each
    info = got["info"]
    assert [i["rank"] for i in info] == list(range(world))
    if cut in ("crlf", "wrapped"):
        assert not any(i["sharded_input"] for i in info)
    else:
        assert all(i["sharded_input"] for i in info)
        files = [work / "SYNTH_1.fq"] + ([work / "SYNTH_2.fq"] if run["bc_read"] == "R2" else [])
        for f, path in enumerate(files):
            data = path.read_bytes()
            at, n_rec = 0, []
            for i in info:
                b, e = i["byte_ranges"][f]
                assert b == at and e >= b              # disjoint, in rank order, no gap
                at = e
                piece = data[b:e]
                assert piece == b"" or piece.startswith(b"@")
                assert piece.count(b"\n") % (4 if run["bc_read"] == "R2" else 8) == 0 or i["rank"] == world - 1
                n_rec.append(piece.count(b"\n") // 4)
            assert at == len(data)
            if f == 0:
                first_counts = n_rec
                assert sum(1 for k in n_rec if k) >= 2          # (more than one rank really read something)
            else:
                assert n_rec == first_counts                   # the files of a pair are cut at the same records
    logs = list((work / "sharded" / "Logs").glob("*.csv"))          # written once, by rank 0, from the summed counters
    assert len(logs) == 1
    keep = [ln for ln in logs[0].read_text().split("\n") if not ln.startswith(("Directory,", "DateFinished,", "TimeFinished,", "TimeTaken"))]
    assert keep == run["summary_lines"]


FAIL_WORKER = textwrap.dedent('''
    import json, os, sys
    import torch.distributed as dist
    sys.path.insert(0, os.environ["DCRX_ROOT"])
    from decombinator_amd import sharded, decombine as dec, _native as nat
    from tests import golden_util as gu, parity_util as pu
    from tests.gloo_backend import GlooComm

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = GlooComm()
    work = os.environ["DCRX_WORK"]
    stage = json.load(open(os.path.join(os.environ["DCRX_ROOT"], "tests", "golden", "stage_human_extended_b.json")))
    ot = gu.oracle_tables(stage["tagset"])
    calls = [0]

    def oracle_device(tables, batch, orientation="reverse", allow_ns=False, lenthreshold=130, flags=0):
        calls[0] += 1
        if rank == 1 and calls[0] == 2:
            raise ValueError("rank 1 stops at its second batch")       # one rank alone fails in the middle of its loop
        return pu.oracle_records(ot, nat.unpack_reads(batch), orientation, allow_ns, lenthreshold)
    nat.decombine = oracle_device
    dec.BATCH_READS = 7
    args = json.load(open(os.path.join(work, "args.json")))
    try:
        sharded.decombinator_sharded(args, comm)
    except ValueError as e:
        assert rank == 1, e
        print("RAISED_OWN", e)
    except RuntimeError as e:
        assert rank == 0 and "rank 1" in str(e), e
        print("RAISED_PEER", e)
    else:
        raise SystemExit("no error surfaced on rank %d" % rank)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_a_rank_that_fails_alone_fails_every_rank_without_a_hang(tmp_path):
    """an exception on one rank (here in the middle of its read loop) used to leave the other ranks blocked in
    the first collective.  Every rank now exchanges an error flag first; the failing rank re-raises its own exception and
    the others raise a RuntimeError naming it.  Both ranks start on an output directory that does not exist yet (the
    makedirs race of the same finding: only rank 0 creates it, with exist_ok)."""
    import json
    from decombinator_amd import io as dio, synth
    stage = json.load(open(os.path.join(ROOT, "tests", "golden", "stage_human_extended_b.json")))
    run = stage["runs"][0]
    work = tmp_path / "w"
    work.mkdir()
    ts = stage["tagset"]
    synth.TagSet(species=ts["species"], tags=ts["tags"], chain=ts["chain"], v_tags=ts["v_tags"], v_jumps=ts["v_jumps"],
                 v_names=ts["v_names"], v_regions=ts["v_regions"], j_tags=ts["j_tags"], j_jumps=ts["j_jumps"],
                 j_names=ts["j_names"], j_regions=ts["j_regions"]).write(str(work / "tags"))
    (work / "SYNTH_1.fq").write_text(stage["fastq_r1"])
    (work / "SYNTH_2.fq").write_text(stage["fastq_r2"])
    args = dio.create_args_dict(infile=str(work / "SYNTH_1.fq"), chain="b", bc_read=run["bc_read"], dontgzip=True, dontcount=True,
                                orientation=run["orientation"], allowNs=run["allowNs"], tagfastadir=str(work / "tags"),
                                outpath=str(work / "fresh") + os.sep, command="decombine")
    json.dump(args, open(work / "args.json", "w"))
    script = tmp_path / "fworker.py"
    script.write_text(FAIL_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DCRX_ROOT=ROOT, DCRX_WORK=str(work), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RAISED_PEER" in outs[0] and "RAISED_OWN" in outs[1], outs


def test_tuple_gather_refuses_tables_the_8_byte_tuple_cannot_hold():
    """the 8-byte tuple masks v to 11 and j to 9 bits — a larger tag set travels as 12-byte tuples, and a jump table
    that does not match the V tags is an error."""
    from decombinator_amd import sharded
    from tests.gloo_backend import GlooBackend
    g = sharded.TupleGather(64, GlooBackend(), v_jumps=[40] * 60, n_v=60, n_j=13)
    assert g.TUPLE_BYTES == 8
    assert sharded.TupleGather(64, GlooBackend(), v_jumps=[40] * 2048).TUPLE_BYTES == 12
    assert sharded.TupleGather(64, GlooBackend(), v_jumps=[40] * 60, n_v=60, n_j=512).TUPLE_BYTES == 12
    with pytest.raises(ValueError):
        sharded.TupleGather(64, GlooBackend(), v_jumps=[40] * 59, n_v=60, n_j=13)
