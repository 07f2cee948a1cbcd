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

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    hits = torch.from_numpy(rec[ok].view(np.uint8).reshape(-1, 16).copy())
    index = torch.from_numpy((ok + lo).astype(np.int64))
    gh, gi = sharded.gather_exact(hits, index, dst=0)
    total = sharded.reduce_counters(torch.from_numpy(cnt.astype(np.int64)))
    if rank == 0:
        whole = nat.synth_reads_host(t, nat.synth_cfg(seed=4), 0, N)
        wrec, wcnt = pu.oracle_records(ot, nat.unpack_reads(whole), "reverse", False, 130)
        wok = np.nonzero(wrec["status"] == 0)[0]
        assert gh.numpy().tobytes() == wrec[wok].tobytes(), "tuples differ from the unsharded run"
        assert (gi.numpy() == wok).all(), "indices differ"
        assert (total.numpy().astype(np.uint64) == wcnt).all(), "counters differ"
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


def test_shard_range_covers_everything():
    from decombinator_amd import sharded
    for n in (0, 1, 7, 1000, 10**9 + 7):
        for w in (1, 2, 3, 8):
            r = [sharded.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
