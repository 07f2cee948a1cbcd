"""`python bench.py --gpus N` as typed: the parent starts its own ranks (no launcher).  On the CPU the same spawner and
rank program run through tests/bench_dry.py (gloo, tests/dry_device.py in place of the GPU); on a GPU box `--gpus 1` goes
through the real path.  bench.py itself has no dry switch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRY = os.path.join("tests", "bench_dry.py")


def _run(args, timeout=900, script="bench.py"):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p


def _line(p):
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return json.loads(lines[0])


def test_two_ranks_started_by_the_bench_itself_dry():
    d = _line(_run(["--gpus", "2", "--reads", "3000", "--steps", "3", "--warmup", "1"], script=DRY))
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and len(d["config"]["devices"]) == 2
    assert d["dry_run"] is True and d["value"] is None            # a dry run never carries a rate
    assert d["scaling"] == "weak" and d["gather"]["tuple_bytes"] == 5
    assert d["gather"]["ms_per_step_without_gather"] is not None


def test_config4_fixed_total_sharded_dry():
    # 10 001 reads over 2 ranks in steps of 2 000: shards of 5 000 / 5 001 reads -> 3 steps each, the last ones short
    d = _line(_run(["--gpus", "2", "--config", "4", "--total-reads", "10001", "--reads", "2000", "--warmup", "1"], script=DRY))
    assert d["scaling"] == "strong" and d["steps"] == 3 and d["n_gpus"] == 2
    assert "10001" in d["config"]["workload"]
    assert 0.3 < d["config"]["decombined_fraction"] < 0.5


def test_eight_ranks_config4_uneven_total_dry():
    # BASELINE config 4's world size: 16 001 reads over 8 ranks in steps of 2 000 -> shards of 2 000 and (one) 2 001 reads:
    # one rank needs a second step for its last read, the other seven run an empty trailing step; every rank's message is
    # checked against its count on rank 0 (TupleGather.check), the line names the world and every rank's own step time
    d = _line(_run(["--gpus", "8", "--config", "4", "--total-reads", "16001", "--reads", "2000", "--warmup", "1"], script=DRY))
    assert d["scaling"] == "strong" and d["steps"] == 2 and d["n_gpus"] == 8 and d["world_size"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and len(d["config"]["devices"]) == 8
    assert d["gather"]["tuple_bytes"] == 5 and d["gather"]["exposed_ms_per_step"] is not None
    assert 0.3 < d["config"]["decombined_fraction"] < 0.5


def test_a_failing_rank_fails_the_bench():
    p = _run(["--gpus", "2", "--reads", "-5", "--steps", "1", "--warmup", "0"], timeout=300, script=DRY)
    assert p.returncode != 0


@pytest.mark.gpu
def test_one_gpu_through_the_plain_command():
    d = _line(_run(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]))
    assert d["n_gpus"] == 1 and d["value"] > 1000 and d["roofline"]["frac"] > 0.05
    assert "dry_run" not in d


@pytest.mark.gpu
def test_config4_on_one_gpu_small_total():
    d = _line(_run(["--gpus", "1", "--config", "4", "--total-reads", "25000000", "--warmup", "1", "--no-cpu-baseline"]))
    assert d["scaling"] == "strong" and d["steps"] == 3 and d["value"] > 1000


def test_bench_itself_has_no_dry_switch_and_no_checker_import():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--dry" not in src and "from tests" not in src and "import tests" not in src
    p = _run(["--gpus", "1", "--dry-gloo"], timeout=120)
    assert p.returncode != 0
