#!/bin/bash
# usage (GPU box): tools/r03_cliff.sh — step time on hostile inputs: many reads with exception bytes, heavy substitution
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for nr in 0.0005 0.01 0.1 0.5; do
  export DCRX_BENCH_N_RATE=$nr
  echo -n "n_rate $nr: "
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
unset DCRX_BENCH_N_RATE
for sr in 0.2 0.3; do
  export DCRX_BENCH_SUB_RATE=$sr
  echo -n "sub_rate $sr: "
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
