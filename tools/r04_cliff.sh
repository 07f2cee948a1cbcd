#!/bin/bash
# usage (GPU box): tools/r04_cliff.sh — step time on inputs with CLUSTERED exception bytes: 0.1 / 1 / 5 % of
# the reads all N (failed clusters) or with a 20-nt N tail, next to the uniform per-base rates of round 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
one() { timeout 600 python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'decombined', d['config']['decombined_fraction'])"; }
echo -n "baseline (0.05 % of the bases, uniform): "; one
for kind in all tail20; do
  for share in 0.001 0.01 0.05; do
    export DCRX_BENCH_N_CLUSTER=$kind:$share
    echo -n "clustered $kind $share: "; one
  done
done
unset DCRX_BENCH_N_CLUSTER
for nr in 0.01 0.1; do
  export DCRX_BENCH_N_RATE=$nr
  echo -n "uniform n_rate $nr: "; one
done
