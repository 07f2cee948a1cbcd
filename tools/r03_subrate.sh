#!/bin/bash
# usage (GPU box): tools/r03_subrate.sh LIB... — step time by substitution rate of the synthetic reads (more half-tag rescues, more leftovers of the lean forms)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sr in 0.005 0.02 0.05 0.10; do
for lib in "$@"; do
  [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
  export DCRX_BENCH_SUB_RATE=$sr
  echo -n "sub_rate $sr $(basename $lib .so): "
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 20 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done; done
