#!/bin/bash
# usage (GPU box): tools/r03_kw.sh — the bench line at several (steps, warmup) pairs, interleaved
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for kw in "20 3" "20 5" "50 5" "20 20" "5 1"; do
  set -- $kw
  echo -n "steps $1 warmup $2: "; timeout 300 python3 $R/bench.py --no-cpu-baseline --gpus 1 --steps $1 --warmup $2 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'device', d['roofline'].get('step_device_ms_avg'))"
done; done
