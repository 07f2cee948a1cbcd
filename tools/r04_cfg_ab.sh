#!/bin/bash
# usage (GPU box): tools/r04_cfg_ab.sh — configs 3 and 5 (two chains per step) in the shipped form, with the tail as a role of the
# finishing launch (131072) and in round 3's shape (65536), at 10 M and 100 M reads per step, on one box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DCRX_DEBUG_FLAGS=1
for c in 3 5; do
  for fl in 0 131072 65536; do
    for reads in 10000000 100000000; do
      st=20; [ $reads = 100000000 ] && st=5
      echo -n "config $c flags $fl reads $reads: "
      timeout 600 python3 $R/bench.py --no-cpu-baseline --config $c --cfg-flags $fl --reads $reads --steps $st --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
    done
  done
done
