#!/bin/bash
# usage (GPU box): tools/r04_tune_ab.sh — the handle's own choice of the waves on list E (V2Tune) against the fixed 16 (DCRX_DEBUG_NO_TUNE=1), one box
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT
cd /tmp
run() { local label=$1; shift; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 12 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"; }
for rep in 1 2 3; do
  for c in 2 5; do
    DCRX_DEBUG_NO_TUNE=1 run "config $c, fixed 16 waves, rep $rep" --config $c
    run "config $c, own choice, rep $rep" --config $c
  done
  DCRX_DEBUG_NO_TUNE=1 DCRX_BENCH_P_REARRANGED=0.7 run "config 2 with 70 % rearranged, fixed 16 waves, rep $rep"
  DCRX_BENCH_P_REARRANGED=0.7 run "config 2 with 70 % rearranged, own choice, rep $rep"
done
