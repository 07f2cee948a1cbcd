#!/bin/bash
# usage (GPU box): tools/r04_finish_params.sh — the finishing launch by the number of waves that share the lean rescue's lists
# (DCRX_DEBUG_RESCUE_WAVES for list E, DCRX_DEBUG_RESCUE_WAVES_C for list C; / regions = waves per region), on configs 2 and 5
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT
cd /tmp
run() { local label=$1; shift; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"; }
for rep in 1 2; do
  for spec in 4096:4096 3072:3072 3072:4096 3072:2048 3072:6144 3072:1024 2816:4096; do
    e=${spec%%:*}; c=${spec##*:}
    DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_RESCUE_WAVES_C=$c run "config 2, rescue waves E $e C $c, rep $rep"
    DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_RESCUE_WAVES_C=$c run "config 5, rescue waves E $e C $c, rep $rep" --config 5
  done
done
