#!/bin/bash
# usage (GPU box): tools/r04_finish_params3.sh — config 3 (extended sets: the tail a role of the finishing launch) by the waves that share a
# region's tail list (DCRX_DEBUG_TAIL_ROLE_WAVES) and its list E (DCRX_DEBUG_RESCUE_WAVES)
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT
cd /tmp
run() { local label=$1; shift; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"; }
for rep in 1 2; do
  for spec in ${SPECS:-8192:4096 4096:4096 6144:4096 12288:4096 16384:4096 8192:3072 8192:2048 6144:3072 4096:2048}; do
    t=${spec%%:*}; e=${spec##*:}
    DCRX_DEBUG_TAIL_ROLE_WAVES=$t DCRX_DEBUG_RESCUE_WAVES=$e run "config 3, tail-role waves $t, rescue waves $e, rep $rep" --config 3
  done
done
