#!/bin/bash
# usage (GPU box): tools/r04_tail_waves.sh TAG — the fused scan with 3 / 4 / 5 tail waves forced (DCRX_DEBUG_TAIL_WAVES) and with the
# blocks' own choice (unset), on configs 2 and 5 at 10 M reads per step and on config 2 with 15 % and 70 % rearranged reads
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
TAG=${1:-r04_tw}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { # label, env assignment, bench args...
  local label=$1; shift
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>$O/err.log | tail -1 > $O/line.json
  python3 -c "import sys,json; d=json.loads(open('$O/line.json').read()); print('$label ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])" || tail -3 $O/err.log
}
for rep in 1 2; do
  for tw in 3 4 5 own; do
    if [ $tw = own ]; then unset DCRX_DEBUG_TAIL_WAVES; else export DCRX_DEBUG_TAIL_WAVES=$tw; fi
    one "config 2, tail waves $tw, rep $rep"
    one "config 5, tail waves $tw, rep $rep" --config 5
    DCRX_BENCH_P_REARRANGED=0.15 one "config 2 with 15 % rearranged, tail waves $tw, rep $rep"
    DCRX_BENCH_P_REARRANGED=0.70 one "config 2 with 70 % rearranged, tail waves $tw, rep $rep"
  done
done
