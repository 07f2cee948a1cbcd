#!/bin/bash
# usage (GPU box): tools/r05_ab.sh TAG LIB[:FLAGS] ... — per build of the library (tools/variants/*.so, or `default`): the smoke
# check against the oracle (20 000 reads, every field and counter), then the bench lines round-robin on ONE box.
# CHECK=0 skips the smoke checks.  BENCH_ARGS: further bench.py arguments.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DCRX_DEBUG_FLAGS=1
if [ "${CHECK:-1}" = "1" ]; then
  for spec in "$@"; do
    lib=${spec%%:*}
    [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
    (cd $R && timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | sed "s|^|$lib: |")
  done
fi
for rep in 1 2 3; do
  for spec in "$@"; do
    lib=${spec%%:*}; fl=0; [[ "$spec" == *:* ]] && fl=${spec##*:}
    [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
    timeout 300 python3 $R/bench.py --no-cpu-baseline --steps ${STEPS:-40} --warmup ${WARMUP:-10} --cfg-flags $fl ${BENCH_ARGS} 2>$O/err.log | tail -1 > $O/line.json
    python3 -c "import sys,json; d=json.loads(open('$O/line.json').read()); print('$spec rep $rep ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'step_dev', d['roofline']['step_device_ms_avg'])" || tail -5 $O/err.log
  done
done
