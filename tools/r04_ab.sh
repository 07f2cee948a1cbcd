#!/bin/bash
# usage (GPU box): tools/r04_ab.sh TAG LIB[:FLAGS] ... — the same bench through several builds of the library (and cfg flags) on
# ONE box (boxes of the pool differ by up to 30 %): bench lines twice round-robin, then the step's timeline from a kernel trace.
# LIB = `default` or a path under the repo (tools/variants/*.so).  TRACE=0 skips the traces.
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DCRX_DEBUG_FLAGS=1
for rep in 1 2 3; do
  for spec in "$@"; do
    lib=${spec%%:*}; fl=0; [[ "$spec" == *:* ]] && fl=${spec##*:}
    [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
    timeout 300 python3 $R/bench.py --no-cpu-baseline --steps ${STEPS:-50} --warmup ${WARMUP:-10} --cfg-flags $fl ${BENCH_ARGS} 2>$O/err.log | tail -1 > $O/line.json
    python3 -c "import sys,json; d=json.loads(open('$O/line.json').read()); print('$spec rep $rep ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'step_dev', d['roofline']['step_device_ms_avg'])" || tail -5 $O/err.log
  done
done
if [ "${TRACE:-1}" = "1" ]; then
  for spec in "$@"; do
    lib=${spec%%:*}; fl=0; [[ "$spec" == *:* ]] && fl=${spec##*:}
    [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
    name=$(basename $lib .so)_$fl
    timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu-baseline --steps 20 --cfg-flags $fl ${BENCH_ARGS} > /dev/null 2>&1
    echo "=== $spec"; python3 $R/tools/timeline.py $O/trace_$name | tee $O/timeline_$name.txt
    rm -rf $O/trace_$name
  done
fi
