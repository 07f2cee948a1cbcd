#!/bin/bash
# usage (GPU box): tools/r03_timeline.sh TAG [bench args] — kernel trace of the bench and the average timeline of a step
TAG=${1:-tl}
shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 "$@" > $O/bench_under_kernel_trace.log 2>&1
python3 $R/tools/timeline.py $O/trace | tee $O/timeline.txt
rm -rf $O/trace
