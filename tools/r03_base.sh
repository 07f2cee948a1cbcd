#!/bin/bash
# usage (GPU box): tools/r03_base.sh TAG — micro-benchmark, bench line, list populations and kernel trace of the current build
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
TAG=${1:-r03_base}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/micro/valu_issue > $O/valu_issue.log 2>&1
timeout 300 python3 $R/bench.py --no-cpu-baseline > $O/bench_default.log 2>&1
DCRX_DEBUG_V2_COUNTS=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_counts.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 > $O/bench_under_kernel_trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
tail -3 $O/bench_default.log
