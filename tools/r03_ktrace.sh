#!/bin/bash
# usage (GPU box): tools/r03_ktrace.sh TAG [bench args] — bench line + per-kernel durations of the current build
TAG=${1:-kt}
shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 30 "$@" > $O/bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 "$@" > $O/bench_under_kernel_trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
python3 - $O <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[1]+"/bench.log").read().strip().splitlines()[-1])
print("bench: ms_per_step", d["ms_per_step"], "step_dev", d["roofline"]["step_device_ms_avg"], "scan", d["roofline"]["dominant_kernel_ms_avg"], "value", d["value"])
for r in csv.DictReader(open(sys.argv[1]+"/kernel_stats.csv")):
    if 'dcrx' in r['Name'] and 'synth' not in r['Name']: print("  %-44s calls %3s avg_us %7.1f min %7.1f max %7.1f" % (r['Name'][5:49], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
