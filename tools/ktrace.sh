#!/bin/bash
# usage: tools/ktrace.sh <flags> (GPU box): per-kernel durations of one bench run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
f=${1:-0}
d=$R/gpurun_out/ktrace/f$f
mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cfg-flags $f > $d/log.txt 2>&1
python3 - $d <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'dcrx' in r['Name']: print("KSTAT", r['Name'][:70], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
