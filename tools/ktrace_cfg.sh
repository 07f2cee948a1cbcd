#!/bin/bash
# usage: tools/ktrace_cfg.sh <config> (GPU box): per-kernel durations of one bench run of BASELINE config 3 or 5
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
c=${1:-3}
d=$R/gpurun_out/ktrace/cfg$c
mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $c > $d/log.txt 2>&1
python3 - $d $c <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'dcrx' in r['Name'] and 'synth' not in r['Name']: print("KSTAT cfg"+sys.argv[2], r['Name'][11:60], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
