#!/bin/bash
# usage: tools/ktrace_env.sh NAME=VALUE ... (GPU box): per-kernel durations of one bench run under the given environment
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/ktrace/env_$(echo "$@" | tr ' =' '__')
mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $d/log.txt 2>&1
python3 - $d "$@" <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'dcrx' in r['Name'] and 'synth' not in r['Name']: print("KSTAT", sys.argv[2:], r['Name'][11:40], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
