#!/bin/bash
# usage: tools/ktrace_lib.sh LIB FLAGS (GPU box): per-kernel durations with another build of the library
export DCRX_LIB_PATH=$1
f=${2:-0}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/ktrace/lib_$(basename $1)_$f
mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cfg-flags $f > $d/log.txt 2>&1
python3 - $d $(basename $1) $f <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'tail2' in r['Name']: print("KSTAT", sys.argv[2], sys.argv[3], r['Name'][11:40], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
