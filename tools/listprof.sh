#!/bin/bash
# usage: tools/listprof.sh [flags...]  (on the GPU box): kernel times per profiling flag set
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in ${@:-0 8}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lst$f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cfg-flags $f > $R/gpurun_out/lst$f.log 2>&1
  python3 - $R/gpurun_out/lst$f $f <<'PY'
import csv,glob,os,sys
ps=sorted(glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(ps[-1])):
    if "dcrx" in r["Name"] and "synth" not in r["Name"]: print("flags="+sys.argv[2], r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
