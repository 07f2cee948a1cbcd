#!/bin/bash
# usage (GPU box): tools/r03_ab.sh TAG LIB_A LIB_B ... — the same bench through several builds of the library on ONE box
# (boxes of the pool differ by up to 10 %): bench line, step timeline and VALU / SALU instruction counts per kernel
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
  echo "=== $name"
  for rep in 1 2; do
    timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
  done
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
  python3 $R/tools/timeline.py $O/trace_$name | sed 's/^/  /'
  rm -rf $O/trace_$name
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python3 - $O/pmc_$name <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"]
        if "dcrx::" not in k or "synth" in k: continue
        k=k.split("dcrx::")[1].split("<")[0].split("(")[0]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print("  pmc %-24s"%k, {c: round(x/ max(1,n[(k,c)])/1e6,2) for c,x in v.items()}, "M per launch")
PY
  rm -rf $O/pmc_$name
done
