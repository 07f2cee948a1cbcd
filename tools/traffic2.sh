#!/bin/bash
# usage: tools/traffic2.sh (GPU box): raw L2 memory-side request counters, normal run and scan-only run
# (the scan-only run moves a known byte count: 40 B read + 16 B written per read -> calibration)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in ${FLAGS:-0 2}; do
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$R/gpurun_out/traffic2/f$f/$(echo $pass | cut -d' ' -f1)
    mkdir -p $d
    timeout 400 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cfg-flags $f > $d.log 2>&1
  done
  python3 - $R/gpurun_out/traffic2/f$f $f <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0]
        if "dcrx::" in k and "synth" not in k and "prologue" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("TRAFFIC2 flags="+sys.argv[2], k[-40:], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
