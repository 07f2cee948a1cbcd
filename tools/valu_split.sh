#!/bin/bash
# usage: tools/valu_split.sh (GPU box): VALU / SALU / LDS instruction counts of the fast kernel, full and scan-only
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 2; do
  d=$R/gpurun_out/valu_split/f$f
  mkdir -p $d
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cfg-flags $f > $d.log 2>&1
  python3 - $d $f <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0]
        if "decombine_kernel" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print("VALUSPLIT flags="+sys.argv[2], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
