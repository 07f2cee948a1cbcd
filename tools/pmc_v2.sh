#!/bin/bash
# usage: tools/pmc_v2.sh <flags...> (GPU box): instruction / cycle counters of the decombine kernels for bench.py --cfg-flags F
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in "$@"; do
  d=$R/gpurun_out/pmc_v2/f$f
  mkdir -p $d
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --cfg-flags $f > $d/a.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $d/b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --cfg-flags $f > $d/b.log 2>&1
  python3 - $d $f <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0].split("<")[0]
        if any(t in k for t in ("decombine", "rescue2", "tail2", "events2", "scan2")): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print("PMC flags="+sys.argv[2], k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
done
