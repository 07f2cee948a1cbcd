#!/bin/bash
# usage: tools/pmc_icache.sh (GPU box): instruction-cache counters of the decombine kernels
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_icache
mkdir -p $d
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $d/a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $d/b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d/b.log 2>&1
python3 - $d <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0].split("<")[0]
        if any(t in k for t in ("decombine", "rescue2", "tail2", "events2", "scan2")): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print("PMC", k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
