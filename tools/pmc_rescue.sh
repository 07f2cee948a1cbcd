#!/bin/bash
# usage: tools/pmc_rescue.sh (GPU box): instruction / cycle counters of the kernels of one bench run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcr1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcr1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmcr2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcr2.log 2>&1
python3 - $R <<'PY'
import csv,glob,sys,collections
R=sys.argv[1]
for d in ("pmcr1","pmcr2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for p in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"][:40]
            if "dcrx" not in k or "synth" in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVES","SQ_ACTIVE_INST_VALU"): n[k]+=1
    for k,v in agg.items():
        print(d,k,"launches",n[k],{c:round(x/max(n[k],1)) for c,x in v.items()})
PY
