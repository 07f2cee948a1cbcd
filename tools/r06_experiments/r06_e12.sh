# Where do the long form's 3.2 ms for 2 M reads of 600 nt go?  Instruction and wait counters of decombine_long_kernel.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_e12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/long_reads.py 500 600 2>/dev/null | grep LONG
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_a -- python3 $R/tools/long_reads.py 600 > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $O/pmc_b -- python3 $R/tools/long_reads.py 600 > $O/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("pmc_a","pmc_b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"][:60]; acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
        for k,v in acc.items():
            if "long" in k or "decombine" in k: print(d,k,{c: round(x/13,0) for c,x in v.items()})
PY
