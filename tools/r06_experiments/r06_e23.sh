# the long form by counters: which instantiation runs, its duration, vector / scalar / LDS instructions and waits (600 and 2000 nt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_e23; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in 600 2000; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$L -- python3 $R/tools/long_reads.py $L > $O/trace_$L.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_a_$L -- python3 $R/tools/long_reads.py $L > $O/pmc_a_$L.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_b_$L -- python3 $R/tools/long_reads.py $L > $O/pmc_b_$L.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_e23'
for L in (600,2000):
    for f in glob.glob(f'{O}/trace_{L}/**/*kernel_stats.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'long' in row['Name']: print('KSTAT', L, row['Name'][:90], row['Calls'], row['AverageNs'])
    for tag in ('a','b'):
        acc=collections.defaultdict(float); nd=collections.defaultdict(set)
        for f in glob.glob(f'{O}/pmc_{tag}_{L}/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                if 'long' not in row['Kernel_Name']: continue
                acc[row['Counter_Name']]+=float(row['Counter_Value']); nd[row['Counter_Name']].add(row['Dispatch_Id'])
        print('PMC', L, {k: round(v/max(1,len(nd[k]))/1e6,2) for k,v in acc.items()})
PY
rm -rf $O/trace_* $O/pmc_*
