#!/bin/bash
# usage: tools/gatherprof.sh (GPU box): kernel times of the single-rank RCCL run (compaction + gather included)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DCRX_BENCH_FORCE_GATHER=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gather -- python3 $R/bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/gather.log 2>&1
tail -1 $R/gpurun_out/gather.log | cut -c1-200
python3 - $R/gpurun_out/gather <<'PY'
import csv,glob,os,sys
ps=sorted(glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(ps[-1])):
    if int(r["Calls"]) >= 10: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
