#!/bin/bash
# usage: tools/traffic.sh <tag>  (GPU box): HBM traffic counters of one bench run, separate passes
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc4.log 2>&1
python3 $R/tools/prof_summary.py $R/gpurun_out/$TAG
