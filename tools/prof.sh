#!/bin/bash
# usage: tools/prof.sh <tag>   (runs on the GPU box through gpurun)
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$TAG/bench.log 2>&1
timeout 300 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cfg-flags 2 > $R/gpurun_out/$TAG/bench_scanonly.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG/trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/$TAG/pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/$TAG/pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc4.log 2>&1
find $R/gpurun_out/$TAG -name "*.csv" | head -30
python3 $R/tools/prof_summary.py $R/gpurun_out/$TAG
