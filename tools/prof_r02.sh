#!/bin/bash
# usage: tools/prof_r02.sh <tag> [bench args]   (runs on the GPU box through gpurun)
# Collects what profiles/<tag>/ keeps: the default bench line, rocprofv3 --kernel-trace --stats of the same
# command, SQ counters and the L2's memory-side request counters (one --pmc pass per group), per kernel and step.
TAG=${1:-r02}
shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py "$@" > $O/bench_default.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/bench_under_kernel_trace.log 2>&1
B="--steps 3 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_a -- python3 $R/bench.py $B "$@" > $O/pmc_a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_b -- python3 $R/bench.py $B "$@" > $O/pmc_b.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_rd -- python3 $R/bench.py $B "$@" > $O/pmc_rd.log 2>&1
timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_wr -- python3 $R/bench.py $B "$@" > $O/pmc_wr.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $B "$@" > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $B "$@" > $O/pmc_write.log 2>&1
python3 $R/tools/prof_summary_r02.py $O
