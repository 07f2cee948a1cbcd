#!/bin/bash
# usage: tools/prof_r06_counters.sh <tag> — only the kernel traces and counter passes of tools/prof_r06.sh (both forms of config 2), into
# gpurun_out/<tag>/summary (kernel_stats.csv, pmc_per_launch.json, traffic.json, timeline.txt) and .../summary/list_e_inside_the_scan
TAG=${1:-r06_counters}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O ${O}_fused
cd /tmp && export TMPDIR=/tmp
B="--steps 3 --warmup 1 --no-cpu-baseline --in-flight 1"
pmc_passes() {   # $1: directory, $2: DCRX_DEBUG_FUSE_E
  export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=$2
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $1/trace -- python3 $R/bench.py --no-cpu-baseline --in-flight 1 > $1/bench_under_kernel_trace.log 2>&1
  python3 $R/tools/timeline.py $1/trace > $1/timeline.txt 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $1/pmc_a -- python3 $R/bench.py $B > $1/pmc_a.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $1/pmc_b -- python3 $R/bench.py $B > $1/pmc_b.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $1/pmc_rd -- python3 $R/bench.py $B > $1/pmc_rd.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $1/pmc_wr -- python3 $R/bench.py $B > $1/pmc_wr.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $1/pmc_fetch -- python3 $R/bench.py $B > $1/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $1/pmc_write -- python3 $R/bench.py $B > $1/pmc_write.log 2>&1
  unset DCRX_DEBUG_FLAGS DCRX_DEBUG_FUSE_E
  python3 $R/tools/prof_summary_r05.py $1 > $1/summary.txt 2>&1
  cp $1/timeline.txt $1/summary.txt $1/summary/
  rm -rf $1/trace $1/pmc_*
}
pmc_passes $O 0
pmc_passes ${O}_fused 1
mkdir -p $O/summary/list_e_inside_the_scan; cp ${O}_fused/summary/* $O/summary/list_e_inside_the_scan/
tail -6 $O/summary.txt; tail -6 ${O}_fused/summary.txt
