#!/bin/bash
# usage: tools/prof_r06.sh <tag> [bench args]   (runs on the GPU box through gpurun)
# Collects what profiles/<tag>/ keeps: the default bench line (with the CPU baseline), rocprofv3 --kernel-trace --stats of the
# same command (default_two_batches_in_flight/) and of --in-flight 1 with each launch form forced, the average timeline of a step, SQ counters and the L2's memory-side request counters (one --pmc pass per
# group), per kernel and step; then the extra lines: the A/B forms on this box (tail as a role of the finishing launch instead
# of inside the scan; round 3's side streams; separate launches; the three-launch form), configs 3 and 5 at 10 M and 100 M reads
# per step, config 2 at 100 M, config 4 on one GPU, the forced gather, orientations, PCIe-inclusive rate, the stage, read
# lengths, clustered exception bytes, the LDS layout micro-benchmark, the read-fetch layouts by themselves.
TAG=${1:-r06}
shift
ARGS=("$@")      # further bench arguments (the counter passes below are a function: its own "$@" are not these)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py "$@" > $O/bench_default.log 2>&1
# the same command under the kernel trace (two batches in flight since round 6: the kernels of consecutive steps overlap)
mkdir -p ${O}_default
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_default/trace -- python3 $R/bench.py --no-cpu-baseline "$@" > ${O}_default/bench_under_kernel_trace.log 2>&1
python3 $R/tools/timeline.py ${O}_default/trace > ${O}_default/timeline.txt 2>&1
python3 $R/tools/prof_summary_r05.py ${O}_default > ${O}_default/summary.txt 2>&1
# one batch in flight (--in-flight 1: a step behind the other, the timeline of a step by itself), list E a role of the finishing launch
timeout 600 python3 $R/bench.py --no-cpu-baseline --in-flight 1 "$@" 2>/dev/null | tail -1 > $O/bench_one_batch_in_flight.log
DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --in-flight 1 "$@" > $O/bench_under_kernel_trace.log 2>&1
python3 $R/tools/timeline.py $O/trace > $O/timeline.txt 2>&1
B="--steps 3 --warmup 1 --no-cpu-baseline --in-flight 1"
# The counters of BOTH forms a config-2 handle may settle on (round 6: list E a role of the finishing launch / inside the scan kernel —
# the handle times both and keeps the faster, which differs from box to box): each form forced for its passes (DCRX_DEBUG_FUSE_E), the
# bench line quotes the traffic of the form it ran.
pmc_passes() {   # $1: directory suffix, $2: DCRX_DEBUG_FUSE_E
  export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=$2
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O$1/pmc_a -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_a.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O$1/pmc_b -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_b.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O$1/pmc_rd -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_rd.log 2>&1
  timeout 400 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O$1/pmc_wr -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_wr.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O$1/pmc_fetch -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O$1/pmc_write -- python3 $R/bench.py $B "${ARGS[@]}" > $O$1/pmc_write.log 2>&1
  unset DCRX_DEBUG_FLAGS DCRX_DEBUG_FUSE_E
}
pmc_passes "" 0
mkdir -p ${O}_fused; cp $O/bench_default.log ${O}_fused/ 2>/dev/null
DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_fused/trace -- python3 $R/bench.py --no-cpu-baseline --in-flight 1 "$@" > ${O}_fused/bench_under_kernel_trace.log 2>&1
python3 $R/tools/timeline.py ${O}_fused/trace > ${O}_fused/timeline.txt 2>&1
pmc_passes "_fused" 1
python3 $R/tools/prof_summary_r05.py ${O}_fused > ${O}_fused/summary.txt 2>&1
python3 $R/tools/prof_summary_r05.py $O > $O/summary.txt 2>&1
S=$O/summary
cp $O/timeline.txt $S/timeline.txt
mkdir -p $S/list_e_inside_the_scan; cp ${O}_fused/summary/* $S/list_e_inside_the_scan/ 2>/dev/null; cp ${O}_fused/timeline.txt ${O}_fused/summary.txt $S/list_e_inside_the_scan/ 2>/dev/null
mkdir -p $S/default_two_batches_in_flight; cp ${O}_default/summary/kernel_stats.csv ${O}_default/timeline.txt ${O}_default/bench_under_kernel_trace.log $S/default_two_batches_in_flight/ 2>/dev/null
cp $O/bench_one_batch_in_flight.log $S/ 2>/dev/null
export DCRX_DEBUG_FLAGS=1
for spec in tail_as_a_role:131072 side_streams:65536 separate_launches:32768 three_launch_form:64 scan_only:2; do
  name=${spec%%:*}; fl=${spec##*:}
  timeout 600 python3 $R/bench.py --no-cpu-baseline --cfg-flags $fl 2>/dev/null | tail -1 > $S/bench_config2_$name.log
done
timeout 600 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $S/bench_config2_again_same_box.log
timeout 600 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $S/bench_config2_steps20_warmup5.log
DCRX_BENCH_STEP_TRACE=1 timeout 600 python3 $R/bench.py --no-cpu-baseline --steps 200 --warmup 5 2>&1 | grep step_trace > $S/step_time_over_200_steps.log
for c in 3 5; do
  timeout 600 python3 $R/bench.py --no-cpu-baseline --config $c --steps 20 2>/dev/null | tail -1 > $S/bench_config${c}.log
  DCRX_BENCH_CHAIN_STREAMS=0 timeout 600 python3 $R/bench.py --no-cpu-baseline --config $c --steps 20 2>/dev/null | tail -1 > $S/bench_config${c}_both_chains_on_one_stream.log
  timeout 600 python3 $R/bench.py --no-cpu-baseline --config $c --reads 100000000 --steps 5 --warmup 1 2>/dev/null | tail -1 > $S/bench_config${c}_100M_reads.log
done
timeout 600 python3 $R/bench.py --no-cpu-baseline --reads 100000000 --steps 5 --warmup 1 2>/dev/null | tail -1 > $S/bench_config2_100M_reads_per_step.log
timeout 600 python3 $R/bench.py --no-cpu-baseline --reads 100000000 --steps 5 --warmup 1 --in-flight 1 2>/dev/null | tail -1 > $S/bench_config2_100M_reads_per_step_one_batch_in_flight.log
timeout 600 python3 $R/bench.py --no-cpu-baseline --config 4 --total-reads 1000000000 --warmup 1 2>/dev/null | tail -1 > $S/bench_config4_one_gpu.log
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29618 DCRX_BENCH_FORCE_GATHER=1 timeout 600 python3 $R/bench.py --no-cpu-baseline > $O/forced_gather.out 2> $O/forced_gather.err; grep "^{" $O/forced_gather.out | tail -1 > $S/bench_forced_gather_one_rank.log; tail -3 $O/forced_gather.err
bash $R/tools/r04_gather_ab.sh ${TAG}_gather > $S/gather_modes_ab.log 2>&1
timeout 300 $R/tools/micro/stream_layouts > $S/stream_layouts.log 2>&1
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29619 DCRX_BENCH_FORCE_GATHER=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_g -- python3 $R/bench.py --no-cpu-baseline --steps 20 --no-gather-ab > /dev/null 2>&1
python3 $R/tools/timeline.py $O/trace_g | tail -6 > $S/timeline_forced_gather_last_step.txt; rm -rf $O/trace_g
timeout 300 python3 $R/tools/both_rate.py 2>/dev/null | grep ORIENTATION > $S/orientations.log
timeout 300 python3 $R/tools/pcie.py 2>/dev/null | grep PCIE > $S/pcie.log
timeout 900 python3 $R/tools/stage.py --reads 4000000 --py-gzip 2>/dev/null | grep STAGE > $S/stage.log
timeout 600 python3 $R/tools/long_reads.py 2>/dev/null | tail -8 > $S/long_reads.log
bash $R/tools/r04_cliff.sh > $S/cliff_clustered_n.log 2>&1
timeout 300 $R/tools/micro/lds_gather > $S/lds_gather_layouts.log 2>&1
rm -rf $O/trace $O/pmc_* ${O}_fused/trace ${O}_fused/pmc_* ${O}_default/trace
cat $O/summary.txt | tail -12
