#!/bin/bash
# usage (GPU box): tools/r03_gather_ab.sh — what the multi-GPU gather costs on one GPU (one rank, no peer: compaction to 8-byte
# tuples + count exchange beside the next step), with 4 and 8 hardware queues, and the kernels of one such step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29618 DCRX_BENCH_FORCE_GATHER=1
for rep in 1 2; do
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  echo -n "hardware queues $q: "
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['gather'])"
done; done
unset GPU_MAX_HW_QUEUES
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gtrace/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 --no-gather-ab > /dev/null 2>&1
python3 $R/tools/trace_step.py $R/gpurun_out/gtrace/trace | head -16
rm -rf $R/gpurun_out/gtrace/trace
