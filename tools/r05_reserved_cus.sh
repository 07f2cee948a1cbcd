#!/bin/bash
# usage (GPU box): tools/r05_reserved_cus.sh — what compute units left to RCCL cost a one-rank step with the gather forced (no
# peer: nothing is sent; the scan's persistent blocks simply have fewer units): 0 / 2 / 4 / 8 / 16 reserved, twice.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_reserved; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
  for n in 0 2 4 8 16; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2963$rep DCRX_BENCH_FORCE_GATHER=1 DCRX_BENCH_RESERVED_CUS=$n timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 30 --no-gather-ab 2>$O/err.log | grep "^{" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced gather (sink), $n CUs reserved', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
  done
done
