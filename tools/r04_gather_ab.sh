#!/bin/bash
# usage (GPU box): tools/r04_gather_ab.sh TAG — what the gather of a sharded run costs on one GPU, on ONE box: the plain step,
# then DCRX_BENCH_FORCE_GATHER=1 with one rank (no peer: the message is made, the count exchanged, nothing sent) in the three
# modes — sink (the decombine call leaves the message), narrow (the same 5-byte tuples compacted from the records on a side
# stream) and tuple8 (round 3: 8-byte tuples, compacted) — and the sink mode with 16 compute units reserved as N > 1 runs do.
TAG=${1:-r04_gather}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "import sys,json; d=json.loads(open('$O/line.json').read()); g=d.get('gather') or {}; print('$1 ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'tuple_bytes', g.get('tuple_bytes'), 'mb_per_step', g.get('mb_per_step_all_ranks'))" || tail -5 $O/err.log; }
for rep in 1 2 3; do
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 2>$O/err.log | tail -1 > $O/line.json; line "plain rep $rep"
  for mode in sink narrow tuple8; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$rep DCRX_BENCH_FORCE_GATHER=1 DCRX_BENCH_GATHER_MODE=$mode timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 --no-gather-ab 2>$O/err.log | grep "^{" | tail -1 > $O/line.json; line "forced gather $mode rep $rep"
  done
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2962$rep DCRX_BENCH_FORCE_GATHER=1 DCRX_BENCH_RESERVED_CUS=16 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 --no-gather-ab 2>$O/err.log | grep "^{" | tail -1 > $O/line.json; line "forced gather sink, 16 CUs reserved rep $rep"
done
