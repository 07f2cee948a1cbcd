#!/bin/bash
# usage (GPU box): tools/r04_sink_exp.sh TAG LIB ... — forced one-rank gather in sink mode through several builds of the library on ONE box
# (experiment builds: tools/variants/libdcrx_sink_NOSTORE.so leaves the lean roles' item stores out, ..._NOTUPLE.so their tuple arithmetic:
# their messages are not results, DCRX_BENCH_NO_GATHER_CHECK=1 lets the bench go on)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "$@"; do
    [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2963$rep DCRX_BENCH_FORCE_GATHER=1 DCRX_BENCH_NO_GATHER_CHECK=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 --no-gather-ab 2>$O/err.log | grep "^{" | tail -1 > $O/line.json
    python3 -c "import sys,json; d=json.loads(open('$O/line.json').read()); print('$lib rep $rep ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'step_dev', d['roofline']['step_device_ms_avg'])" || tail -5 $O/err.log
  done
done
