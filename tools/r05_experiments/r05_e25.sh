# the finishing launch in one round of blocks, list C's jobs first -> profiles/r05/finish_one_round_c_first_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
(cd $R && DCRX_DEBUG_ONE_ROUND=1 DCRX_LIB_PATH=$R/tools/variants/libdcrx_cfirst.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
for rep in 1 2; do
run "two rounds (C jobs first in order)" cfirst
DCRX_DEBUG_ONE_ROUND=1 run "one round, C then E per wave" cfirst
DCRX_DEBUG_ONE_ROUND=1 DCRX_DEBUG_RESCUE_WAVES=3072 run "one round E3072" cfirst
DCRX_DEBUG_ONE_ROUND=1 DCRX_DEBUG_RESCUE_WAVES=4096 run "one round E4096" cfirst
run "cfg5 two rounds" cfirst --config 5
DCRX_DEBUG_ONE_ROUND=1 run "cfg5 one round" cfirst --config 5
done
