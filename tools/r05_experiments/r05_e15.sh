# list C's jobs folded behind list E's on the same waves -> profiles/r05/finish_list_c_folded_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
(cd $R && DCRX_LIB_PATH=$R/tools/variants/libdcrx_foldc.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
for rep in 1 2; do
DCRX_DEBUG_FOLD_C=0 run "separate C waves" foldc
run "C folded" foldc
DCRX_DEBUG_RESCUE_WAVES=4096 run "C folded E4096" foldc
DCRX_DEBUG_RESCUE_WAVES=3072 run "C folded E3072" foldc
DCRX_DEBUG_RESCUE_WAVES=2048 run "C folded E2048" foldc
DCRX_DEBUG_FOLD_C=0 run "cfg5 separate" foldc --config 5
run "cfg5 folded" foldc --config 5
DCRX_DEBUG_FOLD_C=0 run "cfg3 separate" foldc --config 3
run "cfg3 folded" foldc --config 3
done
