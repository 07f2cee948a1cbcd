# List X's blocks at the end of the finishing launch's grid (DCRX_V2_X_LAST=1) against the front, by waves per region on list E
# -> profiles/r05/finish_list_x_blocks_last_ab.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
DCRX_LIB_PATH=$R/tools/variants/libdcrx_xl1.so python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -2
cd /tmp
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "X first, own choice" xl0
run "X last,  own choice" xl1
for e in 12 14 16; do
  DCRX_DEBUG_RESCUE_WAVES=$((e*256)) DCRX_DEBUG_RESCUE_WAVES_C=2048 run "X first, E $e C 8" xl0
  DCRX_DEBUG_RESCUE_WAVES=$((e*256)) DCRX_DEBUG_RESCUE_WAVES_C=2048 run "X last,  E $e C 8" xl1
done
done
DCRX_BENCH_SUB_RATE=0.02 run "X first, 2 % substitutions" xl0
DCRX_BENCH_SUB_RATE=0.02 run "X last,  2 % substitutions" xl1
DCRX_BENCH_N_RATE=0.01 run "X first, 1 % N" xl0
DCRX_BENCH_N_RATE=0.01 run "X last,  1 % N" xl1
