# Same box: round 5's kernels | + the scan's first item landed before its loop | + the lean roles' batch in hand landed before
# the next is asked for -> profiles/r06/prefetch_really_in_flight_ab.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e4; mkdir -p $O
(
for v in scanfix landed; do
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$v.so timeout 600 python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -1 | cut -c1-200
done
DCRX_LIB_PATH=$R/tools/variants/libdcrx_landed.so timeout 600 python3 tests/forced_shape_worker.py 5 2097152 3 2>&1 | tail -1 | cut -c1-200
cd /tmp
run() { n=$1; lib=$2; shift 2
  export DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so
  python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'steady', d.get('ms_per_step_steady'), 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4), 'frac', d['roofline']['frac'])" || tail -3 $O/err.log
}
for rep in 1 2 3; do
run "round 5 kernels      " r05
run "scan landed          " scanfix
run "scan + roles landed  " landed
done
DCRX_BENCH_SUB_RATE=0.02 run "round 5, 2 % substitutions     " r05
DCRX_BENCH_SUB_RATE=0.02 run "scan landed, 2 % substitutions " scanfix
DCRX_BENCH_SUB_RATE=0.02 run "all landed, 2 % substitutions  " landed
) 2>&1 | tee $O/prefetch_really_in_flight_ab.log
