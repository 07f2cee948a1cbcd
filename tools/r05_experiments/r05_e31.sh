# The lean rescue's record stored where its fields are known (rescue2_fast_to) against the fields handed back through every early exit.
R=$GRAFT_REPO_ROOT; cd $R; export DCRX_DEBUG_FLAGS=1
DCRX_LIB_PATH=$R/tools/variants/libdcrx_${1:-onok}.so python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -3
cd /tmp
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2 3; do
run "base cfg2" ${2:-base}
run "${1:-onok} cfg2" ${1:-onok}
done
for rep in 1 2; do
DCRX_BENCH_SUB_RATE=0.02 run "base cfg2 sub 0.02" ${2:-base}
DCRX_BENCH_SUB_RATE=0.02 run "${1:-onok} cfg2 sub 0.02" ${1:-onok}
done
