# The finishing launch compiled for three waves per SIMD (144 registers, nothing in scratch memory) against four (128, a handful of
# spills inside the rescue's loop: 3.5 scratch loads and 1.9 stores per batch, counted) — per number of rescue waves.
R=$GRAFT_REPO_ROOT; cd $R; export DCRX_DEBUG_FLAGS=1
DCRX_LIB_PATH=$R/tools/variants/libdcrx_rw3.so python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -2
cd /tmp
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "4 waves/SIMD own tune" cur2
run "3 waves/SIMD own tune" rw3
for e in 1536 2048 3072; do for c in 768 1536; do
DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_RESCUE_WAVES_C=$c run "3 waves/SIMD E $e C $c" rw3
done; done
done
DCRX_BENCH_SUB_RATE=0.02 run "4 waves/SIMD sub 0.02" cur2
DCRX_BENCH_SUB_RATE=0.02 run "3 waves/SIMD sub 0.02" rw3
