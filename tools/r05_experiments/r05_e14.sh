# waves on list E x waves on list C -> profiles/r05/finish_waves_e_and_c_sweep.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "cur (own tune)" cur
for e in 2048 3072 4096; do for c in 256 512 1024 3072; do
DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_RESCUE_WAVES_C=$c run "E $e C $c" cur
done; done
done
