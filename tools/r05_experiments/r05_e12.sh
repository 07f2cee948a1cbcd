# the finishing launch with the rescue a no-op / every sweep stopped after one pair, configs 2 and 5 (the sweeps' bound on this round's kernels)
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "cur   cfg2" cur
run "rnoop cfg2" rnoop
run "sweep1 cfg2" sweep1
run "cur   cfg2 no-events(1024)" cur --cfg-flags 1024
run "cur   cfg5" cur --config 5
run "rnoop cfg5" rnoop --config 5
run "sweep1 cfg5" sweep1 --config 5
done
