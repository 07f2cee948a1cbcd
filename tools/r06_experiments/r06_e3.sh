# The scan kernel's first item waited for (and the compiler told so) before the loop: no `s_waitcnt vmcnt(0)` in front of every
# item's first look-up any more (it sat behind the next item's loads: a memory round trip per item and wave with nothing in flight)
# -> profiles/r06/scan_prefetch_really_in_flight.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e3; mkdir -p $O
(
timeout 600 python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -2 | cut -c1-300
cd /tmp
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'steady', d.get('ms_per_step_steady'), 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4), 'frac', d['roofline']['frac'])" || tail -3 $O/err.log
}
for rep in 1 2 3; do
run "fixed, 20/5 " --steps 20 --warmup 5
run "fixed, 50/30" --steps 50 --warmup 30
done
DCRX_BENCH_PREROLL_STEPS=0 run "fixed, 20/5 no pre-roll" --steps 20 --warmup 5
run "config 3" --config 3 --steps 20 --warmup 5
run "config 5" --config 5 --steps 20 --warmup 5
) 2>&1 | tee $O/scan_prefetch_really_in_flight.log
