# Why does the same command read 0.31 or 0.35 ms on one box?  Ten runs in a row: the scan's time beside where the buffers lie.
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
for rep in 1 2 3 4 5 6 7 8 9 10; do
DCRX_BENCH_PRINT_PTRS=1 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4), d['tune']['rescue_waves'])"
grep PTRS /tmp/err.log | tail -1
done
