# `--steps 20 --warmup 5` reads 5 % above `--steps 50 --warmup 10`: where are the 0.4 ms?  The step time over the timed region
# (events on every step), and the same 20 steps behind longer warm-ups.
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $*"; DCRX_BENCH_STEP_TRACE=1 python3 bench.py --no-cpu-baseline "$@" 2>&1 | grep -E "step_trace|ms_per_step" | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('  ms_per_step', d['ms_per_step'], 'device step avg', d.get('device_ms_per_step'), 'scan', d['roofline']['dominant_kernel_ms_avg'])
    else: print('  '+l)"; }
run --steps 20 --warmup 5
DCRX_BENCH_EVENT_EVERY=1 run --steps 20 --warmup 5
run --steps 20 --warmup 30
run --steps 20 --warmup 100
run --steps 100 --warmup 5
run --steps 200 --warmup 5
run --steps 20 --warmup 5
