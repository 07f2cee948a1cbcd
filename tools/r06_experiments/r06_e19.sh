# The handle's own choice of where list E is finished (from the first launch's list-E share): config 2 (10 %: inside the scan), config 2 at
# 2 % substitutions and config 5 (a role); parity through the change of form; the line says what ran
export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_TUNE=1
R=$GRAFT_REPO_ROOT; cd $R
unset DCRX_LIB_PATH
for c in 2 5; do timeout 600 python3 tests/forced_shape_worker.py $c 2097152 24 2>&1 | grep -E "dcrx tune|SHAPE_OK|rror" | cut -c1-220; done
cd /tmp
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4), d['tune']['launch_form'])"
  grep "dcrx tune: list E\|dcrx tune: scan +" /tmp/err.log | head -3
}
for rep in 1 2 3; do
run "config 2, own choice"
DCRX_DEBUG_FUSE_E=0 run "config 2, role forced"
done
DCRX_BENCH_SUB_RATE=0.02 run "config 2 at 2 % substitutions"
run "config 5" --config 5
run "config 3" --config 3
run "config 2, 100 M per launch" --reads 100000000 --steps 5 --warmup 3
