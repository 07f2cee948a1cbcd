# the fused scan by parts: tail waves a no-op, no ring push, by tail waves -> profiles/r05/ring_push_parts.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for rep in 1 2; do
run "cur" cur
run "tail no-op" tailnoop
run "no ring push" noring
DCRX_DEBUG_TAIL_WAVES=2 run "no ring push, 2 tail waves" noring
DCRX_DEBUG_TAIL_WAVES=2 run "tail no-op, 2 tail waves" tailnoop
done
