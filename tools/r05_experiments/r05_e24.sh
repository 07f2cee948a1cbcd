# the ring's atomics asked for an item early -> profiles/r05/atomics_asked_early_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
(cd $R && DCRX_LIB_PATH=$R/tools/variants/libdcrx_early.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
for rep in 1 2 3; do
run "cur" cur
run "atomics asked for early" early
DCRX_DEBUG_TAIL_WAVES=3 run "early, 3 tail waves" early
done
run "cur cfg5" cur --config 5
run "early cfg5" early --config 5
run "cur nofuse" cur --cfg-flags 131072
run "early nofuse" early --cfg-flags 131072
