# a raised priority around the ring push (no gain; in atomics_asked_early_ab.log's neighbourhood)
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for rep in 1 2 3; do
run "cur" cur
run "push prio 3" pp3
run "push prio 2" pp2
done
