# scanning waves that also take tail batches when the ring runs full (flex builds) -> profiles/r05/flexible_waves_experiment.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for l in flex0 flex2; do (cd $R && DCRX_LIB_PATH=$R/tools/variants/libdcrx_$l.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1); done
for rep in 1 2; do
run "ph (own tw)" ph
for l in flex0 flex1 flex2 flex4; do
  for tw in 2 3 4; do DCRX_DEBUG_TAIL_WAVES=$tw run "$l tw$tw" $l; done
done
done
