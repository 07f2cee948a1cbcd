# two-choice keyword tables (ph) against the bucket tables (prev), configs 2 and 5, fused and not -> profiles/r05/keyword_tables_two_choice_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
(cd $R && DCRX_LIB_PATH=$R/tools/variants/libdcrx_ph.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
for rep in 1 2 3; do
run "prev cfg2" prev
run "ph    cfg2" ph
done
for tw in 4 5; do DCRX_DEBUG_TAIL_WAVES=$tw run "ph cfg2 tw$tw" ph; DCRX_DEBUG_TAIL_WAVES=$tw run "prev cfg2 tw$tw" prev; done
run "prev cfg5" prev --config 5
run "ph    cfg5" ph --config 5
run "prev cfg2 nofuse" prev --cfg-flags 131072
run "ph    cfg2 nofuse" ph --cfg-flags 131072
