# tail waves at priority 0-3 on three and four of them -> profiles/r05/tail_wave_priority_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for rep in 1 2; do
for l in tp0 tp1 tp2 tp3; do
  for tw in 3 4; do DCRX_DEBUG_TAIL_WAVES=$tw run "tail prio $l tw$tw" $l; done
done
done
