# tail waves that do nothing, by their number -> profiles/r05/scan_with_noop_tail_waves.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; fl=$3; shift 3
  for kv in "$@"; do export "$kv"; done
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 --cfg-flags $fl 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
  for kv in "$@"; do unset "${kv%%=*}"; done
}
for rep in 1 2; do
run "base2" base2 0
for tw in 2 3 4 5 6; do run "tailnoop tw$tw" tailnoop 0 DCRX_DEBUG_TAIL_WAVES=$tw; done
done
