# the finishing launch without list X / with the rescue a no-op / both -> profiles/r05/finish_ablations.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "cur" cur
run "nox" nox
run "rnoop" rnoop
run "rnoop+nox" rnoopnox
done
DCRX_LIB_PATH=$R/tools/variants/libdcrx_cur.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
python3 $R/tools/timeline.py /tmp/tr | tail -12
