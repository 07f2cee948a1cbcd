# config 3 with and without the loop priority / the lean digest, and its kernels' times
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
for l in c3cur c3noprio c3olddig c3both; do run "cfg3 $l" $l --config 3; done
done
DCRX_LIB_PATH=$R/tools/variants/libdcrx_c3cur.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/bench.py --no-cpu-baseline --steps 10 --config 3 > /dev/null 2>&1
python3 $R/tools/timeline.py /tmp/tr | tail -12
