# finishing blocks of 128 / 256 / 512 threads -> profiles/r05/finish_block_size_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for l in fb512 fb128; do (cd $R && DCRX_LIB_PATH=$R/tools/variants/libdcrx_$l.so timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1); done
for rep in 1 2; do
run "256-thread finishing blocks" foldc
run "512" fb512
DCRX_DEBUG_RESCUE_WAVES=4096 run "512 E4096" fb512
run "128" fb128
done
