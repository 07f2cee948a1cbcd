# Three reads per lane, no item in registers ahead (RPL = 3: 48 look-up chains per CU instead of 32), with and without
# cache-warming loads of the next item, against the shipped two reads per lane + one item ahead
# -> profiles/r06/scan_three_reads_per_lane_ab.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e2; mkdir -p $O
(
for v in rpl3 rpl3nw; do
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$v.so timeout 600 python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -2 | cut -c1-300
done
cd /tmp
run() { n=$1; lib=$2; shift 2
  [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so
  python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
}
for rep in 1 2 3; do
run "shipped (2 reads per lane, one item ahead)" default
run "3 reads per lane, warm loads             " rpl3
run "3 reads per lane, no warm loads          " rpl3nw
done
for tw in 3 4 5; do
DCRX_DEBUG_TAIL_WAVES=$tw run "3 reads per lane, warm, $tw tail waves" rpl3
done
) 2>&1 | tee $O/scan_three_reads_per_lane_ab.log
