# Round 6, first call: the shipped step on this box; when the scan kernel's waves start / leave their loops / end (stamps build);
# the read-fetch layouts by themselves (tools/micro/stream_layouts.hip)  -> profiles/r06/scan_wave_stamps.log, stream_layouts.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e1; mkdir -p $O
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
  grep "scan stamps" $O/err.log
}
(
run "shipped 50/30" --steps 50 --warmup 30
run "shipped 20/5" --steps 20 --warmup 5
DCRX_LIB_PATH=$R/tools/variants/libdcrx_stamps.so DCRX_STAMPS_LAUNCH=40 run "stamps build" --steps 40 --warmup 30
DCRX_LIB_PATH=$R/tools/variants/libdcrx_stamps.so DCRX_STAMPS_LAUNCH=41 run "stamps build again" --steps 40 --warmup 30
) 2>&1 | tee $O/scan_wave_stamps.log
./tools/micro/stream_layouts 2>&1 | tee $O/stream_layouts.log
