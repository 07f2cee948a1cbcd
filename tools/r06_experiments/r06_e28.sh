# a scan block of 768 threads (three waves per SIMD: a wave of the other handle's finishing launch fits beside it) with two batches in flight
R=$GRAFT_REPO_ROOT; cd /tmp
export DCRX_DEBUG_FLAGS=1 DCRX_LIB_PATH=$R/tools/variants/libdcrx_st.so
DCRX_DEBUG_SCAN_THREADS=768 timeout 600 python3 $R/tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -1 | cut -c1-160
for rep in 1 2 3; do for th in 1024 768 896; do for fe in 0 1; do for fl in 1 2; do
  DCRX_DEBUG_SCAN_THREADS=$th DCRX_DEBUG_FUSE_E=$fe python3 $R/bench.py --no-cpu-baseline --in-flight $fl 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN threads $th fuse_e $fe in_flight $fl', d['ms_per_step'], d['ms_per_step_steady'], d['value'], d['roofline']['dominant_kernel_ms_avg'])"
done; done; done; done
