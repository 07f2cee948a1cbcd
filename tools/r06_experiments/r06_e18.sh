# role / fused, six interleaved pairs on one box: is the fused form immune to whatever makes the role form's scan read 0.217 or 0.25 ms?
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
export DCRX_LIB_PATH=$R/tools/variants/libdcrx_fuse_e.so
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2 3 4 5 6; do
DCRX_DEBUG_FUSE_E=0 run "role "
DCRX_DEBUG_X_BSPLIT=2 DCRX_DEBUG_TAIL_WAVES=2 run "fused"
done
DCRX_DEBUG_FUSE_E=0 DCRX_DEBUG_TAIL_WAVES=3 run "role, 3 tail waves forced"
DCRX_DEBUG_FUSE_E=0 DCRX_DEBUG_TAIL_WAVES=4 run "role, 4 tail waves forced"
