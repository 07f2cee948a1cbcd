# Waves per region on list E in steps of one (10-16), list C's on the same count or on four: the finishing launch's time by how the
# region's 61 batches divide over its waves -> profiles/r05/finish_waves_per_region_fine_sweep.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
run "own choice"
for w in 10 11 12 13 14 15 16; do
  DCRX_DEBUG_RESCUE_WAVES=$((w*256)) run "E = C = $w waves per region"
  DCRX_DEBUG_RESCUE_WAVES=$((w*256)) DCRX_DEBUG_RESCUE_WAVES_C=1024 run "E $w, C 4 waves per region"
done
done
