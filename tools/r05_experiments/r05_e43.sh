# Would list X's blocks out of the way let sixteen waves per region on list E run in one round?  A build that leaves list X alone
# (its 256 blocks leave at once: the records are NOT results) by waves on lists E and C -> profiles/r05/finish_without_list_x_by_waves.log
export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
R=$GRAFT_REPO_ROOT; cd /tmp
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
for e in 12 14 16; do for c in 4 8; do
  DCRX_DEBUG_RESCUE_WAVES=$((e*256)) DCRX_DEBUG_RESCUE_WAVES_C=$((c*256)) run "as shipped, E $e C $c waves per region" cur
  DCRX_DEBUG_RESCUE_WAVES=$((e*256)) DCRX_DEBUG_RESCUE_WAVES_C=$((c*256)) run "no list X,  E $e C $c waves per region" nox
done; done
done
