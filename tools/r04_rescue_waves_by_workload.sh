export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT
cd /tmp
run() { local label=$1; shift; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"; }
for rep in 1 2; do
 for e in 4096 3072; do
  DCRX_DEBUG_RESCUE_WAVES=$e DCRX_BENCH_SUB_RATE=0.02 run "config 2 at 2 % substitutions, rescue waves $e, rep $rep"
  DCRX_DEBUG_RESCUE_WAVES=$e DCRX_BENCH_SUB_RATE=0.005 run "config 5 at 0.5 % substitutions, rescue waves $e, rep $rep" --config 5
  DCRX_DEBUG_RESCUE_WAVES=$e DCRX_BENCH_SUB_RATE=0.01 run "config 2 at 1 % substitutions, rescue waves $e, rep $rep"
  DCRX_DEBUG_RESCUE_WAVES=$e DCRX_BENCH_P_REARRANGED=0.7 run "config 2 with 70 % rearranged, rescue waves $e, rep $rep"
 done
done
DCRX_DEBUG_V2_COUNTS=1 DCRX_BENCH_SUB_RATE=0.02 timeout 200 python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "dcrx v2 lists" | head -1
DCRX_DEBUG_V2_COUNTS=1 DCRX_BENCH_SUB_RATE=0.005 timeout 200 python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --config 5 2>&1 | grep "dcrx v2 lists" | head -2
