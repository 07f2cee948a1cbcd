# config 3: rescue waves x tail-role waves -> profiles/r05/config3_finish_waves_sweep.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['value'])"
}
run "cfg3 default" fl --config 3
for e in 2048 3072 4096; do for t in 3072 4096 6144 8192; do
DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_TAIL_ROLE_WAVES=$t run "cfg3 E $e T $t" fl --config 3
done; done
run "cfg3 default" fl --config 3
