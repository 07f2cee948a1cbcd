# config 3's beta chain fused (DCRX_DEBUG_FUSE_LIMIT_KB=80) -> profiles/r05/config3_beta_fused_ab.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 8 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['value'], d['tune']['launch_form'], d['tune']['rescue_waves'])"
}
for rep in 1 2; do
run "cfg3 limit 64 KB (beta as role)" fl --config 3
DCRX_DEBUG_FUSE_LIMIT_KB=80 run "cfg3 beta fused" fl --config 3
DCRX_DEBUG_FUSE_LIMIT_KB=80 DCRX_DEBUG_TAIL_WAVES=3 run "cfg3 beta fused tw3" fl --config 3
DCRX_DEBUG_FUSE_LIMIT_KB=80 DCRX_DEBUG_TAIL_WAVES=5 run "cfg3 beta fused tw5" fl --config 3
done
