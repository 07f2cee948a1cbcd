cd /tmp
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DCRX_BENCH_FORCE_GATHER=1 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -12 | cut -c1-600
