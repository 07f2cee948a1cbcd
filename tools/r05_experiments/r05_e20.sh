# (a one-off check; nothing kept)
R=$GRAFT_REPO_ROOT; cd /tmp
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 DCRX_BENCH_FORCE_GATHER=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $R/bench.py --no-cpu-baseline --steps 20 --no-gather-ab > /dev/null 2>&1
python3 $R/tools/timeline.py /tmp/trg | tail -12
