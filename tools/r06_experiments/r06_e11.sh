# The forced gather on one rank (RCCL through libdcrx), the previous step's transfers posted FIRST in a step: plain | sink |
# sink with 8 / 16 compute units reserved, three interleaved runs  -> profiles/r06/gather_one_rank.log
R=$GRAFT_REPO_ROOT; cd /tmp
G="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DCRX_BENCH_FORCE_GATHER=1"
show() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d.get('gather') or {}; print('$1', 'ms_per_step', d['ms_per_step'], 'steady', d['ms_per_step_steady'], 'device', d['roofline']['step_device_ms_avg'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'without gather', g.get('ms_per_step_without_gather'))"; }
for rep in 1 2 3; do
python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "plain               "
env $G python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "gather (sink)       "
env $G DCRX_BENCH_RESERVED_CUS=8 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "sink, 8 CUs reserved "
env $G DCRX_BENCH_RESERVED_CUS=16 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "sink, 16 CUs reserved"
done
env $G DCRX_BENCH_GATHER_MODE=narrow python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "narrow, compacted   "
