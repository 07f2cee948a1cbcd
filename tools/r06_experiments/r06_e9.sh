# packed digest and one-pass exception slice against the commit before, six interleaved runs; then the GPU suite on HEAD, the
# forced gather on one rank by mode (ordering events without timestamps now)  -> profiles/r06/digest_and_exc_slice_ab.log, gather_one_rank.log
R=$GRAFT_REPO_ROOT; cd $R
bash tools/r06_experiments/r06_ab.sh r06_e9 6 base2 digest head2
O=$R/gpurun_out/r06_e9
(
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp
for rep in 1 2 3; do
python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['ms_per_step_steady'])"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DCRX_BENCH_FORCE_GATHER=1 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gather (sink)', d['ms_per_step'], d['ms_per_step_steady'], d['gather'])"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DCRX_BENCH_FORCE_GATHER=1 DCRX_BENCH_RESERVED_CUS=8 python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gather (sink), 8 CUs reserved', d['ms_per_step'], d['ms_per_step_steady'], d['gather'])"
done
) 2>&1 | tee $O/gather_one_rank.log
