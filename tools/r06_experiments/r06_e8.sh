# After the native RCCL binding: the GPU suite, the smoke check, the bench as the driver runs it (N = 1), the bench with the
# gather forced on one rank (RCCL through libdcrx, no torch in the process), the same under torch.distributed.run with one process
# -> profiles/r06/native_rccl_first_run.log
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e8; mkdir -p $O
(
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp
echo "== bench.py --steps 20 --warmup 5"
( time python3 $R/bench.py --steps 20 --warmup 5 ) 2>&1 | tail -6 | cut -c1-1500
echo "== forced gather, one rank (own spawn env)"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DCRX_BENCH_FORCE_GATHER=1 python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -3 | cut -c1-2500
echo "== under torch.distributed.run, one process, gather forced"
DCRX_BENCH_FORCE_GATHER=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 $R/bench.py --gpus 1 --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -3 | cut -c1-2500
) 2>&1 | tee $O/native_rccl_first_run.log
