# config 2: consecutive steps on two streams (two handles, two record planes): a step's finishing launch beside the next step's scan
R=$GRAFT_REPO_ROOT; cd /tmp
export DCRX_DEBUG_FLAGS=1
for rep in 1 2 3; do for fl in 1 2; do for fe in -1 0 1; do
  DCRX_DEBUG_FUSE_E=$fe DCRX_BENCH_BATCHES_IN_FLIGHT=$fl python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN in_flight $fl fuse_e $fe', d['ms_per_step'], d['ms_per_step_steady'], d['value'], d['tune']['launch_form'], d['roofline']['dominant_kernel_ms_avg'])"
done; done; done
for cfg in 3 5; do DCRX_BENCH_CHAIN_STREAMS=1 DCRX_BENCH_BATCHES_IN_FLIGHT=2 python3 $R/bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN config $cfg chain streams + 2 in flight', d['ms_per_step'], d['ms_per_step_steady'], d['value'])"; done
