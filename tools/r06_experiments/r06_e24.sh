# two-chain configs: a stream per chain (one chain's finishing launches beside the other's scan) against one stream
R=$GRAFT_REPO_ROOT; cd /tmp
for rep in 1 2 3; do for cfg in 3 5; do for cs in 0 1; do
  DCRX_BENCH_CHAIN_STREAMS=$cs python3 $R/bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN config $cfg chain_streams $cs', d['ms_per_step'], d['ms_per_step_steady'], d['value'])"
done; done; done
