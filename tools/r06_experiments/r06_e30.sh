# two batches in flight, list E inside the scan (forced): rescue waves per scan block 2 / 3 / 4 / 5, and the tail waves' count
R=$GRAFT_REPO_ROOT; cd /tmp
export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=1
for rep in 1 2 3; do for rw in 3 2 4 5; do
  DCRX_DEBUG_FUSE_E_WAVES=$rw python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN rescue_waves $rw', d['ms_per_step'], d['ms_per_step_steady'], d['ms_per_step_one_batch_in_flight'], d['value'])"
done; done
for tw in 1 2 3; do
  DCRX_DEBUG_TAIL_WAVES=$tw python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN tail_waves $tw (rescue 3)', d['ms_per_step'], d['ms_per_step_steady'], d['ms_per_step_one_batch_in_flight'], d['value'])"
done
