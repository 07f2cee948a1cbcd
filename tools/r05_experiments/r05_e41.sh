# Reads per lane in the scan (chains in flight per wave): the shipped two with the next item prefetched (shape 2) against four without
# prefetch (shape 1), after tools/micro/lds_chains.hip said the LDS serves look-ups faster with more chains in flight
# (2.99 ns per wave-look-up at 2 x 16 chains, 2.44 at 4 x 16) -> profiles/r05/scan_reads_per_lane_ab.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], d['tune']['launch_form'])"
}
for rep in 1 2; do
run "shape 2 (2 reads per lane, prefetch)" --cfg-flags 512
run "shape 1 (4 reads per lane, no prefetch)" --cfg-flags 256
for tw in 2 3 4; do DCRX_DEBUG_TAIL_WAVES=$tw run "shape 1, $tw tail waves" --cfg-flags 256; done
run "shape 2 scan only" --cfg-flags 514
run "shape 1 scan only" --cfg-flags 258
done
