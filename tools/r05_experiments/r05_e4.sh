# the ring's size (16 / 8 / 4 batches), a scan that skips every third look-up (skip3: three bases per look-up's ceiling), the tail outside the scan -> profiles/r05/ring_batches_and_skip3_lookups.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { # name lib flags envs...
  n=$1; lib=$2; fl=$3; shift 3
  for kv in "$@"; do export "$kv"; done
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 --cfg-flags $fl 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
  for kv in "$@"; do unset "${kv%%=*}"; done
}
for rep in 1 2; do
run "base2 ring16" base2 0
run "base2 ring8" base2 0 DCRX_DEBUG_RING_BATCHES=8
run "base2 ring4" base2 0 DCRX_DEBUG_RING_BATCHES=4
run "base2 ring4 tw6" base2 0 DCRX_DEBUG_RING_BATCHES=4 DCRX_DEBUG_TAIL_WAVES=6
run "base2 scan-only(2)" base2 2
run "skip3 scan-only(2)" skip3 2
run "base2 nofinish(128)" base2 128
run "skip3 nofinish(128)" skip3 128
run "base2 nofuse(131072)" base2 131072
done
