# the scan-only launch without record stores / without look-ups / without both -> profiles/r05/scan_only_without_lookups.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; lib=$2; fl=$3; shift 3
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 --cfg-flags $fl "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for rep in 1 2; do
run "base2 scan-only" base2 2
run "norec scan-only" norec 2
run "noloop scan-only" noloop 2
run "noloop+norec scan-only" noloopnorec 2
run "base2 scan-only 20M reads" base2 2 --reads 20000000
run "noloop scan-only 20M reads" noloop 2 --reads 20000000
done
