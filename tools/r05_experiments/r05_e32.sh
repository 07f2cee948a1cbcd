# What the finishing launch takes with nothing to do: its blocks leave at once (fempty), stage their tables and leave (fstage), stream
# the entries without resolving any (rnoop; + list X left alone: rnoopnox), against the shipped form (cur) — per wave count.
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
run() { n=$1; lib=$2; shift 2
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 30 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))"
}
for rep in 1 2; do
for v in cur fempty fstage rnoopnox rnoop; do run "$v" $v; done
done
for w in 1024 2048 3072 4096; do
for v in fempty fstage rnoopnox; do DCRX_DEBUG_RESCUE_WAVES=$w run "$v E=C=$w waves" $v; done
done
