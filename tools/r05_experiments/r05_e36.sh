# Rescue waves by batch size: configs 2 and 5 at 30 M and 100 M reads per step.
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'G reads/s', round(d['value']/1e3,2), 'scan', d['roofline']['dominant_kernel_ms_avg'])"
}
for reads in 30000000 100000000; do
for cfg in 2 5; do
run "cfg$cfg $reads default" --config $cfg --reads $reads
for w in 3072 4096 6144 8192 12288 16384; do DCRX_DEBUG_RESCUE_WAVES=$w run "cfg$cfg $reads E=C=$w" --config $cfg --reads $reads; done
done
done
for w in 8192 12288; do DCRX_DEBUG_RESCUE_WAVES=$w run "cfg3 100M E=C=$w" --config 3 --reads 100000000; done
