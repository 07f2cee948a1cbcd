# Configs 5 and 3 at 100 M reads per step: the launch knobs once more at this size (the defaults were set at 10 M).
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --reads 100000000 --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'G reads/s', round(d['value']/1e3,2), 'scan', d['roofline']['dominant_kernel_ms_avg'], d.get('tune',{}).get('rescue_waves'))"
}
run "cfg5 default" --config 5
run "cfg5 default" --config 5
for tw in 2 3 4; do DCRX_DEBUG_TAIL_WAVES=$tw run "cfg5 tail waves $tw" --config 5; done
for w in 2048 3072 4096 6144 8192; do DCRX_DEBUG_RESCUE_WAVES=$w run "cfg5 E=C=$w" --config 5; done
for c in 1024 2048; do DCRX_DEBUG_RESCUE_WAVES=4096 DCRX_DEBUG_RESCUE_WAVES_C=$c run "cfg5 E 4096 C $c" --config 5; done
run "cfg3 default" --config 3
run "cfg3 default" --config 3
for w in 2048 3072 4096 6144; do DCRX_DEBUG_RESCUE_WAVES=$w run "cfg3 E=C=$w" --config 3; done
for w in 3072 6144 8192; do DCRX_DEBUG_TAIL_ROLE_WAVES=$w run "cfg3 tail role waves $w" --config 3; done
