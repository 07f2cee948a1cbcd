# Config 3 (both chains' tail a role of the finishing launch): rescue waves x tail-role waves, at 100 M and 10 M reads per step.
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --config 3 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'G reads/s', round(d['value']/1e3,2))"
}
for rep in 1 2; do run "100M default" --reads 100000000 --steps 5 --warmup 2; done
for e in 3072 4096 6144; do for t in 2048 3072 4096; do
DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_TAIL_ROLE_WAVES=$t run "100M E=C=$e T=$t" --reads 100000000 --steps 5 --warmup 2
done; done
for rep in 1 2; do run "10M default" --steps 30 --warmup 10; done
for e in 3072 4096; do for t in 3072 4096; do
DCRX_DEBUG_RESCUE_WAVES=$e DCRX_DEBUG_TAIL_ROLE_WAVES=$t run "10M E=C=$e T=$t" --steps 30 --warmup 10
done; done
