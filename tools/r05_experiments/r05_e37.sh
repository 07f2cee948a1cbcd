# After the tuner learnt 8 192 | 4 096 for big batches: the three configs at 100 M reads per step with a warm-up that lets it settle, then the GPU suite.
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT; cd /tmp
run() { n=$1; shift
  DCRX_DEBUG_TUNE=1 python3 $R/bench.py --no-cpu-baseline --reads 100000000 --steps 5 --warmup 5 "$@" 2>&1 | grep -E "^dcrx tune|^\{" | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$n', d['ms_per_step'], 'G reads/s', round(d['value']/1e3,2), 'frac', d['roofline']['frac'], d.get('tune',{}).get('rescue_waves'), d.get('tune',{}).get('samples_us'))
    else: print('  '+l)"
}
for rep in 1 2; do run "cfg2 100M" ; run "cfg3 100M" --config 3; run "cfg5 100M" --config 5; done
cd $R; python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
