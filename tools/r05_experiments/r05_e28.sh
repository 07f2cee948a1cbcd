# Tune with one sample per setting: does the choice land inside the driver's five warm-up launches, and what does
# `--steps 20 --warmup 5` read then?  Then the GPU suite on the same library.
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  DCRX_DEBUG_TUNE=1 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep -E "tune|ms_per_step" | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('20/5', d['ms_per_step'], 'frac', d['roofline']['frac'], 'tune', d.get('tune'))
    else: print(l)"
done
python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('50/10', d['ms_per_step'], 'frac', d['roofline']['frac'])"
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
