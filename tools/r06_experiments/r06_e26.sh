# the bench's defaults after round 6's overlap of launches (two batches in flight; a stream per chain) against --in-flight 1 / one stream
R=$GRAFT_REPO_ROOT; cd /tmp
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $1', d['ms_per_step'], d['ms_per_step_steady'], d['value'], d['roofline']['frac'], d['batches_in_flight'] if 'batches_in_flight' in d else d['config'].get('batches_in_flight'), d['tune']['launch_form'], d['roofline']['traffic'])"; }
for rep in 1 2 3; do
python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -n 1 | p "config 2 default"
python3 $R/bench.py --no-cpu-baseline --in-flight 1 2>/dev/null | tail -n 1 | p "config 2 --in-flight 1"
python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -n 1 | p "config 2 --steps 20 --warmup 5"
done
python3 $R/bench.py --no-cpu-baseline --config 3 2>/dev/null | tail -n 1 | p "config 3 default"
python3 $R/bench.py --no-cpu-baseline --config 3 --in-flight 1 2>/dev/null | tail -n 1 | p "config 3 --in-flight 1"
DCRX_BENCH_CHAIN_STREAMS=0 python3 $R/bench.py --no-cpu-baseline --config 3 --in-flight 1 2>/dev/null | tail -n 1 | p "config 3 --in-flight 1, one stream"
python3 $R/bench.py --no-cpu-baseline --config 5 2>/dev/null | tail -n 1 | p "config 5 default"
python3 $R/bench.py --no-cpu-baseline --config 5 --in-flight 1 2>/dev/null | tail -n 1 | p "config 5 --in-flight 1"
DCRX_BENCH_CHAIN_STREAMS=0 python3 $R/bench.py --no-cpu-baseline --config 5 --in-flight 1 2>/dev/null | tail -n 1 | p "config 5 --in-flight 1, one stream"
python3 $R/bench.py --no-cpu-baseline --reads 100000000 --steps 5 --warmup 1 2>/dev/null | tail -n 1 | p "config 2 100 M reads per step"
python3 $R/bench.py 2>/dev/null | tail -n 1 | cut -c1-300
