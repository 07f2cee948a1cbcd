# usage (GPU box): tools/r06_experiments/r06_ab.sh TAG REPS VARIANT ... — per variant library (tools/variants/libdcrx_V.so) a parity
# check (2 M reads, every record and counter against the oracle: tests/forced_shape_worker.py), then the bench round-robin REPS
# times on ONE box and the medians.  BENCH_ARGS: further bench.py arguments; CHECK=0 skips the parity checks.
TAG=$1; REPS=$2; shift 2
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/$TAG; mkdir -p $O
(
if [ "${CHECK:-1}" = "1" ]; then
for v in "$@"; do
  DCRX_LIB_PATH=$R/tools/variants/libdcrx_$v.so timeout 600 python3 tests/forced_shape_worker.py ${CHECK_CONFIG:-2} 2097152 3 2>&1 | tail -1 | cut -c1-160 | sed "s|^|$v: |"
done
fi
cd /tmp
for rep in $(seq 1 $REPS); do
for v in "$@"; do
  export DCRX_LIB_PATH=$R/tools/variants/libdcrx_$v.so
  python3 $R/bench.py --no-cpu-baseline --steps ${STEPS:-100} --warmup 10 $BENCH_ARGS 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $v', d['ms_per_step'], d.get('ms_per_step_steady'), d['roofline']['dominant_kernel_ms_avg'], round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
done
done
) 2>&1 | tee $O/raw.log
python3 - $O/raw.log <<'PY' | tee $O/summary.log
import statistics,collections,sys
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    p=l.split()
    if p and p[0]=='RUN': d[p[1]].append(tuple(float(x) for x in p[2:6]))
    elif 'SHAPE_OK' in l or 'Error' in l or 'rror' in l: print(l.strip()[:150])
for k,v in d.items():
    print(f"{k:14s} n={len(v)} ms_per_step median {statistics.median(x[0] for x in v):.4f} (min {min(x[0] for x in v):.4f})  steady {statistics.median(x[1] for x in v):.4f}  scan {statistics.median(x[2] for x in v):.4f}  rest {statistics.median(x[3] for x in v):.4f}")
PY
