# Same box, six interleaved repetitions, medians: round 5's kernels | the scan's first item landed before its loop | + the lean
# roles' batch in hand landed before the next is asked for | + the rescue's ticket atomic behind that landing point
# -> profiles/r06/prefetch_really_in_flight_ab2.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e5; mkdir -p $O
(
DCRX_LIB_PATH=$R/tools/variants/libdcrx_landed2.so timeout 600 python3 tests/forced_shape_worker.py 2 2097152 3 2>&1 | tail -1 | cut -c1-200
DCRX_LIB_PATH=$R/tools/variants/libdcrx_landed2.so timeout 600 python3 tests/forced_shape_worker.py 5 2097152 3 2>&1 | tail -1 | cut -c1-200
cd /tmp
run() { n=$1; lib=$2; shift 2
  export DCRX_LIB_PATH=$R/tools/variants/libdcrx_$lib.so
  python3 $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], 'steady', d.get('ms_per_step_steady'), 'scan', d['roofline']['dominant_kernel_ms_avg'], 'rest', round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
}
for rep in 1 2 3 4 5 6; do
for v in r05 scanfix landed landed2; do run "$v" $v; done
done
for rep in 1 2 3; do
for v in r05 scanfix landed2; do DCRX_BENCH_SUB_RATE=0.02 run "sub2pct_$v" $v; done
done
) 2>&1 | tee $O/raw.log
python3 - <<'PY' | tee $O/prefetch_really_in_flight_ab2.log
import statistics,collections,os
d=collections.defaultdict(list)
for l in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_e5/raw.log'):
    p=l.split()
    if len(p)>=8 and p[2]=='steady':
        d[p[0]].append((float(p[1]),float(p[3]),float(p[5]),float(p[7])))
    elif l.startswith('SHAPE_OK'): print(l.strip()[:120])
for k,v in d.items():
    print(f"{k:18s} n={len(v)} ms_per_step median {statistics.median(x[0] for x in v):.4f} (min {min(x[0] for x in v):.4f})  steady {statistics.median(x[1] for x in v):.4f}  scan {statistics.median(x[2] for x in v):.4f}  rest {statistics.median(x[3] for x in v):.4f}")
PY
