# FUSE_E: list E through an event ring inside the scan kernel (rescue waves beside the tail waves) against list E as a role of the
# finishing launch (DCRX_DEBUG_FUSE_E=0), by rescue waves per block  -> profiles/r06/list_e_in_the_scan_ab.log
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r06_e14; mkdir -p $O
export DCRX_LIB_PATH=$R/tools/variants/libdcrx_fuse_e.so
(
for c in 2 5; do timeout 600 python3 tests/forced_shape_worker.py $c 2097152 3 2>&1 | tail -1 | cut -c1-200; done
DCRX_DEBUG_FUSE_E_WAVES=2 timeout 600 python3 tests/forced_shape_worker.py 2 2097152 2 2>&1 | tail -1 | cut -c1-200
DCRX_DEBUG_FUSE_E_WAVES=5 timeout 600 python3 tests/forced_shape_worker.py 2 2097152 2 2>&1 | tail -1 | cut -c1-200
cd /tmp
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $n', d['ms_per_step'], d.get('ms_per_step_steady'), d['roofline']['dominant_kernel_ms_avg'], round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
}
for rep in 1 2 3; do
DCRX_DEBUG_FUSE_E=0 run "list_E_a_role"
for w in 2 3 4 5; do DCRX_DEBUG_FUSE_E_WAVES=$w run "fused_${w}_rescue_waves"; done
done
for tw in 2 3 4; do DCRX_DEBUG_TAIL_WAVES=$tw DCRX_DEBUG_FUSE_E_WAVES=3 run "fused_3_rescue_${tw}_tail"; done
DCRX_BENCH_SUB_RATE=0.02 DCRX_DEBUG_FUSE_E=0 run "sub2pct_role"
for w in 3 4 5 6; do DCRX_BENCH_SUB_RATE=0.02 DCRX_DEBUG_FUSE_E_WAVES=$w run "sub2pct_fused_$w"; done
) 2>&1 | tee $O/raw.log
python3 - $O/raw.log <<'PY' | tee $O/summary.log
import statistics,collections,sys
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    p=l.split()
    if p and p[0]=='RUN': d[p[1]].append(tuple(float(x) for x in p[2:6]))
    elif 'SHAPE_OK' in l or 'rror' in l: print(l.strip()[:170])
for k,v in d.items():
    print(f"{k:26s} n={len(v)} ms_per_step median {statistics.median(x[0] for x in v):.4f} (min {min(x[0] for x in v):.4f})  scan {statistics.median(x[2] for x in v):.4f}  rest {statistics.median(x[3] for x in v):.4f}")
PY
