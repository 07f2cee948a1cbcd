# FUSE_E against list E as a role, interleaved: config 2, config 2 at 2 % substitutions, config 5
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/r06_e16; mkdir -p $O
export DCRX_LIB_PATH=$R/tools/variants/libdcrx_fuse_e.so
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $n', d['ms_per_step'], d.get('ms_per_step_steady'), d['roofline']['dominant_kernel_ms_avg'], round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
}
(
for rep in 1 2 3 4; do
DCRX_DEBUG_FUSE_E=0 run "c2_role"
DCRX_DEBUG_X_BSPLIT=2 run "c2_fused_3"
DCRX_DEBUG_X_BSPLIT=2 DCRX_DEBUG_TAIL_WAVES=2 run "c2_fused_3_tail2"
DCRX_BENCH_SUB_RATE=0.02 DCRX_DEBUG_FUSE_E=0 run "c2sub2_role"
DCRX_BENCH_SUB_RATE=0.02 DCRX_DEBUG_X_BSPLIT=2 DCRX_DEBUG_FUSE_E_WAVES=5 run "c2sub2_fused_5"
DCRX_DEBUG_FUSE_E=0 run "c5_role" --config 5
DCRX_DEBUG_X_BSPLIT=2 DCRX_DEBUG_FUSE_E_WAVES=4 run "c5_fused_4" --config 5
DCRX_DEBUG_X_BSPLIT=2 DCRX_DEBUG_FUSE_E_WAVES=5 run "c5_fused_5" --config 5
done
) 2>&1 | tee $O/raw.log
python3 - $O/raw.log <<'PY' | tee $O/summary.log
import statistics,collections,sys
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    p=l.split()
    if p and p[0]=='RUN': d[p[1]].append(tuple(float(x) for x in p[2:6]))
for k,v in d.items():
    print(f"{k:22s} n={len(v)} ms_per_step median {statistics.median(x[0] for x in v):.4f} (min {min(x[0] for x in v):.4f}, max {max(x[0] for x in v):.4f})  scan {statistics.median(x[2] for x in v):.4f}  rest {statistics.median(x[3] for x in v):.4f}")
PY
