# Instruction counts of the finishing launch per experiment build (exact, unlike timings): where its vector instructions go.
# usage (GPU box): tools/r05_pmc_variants.sh NAME...   (tools/variants/libdcrx_NAME.so; fast builds: config 2's launch shape)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
for v in "$@"; do
  export DCRX_LIB_PATH=$R/tools/variants/libdcrx_$v.so
  rm -rf /tmp/pmcv_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/pmcv_$v -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pmcv_$v.log 2>&1
  python3 - "$v" <<'PY'
import collections, csv, glob, sys
v = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for p in glob.glob(f"/tmp/pmcv_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "finish2" not in k and "scan2" not in k: continue
        k = "finish2" if "finish2" in k else "scan2"
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(per):
    print(v, k, {c: round(x / max(len(n[k][c]), 1) / 1e6, 3) for c, x in sorted(per[k].items())}, "(millions per launch)")
PY
done
