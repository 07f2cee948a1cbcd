# Instruction counts of the scan kernel by form (exact): scan only (cfg flag 2), the tail outside (131072: a role of the finishing launch), fused.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp DCRX_DEBUG_FLAGS=1 DCRX_BENCH_NO_CHECK=1
for spec in fused:0 tail_as_a_role:131072 scan_only:2; do
  name=${spec%%:*}; fl=${spec##*:}
  rm -rf /tmp/pmcf_$name
  if [ $fl = 0 ]; then extra=""; else extra="--cfg-flags $fl"; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/pmcf_$name -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 $extra > /tmp/pmcf_$name.log 2>&1
  python3 - "$name" <<'PY'
import collections, csv, glob, sys
v = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for p in glob.glob(f"/tmp/pmcf_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "dcrx::" not in k or "synth" in k: continue
        k = k.split("<")[0].replace("void ", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(per):
    print(v, k, {c: round(x / max(len(n[k][c]), 1) / 1e6, 3) for c, x in sorted(per[k].items())}, "(millions per launch)")
PY
done
