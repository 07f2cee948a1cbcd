"""Condenses the rocprofv3 output of tools/prof.sh into the small files kept under profiles/:
kernel_stats.csv (from --kernel-trace --stats) and pmc_per_launch.json (every counter of the
--pmc passes, averaged per launch and kernel).  usage: tools_prof_summary.py <gpurun_out/TAG>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
out = os.path.join(tag, "summary")
os.makedirs(out, exist_ok=True)
for name in ("bench.log", "bench_scanonly.log"):
    if os.path.exists(os.path.join(tag, name)):
        lines = [ln for ln in open(os.path.join(tag, name)) if ln.startswith("{")]
        open(os.path.join(out, name), "w").writelines(lines[-1:])
stats = sorted(glob.glob(os.path.join(tag, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(out, "kernel_stats.csv"))
per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for p in glob.glob(os.path.join(tag, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "dcrx::" not in k or "synth" in k:
            continue
        k = k.split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {k: {c: v / max(len(launches[k][c]), 1) for c, v in cs.items()} for k, cs in per.items()}
json.dump({"note": "rocprofv3 --pmc, separate passes (tools/prof.sh), bench.py --steps 3 --warmup 1; values are per launch",
           "kernels": res}, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
        print(k, "FETCH_SIZE_KiB", cs.get("FETCH_SIZE"), "WRITE_SIZE_KiB", cs.get("WRITE_SIZE"))
