#!/usr/bin/env python3
"""usage: tools/trace_step.py <rocprofv3 output dir> [step] — every kernel (dcrx and others: copies, fills, RCCL) of one
step of the run, start and end relative to the step's prologue kernel (us), in start order."""
import csv, glob, sys
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        short = n.split("dcrx::")[1].split("<")[0].split("(")[0] if "dcrx::" in n else n.split("(")[0][-60:]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
rows.sort()
pro = [i for i, r in enumerate(rows) if r[2] == "prologue_kernel"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(pro) * 2 // 3
lo, hi = pro[k], pro[k + 2] if k + 2 < len(pro) else len(rows)
t0 = rows[lo][0]
for s, e, n, q in rows[lo:hi]:
    print(f"  q{q:>3s} {n:60s} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f}")
