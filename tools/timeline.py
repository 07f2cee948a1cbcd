#!/usr/bin/env python3
"""usage: tools/timeline.py <rocprofv3 output dir> — the average timeline of a step from the kernel trace: every kernel's start
and end relative to the start of the step's first kernel (us), over the steps of the run."""
import csv, glob, sys, collections
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "dcrx::" not in n or "synth" in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("dcrx::")[1].split("<")[0].split("(")[0]))
rows.sort()
steps, cur = [], []
for s, e, n in rows:
    # a step starts with the prologue (three-launch form) or with the scan that follows a list kernel (v2: no prologue)
    if cur and (n == "prologue_kernel" or (n == "scan2_kernel" and cur[-1][2] == "decombine_list_kernel")):
        steps.append(cur); cur = []
    cur.append((s, e, n))
if cur:
    steps.append(cur)
steps = steps[len(steps) // 3:]          # drop the warm-up third
acc = collections.OrderedDict()
for st in steps:
    t0 = st[0][0]
    seen = collections.Counter()
    for s, e, n in st:
        seen[n] += 1
        key = n if seen[n] == 1 else f"{n}#{seen[n]}"
        a = acc.setdefault(key, [0.0, 0.0, 0])
        a[0] += (s - t0) / 1e3; a[1] += (e - t0) / 1e3; a[2] += 1
span = sum((max(e for _, e, _ in st) - st[0][0]) for st in steps) / len(steps) / 1e3
nxt = [steps[i + 1][0][0] - steps[i][0][0] for i in range(len(steps) - 1)]
print(f"steps {len(steps)}: first kernel start -> last kernel end {span:.1f} us; step period {sum(nxt) / max(1, len(nxt)) / 1e3:.1f} us")
for k, (a, b, c) in acc.items():
    print(f"  {k:28s} start {a / c:7.1f}  end {b / c:7.1f}  dur {(b - a) / c:7.1f}   (n={c})")
