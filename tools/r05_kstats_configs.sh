#!/bin/bash
# usage (GPU box): tools/r05_kstats_configs.sh — rocprofv3 --kernel-trace --stats of configs 3 and 5 at 100 M reads per step
# (the kernels of both chains per step), the statistics kept under gpurun_out/r05_kstats/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_kstats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 3 5; do
  rm -rf /tmp/ks$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$c -- python3 $R/bench.py --no-cpu-baseline --config $c --reads 100000000 --steps 5 --warmup 5 > $O/bench_config${c}_100M_under_trace.log 2>&1
  f=$(find /tmp/ks$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/kernel_stats_config${c}_100M.csv <<'PY'
import csv, sys
print("kernel,calls,average_us,min_us,max_us,percentage")
for r in csv.DictReader(open(sys.argv[1])):
    if "dcrx::" in r["Name"] and "synth" not in r["Name"]:
        print(",".join(['"' + r["Name"].split("(")[0].replace("void ", "") + '"', r["Calls"], "%.1f" % (float(r["AverageNs"]) / 1e3), "%.1f" % (float(r["MinNs"]) / 1e3), "%.1f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]]))
PY
  tail -1 $O/bench_config${c}_100M_under_trace.log | cut -c1-200
done
