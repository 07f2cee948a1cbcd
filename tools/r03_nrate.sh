#!/bin/bash
# per-kernel durations at several shares of reads with an N (the slow list's population)
export DCRX_DEBUG_FLAGS=1      # (the library honours its DCRX_DEBUG_* switches only with this set)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_nrate
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for nr in 0 0.0005 0.005; do
  DCRX_BENCH_N_RATE=$nr DCRX_DEBUG_V2_COUNTS=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "v2 lists" | tail -1
  DCRX_BENCH_N_RATE=$nr timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$nr -- python3 $R/bench.py --no-cpu-baseline --steps 20 > $O/log$nr.txt 2>&1
  python3 - $O/t$nr $nr <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'dcrx' in r['Name'] and 'synth' not in r['Name']: print("n_rate", sys.argv[2], r['Name'][5:45], 'avg_us', round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf $O/t$nr
done
