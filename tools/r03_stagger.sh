#!/bin/bash
# usage (GPU box): tools/r03_stagger.sh — scan kernel with its waves started apart (DCRX_SCAN_STAGGER x 1024 clocks per wave slot)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_stagger
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for st in 0 2 4 6 8 12; do
  echo "== stagger $st" >> $O/log.txt
  DCRX_SCAN_STAGGER=$st timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 30 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'], d['roofline']['step_device_ms_avg'], d['roofline']['dominant_kernel_ms_avg'])" >> $O/log.txt 2>&1
  DCRX_SCAN_STAGGER=$st timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 30 --cfg-flags 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan-only', d['ms_per_step'], d['roofline']['step_device_ms_avg'], d['roofline']['dominant_kernel_ms_avg'])" >> $O/log.txt 2>&1
done
cat $O/log.txt
