#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp
for st in 0 65536; do
  echo "== DCRX_SCAN_STAGGER=$st"
  DCRX_SCAN_STAGGER=$st DCRX_LIB_PATH=$R/tools/variants/libdcrx_stamps.so python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --cfg-flags 128 2>&1 | grep -E "stamps" | tail -1
  DCRX_SCAN_STAGGER=$st python3 $R/bench.py --no-cpu-baseline --steps 30 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'], d['roofline']['step_device_ms_avg'], d['roofline']['dominant_kernel_ms_avg'])"
  DCRX_SCAN_STAGGER=$st python3 $R/bench.py --no-cpu-baseline --steps 30 --cfg-flags 2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan-only', d['ms_per_step'], d['roofline']['step_device_ms_avg'], d['roofline']['dominant_kernel_ms_avg'])"
done
