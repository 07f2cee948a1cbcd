#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp
for th in 1024 768 512 384 256; do
  DCRX_SCAN_THREADS=$th python3 $R/bench.py --no-cpu-baseline --steps 30 --cfg-flags 128 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('scan threads $th: no-finish ms_per_step', d['ms_per_step'], 'scan kernel', d['roofline']['dominant_kernel_ms_avg'])"
done
DCRX_EXPECT_GPU=1 timeout 1200 python3 -m pytest $R/tests/test_gpu_parity.py -x -q 2>&1 | tail -3
