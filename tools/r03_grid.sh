#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp
run() { python3 $R/bench.py --no-cpu-baseline --steps 30 --cfg-flags $1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 ms_per_step', d['ms_per_step'])"; }
# lean kernels in series (32768): per-kernel alone times show up in the step time
for t in 0 3 4 5 8; do DCRX_TAIL_BPC=$t run 32768 "serial tail_bpc=$t resc_bpc=0"; done
for t in 3 4 8; do DCRX_TAIL_BPC=4 DCRX_RESC_BPC=$t run 32768 "serial tail_bpc=4 resc_bpc=$t"; done
for t in 0 4; do DCRX_TAIL_BPC=$t DCRX_RESC_BPC=$t run 0 "default tail_bpc=$t resc_bpc=$t"; done
