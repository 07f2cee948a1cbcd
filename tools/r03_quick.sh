#!/bin/bash
# usage (GPU box): tools/r03_quick.sh — three bench lines, the kernels of a step from a trace, the GPU suite
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/quick/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/quick/trace
rm -rf $R/gpurun_out/quick/trace
cd $R && timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
