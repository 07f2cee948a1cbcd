#!/bin/bash
# usage (GPU box): tools/r03_env_ab.sh VAR — bench with and without VAR=1 in the environment, interleaved, then the timeline with it
R=$GRAFT_REPO_ROOT
V=$1
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  unset $V
  echo -n "default: "; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
  export $V=1
  echo -n "$V: "; timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/envab/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/envab/trace
rm -rf $R/gpurun_out/envab/trace
