#!/bin/bash
# usage (GPU box): tools/r03_ab2.sh LIB_A LIB_B ... — bench lines through several builds of the library, interleaved, on ONE box;
# then the GPU parity tests through the default build
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in "$@"; do
  [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
  echo -n "$(basename $lib .so): "
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'])"
done; done
unset DCRX_LIB_PATH
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab2/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 > /dev/null 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/ab2/trace
rm -rf $R/gpurun_out/ab2/trace
cd $R && timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
