#!/bin/bash
# usage: tools_listprof.sh  (on the GPU box): kernel times with and without the list kernel's dcr_frame
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 0 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lst$f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cfg-flags $f > $R/gpurun_out/lst$f.log 2>&1
  for p in $(find $R/gpurun_out/lst$f -name "*kernel_stats.csv"); do echo "flags=$f"; cut -d, -f1-4 $p | cut -c1-150; done
done
