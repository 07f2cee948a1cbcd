#!/bin/bash
# usage (GPU box): tools/r04_counts.sh TAG LIB[:FLAGS] ... — list populations (DCRX_DEBUG_V2_COUNTS) and a serial trace per spec
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DCRX_DEBUG_FLAGS=1
for spec in "$@"; do
  lib=${spec%%:*}; fl=0; [[ "$spec" == *:* ]] && fl=${spec##*:}
  [ "$lib" = "default" ] && unset DCRX_LIB_PATH || export DCRX_LIB_PATH=$R/$lib
  name=$(basename $lib .so)_$fl
  echo "=== $spec"
  DCRX_DEBUG_V2_COUNTS=1 DCRX_DEBUG_HANDOVER=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --cfg-flags $fl 2>&1 | grep "dcrx" | sort | uniq -c | head -5
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu-baseline --steps 20 --cfg-flags $fl > /dev/null 2>&1
  python3 $R/tools/timeline.py $O/trace_$name | tee $O/timeline_$name.txt
  rm -rf $O/trace_$name
done
