#!/bin/bash
# usage (GPU box): tools/r04_quick.sh TAG — after a kernel change: the GPU test suite, then on ONE box the bench in its shipped
# form (roles of one launch), in round 3's shape (side streams, --cfg-flags 65536) and with separate launches in series (32768),
# and a kernel trace of the shipped form
TAG=${1:-r04_quick}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for fl in 0 65536 32768; do
    DCRX_DEBUG_FLAGS=1 timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 40 --cfg-flags $fl 2>$O/bench_$fl.err | tail -1 > $O/bench_$fl.json
    python3 -c "import sys,json; d=json.loads(open('$O/bench_$fl.json').read()); print('flags $fl ms_per_step', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'step_dev', d['roofline']['step_device_ms_avg'])" || tail -5 $O/bench_$fl.err
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 > $O/bench_traced.log 2>&1
python3 $R/tools/timeline.py $O/trace | tee $O/timeline.txt
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/trace
