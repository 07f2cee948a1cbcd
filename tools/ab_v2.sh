#!/bin/bash
# A/B of the v2 kernel's launch shapes against the three-launch form (bench.py --cfg-flags):
#   64 = three-launch form; 256/512/768 = v2 shape 1/2/3; +2 = scan only
mkdir -p gpurun_out/ab
for f in ${@:-64 256 512 768 66 258 514 770}; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 --cfg-flags $f > gpurun_out/ab/bench_$f.log 2>&1
  tail -1 gpurun_out/ab/bench_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['roofline']['step_device_ms_avg'], d['roofline']['dominant_kernel_ms_avg'])"
done
