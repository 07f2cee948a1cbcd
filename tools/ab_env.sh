#!/bin/bash
# usage: tools/ab_env.sh FLAGS NAME=VALUE ... (GPU box): one bench line under the given environment and --cfg-flags
f=$1; shift
for kv in "$@"; do export "$kv"; done
python bench.py --no-cpu-baseline --steps 60 --warmup 5 --cfg-flags $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB', '$f', '$*', d['ms_per_step'], d['roofline']['step_device_ms_avg'])"
