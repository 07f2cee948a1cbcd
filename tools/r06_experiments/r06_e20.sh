export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_TUNE=1
R=$GRAFT_REPO_ROOT; cd /tmp
export DCRX_LIB_PATH=$R/tools/variants/libdcrx_cur.so
for rep in 1 2 3 4; do
python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN own choice', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], d['tune']['launch_form'])"
grep "dcrx tune: scan" /tmp/err.log | head -2
DCRX_DEBUG_FUSE_E=0 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN role forced', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], d['tune']['launch_form'])"
DCRX_DEBUG_FUSE_E=1 python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN fused forced', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], d['tune']['launch_form'])"
done
python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --config 5 2>/tmp/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN config 5', d['ms_per_step'], d['tune']['launch_form'])"; grep "dcrx tune: scan" /tmp/err.log | head -2
