# tail waves 2-6 by the share of rearranged reads, tail waves at priority 1 -> profiles/r05/tail_waves_by_share_tail_priority_1.log
R=$GRAFT_REPO_ROOT; cd /tmp; export DCRX_DEBUG_FLAGS=1
for p in 0.15 0.3 0.45 0.6 0.75 0.9; do
  for tw in 2 3 4 5 6; do
    DCRX_BENCH_P_REARRANGED=$p DCRX_DEBUG_TAIL_WAVES=$tw DCRX_LIB_PATH=$R/tools/variants/libdcrx_tp1.so python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('p_rearranged $p tw $tw', d['ms_per_step'], 'scan', d['roofline']['dominant_kernel_ms_avg'], 'decombined', d['config']['decombined_fraction'])"
  done
done
