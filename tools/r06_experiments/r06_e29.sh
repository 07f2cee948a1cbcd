# the profile set's three headline lines once more with the final bench.py (the line carries bench.py's digest)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final_lines; mkdir -p $O; cd /tmp
timeout 600 python3 $R/bench.py > $O/bench_default.out 2>$O/bench_default.err; grep "^{" $O/bench_default.out | tail -1 > $O/bench_default.log
timeout 600 python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_again_same_box.log
timeout 600 python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config2_steps20_warmup5.log
timeout 600 python3 $R/bench.py --no-cpu-baseline --in-flight 1 2>/dev/null | tail -1 > $O/bench_one_batch_in_flight.log
for f in $O/*.log; do tail -n 1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$f'.split('/')[-1], d['ms_per_step'], d.get('ms_per_step_steady'), d.get('ms_per_step_one_batch_in_flight'), d['value'], r['frac'], r['traffic'], r.get('roofline_lds') and r['roofline_lds'].get('frac'), d['tune']['launch_form'], d.get('cpu_baseline',{}).get('value'))"; done
tail -n 3 $O/bench_default.err
