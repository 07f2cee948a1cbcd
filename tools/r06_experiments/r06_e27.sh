# config 2: one to four batches in flight
R=$GRAFT_REPO_ROOT; cd /tmp
for rep in 1 2 3; do for fl in 1 2 3 4; do
  python3 $R/bench.py --no-cpu-baseline --in-flight $fl 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN in_flight $fl', d['ms_per_step'], d['ms_per_step_steady'], d['value'], d['roofline']['frac'], d['tune']['launch_form'])"
done; done
