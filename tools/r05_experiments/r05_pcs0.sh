# Exploratory: does PC sampling work on this pool's MI355X, and what do its files look like?
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pcs0
timeout 60 rocprofv3 --list-avail 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60 > $R/gpurun_out/pcs0/avail.txt
for m in host_trap stochastic; do
  if [ $m = host_trap ]; then unit=time; iv=50; else unit=cycles; iv=65536; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv --output-format csv -d /tmp/pcs_$m -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/pcs0/run_$m.log 2>&1
  echo "rc $?" >> $R/gpurun_out/pcs0/run_$m.log
  find /tmp/pcs_$m -type f | head -20 >> $R/gpurun_out/pcs0/run_$m.log
  for f in $(find /tmp/pcs_$m -type f -name "*.csv"); do echo "== $f $(wc -l < $f) lines"; head -5 $f; done >> $R/gpurun_out/pcs0/run_$m.log 2>&1
done
