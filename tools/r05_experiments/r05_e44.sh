# Every kernel of the last steps of a one-rank run with the gather forced (sink mode): what sits in the 11 us between the place kernel
# and the next step's scan -> profiles/r05/forced_gather_all_kernels_of_a_step.log
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/trg
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 DCRX_BENCH_FORCE_GATHER=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -- python3 $R/bench.py --no-cpu-baseline --steps 12 --warmup 3 --no-gather-ab > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/trg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if 'scan2_kernel' in r['Kernel_Name']]
last = rows[idx[-3] - 3: idx[-1] + 8]
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else f"gap {max(0, s - prev_end) / 1e3:6.1f}"
    print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} dur {(e - s) / 1e3:7.1f} us  {gap}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end or 0, e)
PY
