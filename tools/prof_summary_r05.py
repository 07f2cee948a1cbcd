"""Condenses the rocprofv3 output of tools/prof_r05.sh into the small files kept under profiles/ (round 5: traffic.json carries
the digest of the kernel sources it was measured on — bench.py quotes it only while that digest is the tree's — and the LDS side
of the dominant kernel: its LDS-array cycles over the kernel's cycles on all compute units):
bench lines, kernel_stats.csv (from --kernel-trace --stats), pmc_per_launch.json (every counter of the --pmc
passes, averaged per launch and kernel) and traffic.json (memory-side bytes of the L2 per kernel and per step,
priced as MI355X_MICROARCH.md's HBM section says: 128 B per read request on gfx950 — FETCH_SIZE halves them —,
64 / 32 B per write request by kind).  usage: prof_summary_r02.py <gpurun_out/TAG>"""
import collections
import csv
import glob
import json
import os
import shutil
import hashlib
import sys


# (what the decombine call never runs: the FASTQ reader, row assembly, the collapse front, CDR3 translation)
HOST_ONLY_SOURCES = ("dcrx_fastq.cpp", "dcrx_rows.cpp", "dcrx_collapse.cpp", "dcrx_translate.cpp")


def csrc_digest(root):
    """sha256 over the sources of the library that the decombine call runs (the same function as bench.py's)."""
    d = os.path.join(root, "decombinator_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) and name not in HOST_ONLY_SOURCES:
            h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(tag, "summary")
os.makedirs(out, exist_ok=True)
for name in ("bench_default.log", "bench_under_kernel_trace.log"):
    p = os.path.join(tag, name)
    if os.path.exists(p):
        lines = [ln for ln in open(p) if ln.startswith("{")]
        open(os.path.join(out, name), "w").writelines(lines[-1:])
stats = sorted(glob.glob(os.path.join(tag, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(out, "kernel_stats.csv"))
per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for p in glob.glob(os.path.join(tag, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "dcrx::" not in k or "synth" in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
res = {k: {c: v / max(len(launches[k][c]), 1) for c, v in cs.items()} for k, cs in per.items()}
json.dump({"note": "rocprofv3 --pmc, one pass per counter group (tools/prof_r02.sh), bench.py --steps 3 --warmup 1; values are per launch",
           "kernels": res}, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
traffic = {}
tot_r = tot_w = 0.0
for k, cs in res.items():
    if "TCC_EA0_RDREQ_sum" not in cs:
        continue
    rd = cs["TCC_EA0_RDREQ_sum"] * 128 - cs.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
    w64 = cs.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = w64 * 64 + (cs.get("TCC_EA0_WRREQ_sum", 0) - w64) * 32
    traffic[k] = {"read_bytes": round(rd), "write_bytes": round(wr), "FETCH_SIZE_KiB": cs.get("FETCH_SIZE"), "WRITE_SIZE_KiB": cs.get("WRITE_SIZE"),
                  "requests": {c: round(v) for c, v in cs.items() if c.startswith("TCC_")}}
    tot_r += rd
    tot_w += wr
v2 = any("scan2" in k for k in traffic)
# the LDS side of the dominant kernel: SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over the compute units, SQ_BUSY_CYCLES the
# cycles the kernel kept the chip's shader engines busy (per SE: x 32 on MI355X; the kernel's duration from the trace is the
# plainer divisor and the one used here)
lds = None
scan = [k for k in res if "scan2" in k]
if scan and stats:
    k = scan[0]
    dur_ns = None
    for r in csv.DictReader(open(stats[-1])):
        if "scan2_kernel" in r["Name"]:
            dur_ns = float(r["AverageNs"])
    if dur_ns and "SQ_LDS_IDX_ACTIVE" in res[k]:
        cus, ghz = 256, 2.4
        lds = {"kernel": k, "lds_array_cycles_per_launch": round(res[k]["SQ_LDS_IDX_ACTIVE"]), "of_them_bank_conflicts": round(res[k].get("SQ_LDS_BANK_CONFLICT", 0)),
               "kernel_us": round(dur_ns / 1e3, 1), "compute_units": cus, "clock_ghz_assumed": ghz,
               "frac": round(res[k]["SQ_LDS_IDX_ACTIVE"] / (dur_ns * ghz * cus), 4),
               "note": "LDS-array cycles of all compute units over kernel duration x assumed clock x compute units: the path's real bound is the LDS gather of the scan, not HBM"}
json.dump({"kernels": "v2" if v2 else "v1", "reads_per_launch": 10000000, "read_len": 150, "profile": os.path.basename(tag),
           "method": "bytes = RDREQ x 128 (32-byte requests x 32) + WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32, memory-side requests of the L2 "
                     "(Infinity-Cache hits included), summed over the kernels of one step; FETCH_SIZE (which tallies 128-byte reads at 64 B on gfx950) "
                     "and WRITE_SIZE kept beside them",
           "csrc_sha16": csrc_digest(ROOT), "roofline_lds": lds,
           "per_kernel": traffic, "hbm_read_bytes_per_step": round(tot_r), "hbm_write_bytes_per_step": round(tot_w),
           "hbm_bytes_per_step": round(tot_r + tot_w), "algorithmic_bytes_per_step": 540000000},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
for k, t in traffic.items():
    print("TRAFFIC", k[:60], "read MB", round(t["read_bytes"] / 1e6, 1), "write MB", round(t["write_bytes"] / 1e6, 1))
print("TRAFFIC per step MB", round((tot_r + tot_w) / 1e6, 1))
if stats:
    for r in csv.DictReader(open(stats[-1])):
        if "dcrx" in r["Name"] and "synth" not in r["Name"]:
            print("KSTAT", r["Name"][:64], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
