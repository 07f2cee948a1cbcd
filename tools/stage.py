"""Whole decombine stage on real files (FASTQ in -> `.n12` rows), on the GPU box: synthetic
human-beta reads (the bench workload's generator) are written as an R1/R2 FASTQ pair, then
decombinator() runs over them; prints reads/s and the phase split.  Not the
benchmark: bench.py times the device-resident hot path."""
import argparse
import gzip
import os
import tempfile
import time
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))   # repo root

import numpy as np

from decombinator_amd import _native as nat, decombine as dec, io as dio, synth, collapse

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4_000_000)
ap.add_argument("--gz", action="store_true")
ap.add_argument("--py-gzip", action="store_true", help="also time the reference's gzip.open step on the same rows")
args = ap.parse_args()

ts = synth.config_tagset(2)
t = nat.Tables(ts.v_tags, ts.v_jumps, ts.v_regions, ts.j_tags, ts.j_jumps, ts.j_regions, *ts.half_splits)
n = args.reads
hb = nat.synth_reads_host(t, nat.synth_cfg(seed=2), 0, n)
reads = nat.unpack_reads(hb)
with tempfile.TemporaryDirectory() as td:
    ts.write(os.path.join(td, "tags"))
    ext = ".fq.gz" if args.gz else ".fq"
    op = (lambda p: gzip.open(p, "wt", compresslevel=1)) if args.gz else (lambda p: open(p, "w"))
    q1, q2 = "I" * 150, "I" * 50
    rng = np.random.default_rng(0)
    # barcode regions of the M13 oligo (spacer, N6, spacer, N6) so that the front half of collapse has UMIs to extract
    bcs = ["GTCGTGACTGGGAAAACCCTGG" + "".join(rng.choice(list("ACGT"), size=6)) + "GTCGTGAT" + "".join(rng.choice(list("ACGT"), size=14)) for _ in range(1024)]
    with op(os.path.join(td, "S_1" + ext)) as f1, op(os.path.join(td, "S_2" + ext)) as f2:
        for i, r in enumerate(reads):
            f1.write(f"@SYN:{i}:1101:{i % 9973}:{i % 7919} 1:N:0:ACGT\n{r}\n+\n{q1}\n")
            f2.write(f"@SYN:{i}:1101:{i % 9973}:{i % 7919} 2:N:0:ACGT\n{bcs[i & 1023]}\n+\n{q2}\n")
    size = os.path.getsize(os.path.join(td, "S_1" + ext)) + os.path.getsize(os.path.join(td, "S_2" + ext))
    a = dio.create_args_dict(infile=os.path.join(td, "S_1" + ext), chain="b", bc_read="R2", dontgzip=True, dontcount=True,
                             dontcheck=True, suppresssummary=True, tagfastadir=os.path.join(td, "tags"),
                             outpath=td + os.sep, command="decombine", tags=ts.tags, species=ts.species)
    for rep in range(2):
        dec.counts.clear()
        t0 = time.perf_counter()
        rows = dec.decombinator(a)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        out = dio.write_out_intermediate(rows, a, ".n12")
        dw = time.perf_counter() - t1
        t1 = time.perf_counter()
        outz = dio.write_out_intermediate(rows, dict(a, dontgzip=False), ".n12")      # the reference's default: gzipped
        dwz = time.perf_counter() - t1
        mb, mbz = os.path.getsize(out) / 1e6, os.path.getsize(outz) / 1e6
        dpy = None
        if rep == 1 and args.py_gzip:          # what the reference's gzip.open step takes on the same text (one thread, level 9)
            t1 = time.perf_counter()
            with open(out) as fi, gzip.open(out + ".py.gz", "wt") as fo:
                fo.writelines(fi)
            dpy = time.perf_counter() - t1
        ph = ", ".join(f"{k} {v:.2f}s" for k, v in dec.stage_seconds.items())
        collapse.counts.clear()
        t2 = time.perf_counter()
        front = collapse.read_in_rows(rows, {"oligo": "m13", "allowNs": False, "lenthreshold": 130}, [20, 1, 30])
        df = time.perf_counter() - t2
        print(f"STAGE reads={n} gz={args.gz} input_MB={size / 1e6:.0f} rows={len(rows)} decombinator={dt:.2f}s "
              f"({n / dt / 1e6:.2f} Mreads/s; {ph}) write_n12={dw:.2f}s write_n12_gz={dwz:.2f}s ({mb:.0f} -> {mbz:.0f} MB{'' if dpy is None else f'; gzip.open level 9: {dpy:.1f}s'}) collapse_front={df:.2f}s "
              f"({len(rows) / max(df, 1e-9) / 1e6:.2f} Mrows/s, {len(front.kept())} rows kept, {int((front.status == 255).sum())} deferred to the regex)")
