"""The lean rescue's loop trips on the bench's own reads, counted on the CPU (tests/host_emul built with -DDCRX_R2_PROFILE).

A wave of the finishing launch works on 64 consecutive entries of list E in lock step: every loop runs as often as the lane
that needs it most.  This tool runs config 2's synthetic reads through the host build of the device functions, takes the trips
of each entry's loops, puts the entries into waves of 64 in read order (the order the scan's waves push them in, near enough)
and prints per loop: the trips a lane needs on average, and the trips its wave makes.

    python tools/r2_profile.py [reads]        (default 400 000)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from decombinator_amd import _native as nat, synth          # noqa: E402
from tests import parity_util as pu                          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    d = os.path.join(ROOT, "tests", "host_emul")
    out = os.path.join(d, "build", "libdcrx_emul_prof.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-DDCRX_R2_PROFILE", "-Wno-unknown-pragmas", "-shared", "-o", out,
                           os.path.join(d, "emul.cpp"), os.path.join(ROOT, "decombinator_amd", "csrc", "dcrx_tables.cpp")])
    L = C.CDLL(out)
    L.emul_decombine.restype = C.c_int
    L.emul_decombine.argtypes = [C.POINTER(nat.TagSetC), C.POINTER(nat.CfgC), C.POINTER(nat.BatchC), C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    L.emul_r2_trace.restype = C.c_uint64
    L.emul_r2_trace.argtypes = [C.c_void_p, C.c_uint64]
    x = synth.config_tagset(2)
    tables = nat.Tables(x.v_tags, x.v_jumps, x.v_regions, x.j_tags, x.j_jumps, x.j_regions, *x.half_splits)
    ts = dict(v_tags=x.v_tags, v_jumps=x.v_jumps, v_regions=x.v_regions, j_tags=x.j_tags, j_jumps=x.j_jumps, j_regions=x.j_regions,
              v_half_split=x.half_splits[0], j_half_split=x.half_splits[1])
    tsc, keep = pu.tagset_c(ts)
    cfg_s = nat.synth_cfg(seed=1234, read_len=150, p_rearranged=0.45, sub_rate=0.005, n_rate=0.0005)
    batch = nat.synth_reads_host(tables, cfg_s, 0, n)
    cfg = nat.make_cfg("reverse", False, 130, 0)
    rec = np.zeros(n, dtype=nat.RECORD_DTYPE)
    cnt = np.zeros(nat.N_COUNTERS, dtype=np.uint64)
    b = batch.as_c()
    err = C.create_string_buffer(512)
    rc = L.emul_decombine(C.byref(tsc), C.byref(cfg), C.byref(b), rec.ctypes.data, cnt.ctypes.data, err, 512)
    assert rc == 0, (rc, err.value)
    m = int(L.emul_r2_trace(None, 0))
    tr = np.zeros(m, dtype=np.int32)
    L.emul_r2_trace(tr.ctypes.data, m)
    tr = tr.reshape(-1, 2)

    # ---- per entry: the tree of trips ----
    entries = []          # dict(shape, status, sweeps=[ [ [pairs of word: [ (y trips, [cand trips per hit]) ]] ] ], h2=[cand trips])
    cur = None
    for what, a in tr:
        if what == 0:
            cur = dict(read=int(a), shape=None, status=None, sweeps=[])
            entries.append(cur)
        elif what == 9:
            cur["shape"] = int(a)
        elif what == 1:
            cur["sweeps"].append(dict(words=[], h2=[]))
        elif what == 2:
            cur["sweeps"][-1]["words"].append([])
        elif what == 3:
            cur["sweeps"][-1]["words"][-1].append(dict(y=0, hits=[]))
        elif what == 4:
            cur["sweeps"][-1]["words"][-1][-1]["y"] += 1
        elif what == 8:
            cur["sweeps"][-1]["words"][-1][-1]["hits"].append(0)
            cur["_last"] = cur["sweeps"][-1]["words"][-1][-1]["hits"]
        elif what == 6:
            cur["sweeps"][-1]["h2"].append(0)
            cur["_last"] = cur["sweeps"][-1]["h2"]
        elif what == 5:
            cur["_last"][-1] += 1
        elif what == 7:
            cur["status"] = int(a)
    one = [e for e in entries if e["shape"] == 0 and len(e["sweeps"]) == 1]
    both = [e for e in entries if e["shape"] == 1]
    print(f"{n} reads: {len(entries)} clean event entries ({len(entries) / n:.3%}), shape ONE {len(one)}, BOTH {len(both)}; "
          f"settled {sum(e['status'] >= 0 for e in entries)}, OK {sum(e['status'] == 0 for e in entries)}")

    def lane_stats(e):
        s = e["sweeps"][0]
        words = len(s["words"])
        pairs = sum(len(w) for w in s["words"])
        ys = sum(p["y"] for w in s["words"] for p in w)
        hits = sum(len(p["hits"]) for w in s["words"] for p in w)
        cands = sum(sum(p["hits"]) for w in s["words"] for p in w)
        return words, pairs, ys, hits, cands, len(s["h2"]), sum(s["h2"])

    ls = np.array([lane_stats(e) for e in one], dtype=np.float64)
    names = ["words (outer trips)", "flagged pairs (inner trips)", "positions looked up (y trips)", "half-1 hits", "half-1 candidate trips", "half-2 tries", "half-2 candidate trips"]
    print("shape ONE, per entry (what a lane needs):")
    for k, nm in enumerate(names):
        print(f"  {nm:34s} mean {ls[:, k].mean():.3f}  max {ls[:, k].max():.0f}   hist {np.bincount(ls[:, k].astype(int))[:8].tolist()}")

    # ---- per wave of 64: the trips in lock step ----
    def wave_trips(es):
        # outer loop: trip t exists while any lane has a t-th word; inside, the pair loop runs max over lanes of that word's pairs;
        # per pair trip the y loop runs twice (max over lanes), each y trip = one lock-step look-up pair; a candidate call runs
        # when any lane has a hit in that y trip: its loop = the longest lane's
        outer = max(len(e["sweeps"][0]["words"]) for e in es)
        inner = 0
        ytr = 0
        cand_calls = 0
        cand_trips = 0
        for t in range(outer):
            ws = [e["sweeps"][0]["words"][t] for e in es if len(e["sweeps"][0]["words"]) > t]
            it = max(len(w) for w in ws)
            inner += it
            for i in range(it):
                ps = [w[i] for w in ws if len(w) > i]
                yy = max(p["y"] for p in ps)
                ytr += yy
                # (which y trip a hit falls into is not in the trace: at most one call per hit index)
                hmax = max(len(p["hits"]) for p in ps)
                for h in range(hmax):
                    cand_calls += 1
                    cand_trips += max(p["hits"][h] for p in ps if len(p["hits"]) > h)
        h2 = max(len(e["sweeps"][0]["h2"]) for e in es)
        h2t = 0
        for i in range(h2):
            h2t += max(e["sweeps"][0]["h2"][i] for e in es if len(e["sweeps"][0]["h2"]) > i)
        flat = max(sum(len(w) for w in e["sweeps"][0]["words"]) for e in es)
        return outer, inner, ytr, cand_calls, cand_trips, h2, h2t, flat

    waves = [one[i:i + 64] for i in range(0, len(one) - 63, 64)]
    wt = np.array([wave_trips(w) for w in waves], dtype=np.float64)
    wn = ["outer trips", "pair trips (sum over outer of the longest lane's)", "y trips (look-up pairs)", "half-1 candidate calls", "half-1 candidate trips",
          "half-2 tries", "half-2 candidate trips", "pair trips if ONE loop ran over a lane's pairs"]
    print(f"shape ONE, per wave of 64 entries in read order ({len(waves)} waves):")
    for k, nm in enumerate(wn):
        print(f"  {nm:52s} mean {wt[:, k].mean():.2f}  max {wt[:, k].max():.0f}")
    # V sweeps and J sweeps side by side in one wave: how many waves hold both
    nv = sum(1 for w in waves if any(e["status"] is not None for e in w))
    _ = nv


if __name__ == "__main__":
    main()
