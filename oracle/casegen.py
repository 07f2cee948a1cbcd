"""Read constructors for parity cases (TEST INFRASTRUCTURE).

Builds reads against a decombinator_amd.synth.TagSet: clean rearrangements,
single-mismatch tags (half-tag rescue), multiple hits, walk failures, filter
triggers, short / N-containing reads and a TINY-like random mixture.  All reads
are produced in the SENSE frame (the frame dcr() scans); `as_fastq_frame` turns
them into what a FASTQ would hold for the default `reverse` orientation.
"""
from __future__ import annotations

import numpy as np

_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def revcomp(s: str) -> str:
    return s.translate(_COMP)[::-1]


def rand_seq(rng, n: int) -> str:
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def substitute(rng, s: str, pos: int) -> str:
    b = "ACGT".index(s[pos]) if s[pos] in "ACGT" else 0
    return s[:pos] + "ACGT"[(b + 1 + int(rng.integers(0, 3))) % 4] + s[pos + 1:]


def rearranged(ts, rng, v: int, j: int, vdel: int, jdel: int, ins: int, vtag_start: int,
               n: int = 150, insert: str | None = None):
    """Sense-frame amplicon.  Returns (read, info) where info locates the pieces."""
    vreg = ts.v_regions[v].upper()
    jreg = ts.j_regions[j].upper()
    tagoff_v = len(vreg) - ts.v_jumps[v]
    start = tagoff_v - vtag_start
    prefix = ""
    if start < 0:
        prefix = rand_seq(rng, -start)
        start = 0
    vpart = vreg[start:len(vreg) - vdel]
    if insert is None:
        insert = rand_seq(rng, ins)
    jpart = jreg[jdel:]
    read = prefix + vpart + insert + jpart
    info = {
        "v_tag_pos": vtag_start,
        "v_end": len(prefix) + len(vpart) - 1,
        "j_start": len(prefix) + len(vpart) + len(insert),
        "j_tag_pos": len(prefix) + len(vpart) + len(insert) + ts.j_jumps[j] - jdel,
    }
    if len(read) < n:
        read += rand_seq(rng, n - len(read))
    return read[:n], info


def mixture_read(ts, rng, n: int = 150, p_rearr: float = 0.45, sub_rate: float = 0.005,
                 n_rate: float = 0.0005):
    """One read of the TINY-like mixture of SURVEY.md §8(d), sense frame."""
    if rng.random() < p_rearr:
        v = int(rng.integers(0, len(ts.v_tags)))
        j = int(rng.integers(0, len(ts.j_tags)))
        read, _ = rearranged(ts, rng, v, j, int(rng.integers(0, 11)), int(rng.integers(0, 13)),
                             int(rng.integers(0, 26)), int(rng.integers(20, 61)), n)
    else:
        read = rand_seq(rng, n)
    if sub_rate > 0:
        k = rng.binomial(len(read), sub_rate)
        for p in rng.choice(len(read), size=k, replace=False) if k else ():
            read = substitute(rng, read, int(p))
    if n_rate > 0 and rng.random() < n_rate and len(read):
        p = int(rng.integers(0, len(read)))
        read = read[:p] + "N" + read[p + 1:]
    return read


def engineered_cases(ts, rng, n: int = 150):
    """(label, sense-frame read, kwargs) triples aimed at each exit path of
    dcr()/vanalysis()/janalysis()/get_*_deletions."""
    cases = []
    nv, nj = len(ts.v_tags), len(ts.j_tags)
    vsplit, jsplit = ts.half_splits

    def add(label, read, **kw):
        cases.append((label, read, kw))

    def rr(v=None, j=None, vdel=None, jdel=None, ins=None, vts=None, nn=n, insert=None):
        v = int(rng.integers(0, nv)) if v is None else v
        j = int(rng.integers(0, nj)) if j is None else j
        vdel = int(rng.integers(0, 11)) if vdel is None else vdel
        jdel = int(rng.integers(0, 13)) if jdel is None else jdel
        ins = int(rng.integers(0, 26)) if ins is None else ins
        vts = int(rng.integers(20, 61)) if vts is None else vts
        r, info = rearranged(ts, rng, v, j, vdel, jdel, ins, vts, nn, insert)
        return r, info, v, j

    # 1. clean rearrangements, every gene at least once
    for v in range(nv):
        r, _, _, _ = rr(v=v)
        add("clean_v", r)
    for j in range(nj):
        r, _, _, _ = rr(j=j)
        add("clean_j", r)
    for vdel in range(0, 14):
        for jdel in (0, 5, 12, 19, 20):
            r, _, _, _ = rr(vdel=vdel, jdel=jdel)
            add("dels", r)
    for ins in (0, 0, 1, 2, 10, 25, 40):
        r, _, _, _ = rr(ins=ins)
        add("ins", r)
    # empty insert with junction bases shared by V and J (decombine.py:796-800)
    for _ in range(10):
        r, _, _, _ = rr(vdel=0, jdel=0, ins=0)
        add("ins0", r)

    # 2. one substitution at each position of the V tag / J tag (half-tag rescue both ways)
    for _ in range(3):
        r0, info, v, j = rr()
        for off in range(len(ts.v_tags[v])):
            add("vtag_sub", substitute(rng, r0, info["v_tag_pos"] + off))
        for off in range(len(ts.j_tags[j])):
            p = info["j_tag_pos"] + off
            if p < len(r0):
                add("jtag_sub", substitute(rng, r0, p))
    # two substitutions in one half (rescue must fail), one in each half
    for _ in range(20):
        r0, info, v, j = rr()
        p = info["v_tag_pos"]
        add("vtag_sub2_same_half", substitute(rng, substitute(rng, r0, p + 1), p + 7))
        add("vtag_sub2_both_halves", substitute(rng, substitute(rng, r0, p + 2), p + vsplit + 3))
        q = info["j_tag_pos"]
        if q + len(ts.j_tags[j]) <= len(r0):
            add("jtag_sub2_same_half", substitute(rng, substitute(rng, r0, q + jsplit), q + jsplit + 3))
            add("jtag_sub2_both_halves", substitute(rng, substitute(rng, r0, q + 1), q + jsplit + 2))

    # 3. tags sharing a half with another tag: every gene, sub in each half
    for v in range(nv):
        r0, info, _, _ = rr(v=v)
        add("vshare_sub_h1", substitute(rng, r0, info["v_tag_pos"] + 3))
        add("vshare_sub_h2", substitute(rng, r0, info["v_tag_pos"] + vsplit + 3))
    for j in range(nj):
        r0, info, _, _ = rr(j=j)
        if info["j_tag_pos"] + len(ts.j_tags[j]) <= len(r0):
            add("jshare_sub_h1", substitute(rng, r0, info["j_tag_pos"] + 2))
            add("jshare_sub_h2", substitute(rng, r0, info["j_tag_pos"] + jsplit + 2))

    # 4. multiple hits
    for _ in range(6):
        r0, info, v, j = rr(vts=30)
        other = ts.v_tags[int(rng.integers(0, nv))]
        add("multi_v", other + r0[len(other):])                       # second V tag at 0
        add("multi_v_same", ts.v_tags[v] + r0[len(ts.v_tags[v]):])    # same tag twice
        tail = ts.j_tags[int(rng.integers(0, nj))]
        add("multi_j", r0[:len(r0) - len(tail)] + tail)                # second J tag at the end
        # bare half tags elsewhere (must not disturb a full hit)
        add("extra_half", ts.v_tags[(v + 1) % nv][:vsplit] + r0[vsplit:])

    # 5. V tag near the read end / walk failures
    for v in range(min(nv, 8)):
        tag = ts.v_tags[v]
        for room in (0, 1, 5, 15, ts.v_jumps[v] - len(tag) - 1, ts.v_jumps[v] - len(tag),
                     ts.v_jumps[v] - len(tag) + 1):
            room = max(0, room)
            vreg = ts.v_regions[v].upper()
            off = len(vreg) - ts.v_jumps[v]
            piece = vreg[off:off + len(tag) + room]
            add("v_at_end", (rand_seq(rng, n) + piece)[-n:])
    for _ in range(10):
        r0, info, v, j = rr(vdel=0)
        # scramble the germline end so that no 10-mer matches: long walk / failure
        e = info["v_end"]
        scr = r0[:e - 25] + rand_seq(rng, 26) + r0[e + 1:]
        add("v_walk_scrambled", scr)
    # V tag at the very start (walk runs into the left edge: negative slices)
    for v in range(min(nv, 6)):
        add("v_at_start", (ts.v_tags[v] + rand_seq(rng, n))[:n])
        add("v_at_start_short", (ts.v_tags[v] + rand_seq(rng, 12)))

    # 6. J problems
    for _ in range(10):
        r0, info, v, j = rr(jdel=0)
        s = info["j_start"]
        add("j_walk_scrambled", r0[:s] + rand_seq(rng, 30) + r0[s + 30:])
        add("no_j", r0[:s] + rand_seq(rng, len(r0) - s))
    for j in range(min(nj, 6)):
        # J tag before any room for the germline start: ts < -2 (appendix A.7 quirk 5)
        r0, info, v, _ = rr(vts=20)
        add("j_tag_too_early", r0)  # control
        jt = ts.j_tags[j]
        for p in (0, 5, 17, 18, 19):
            rr2 = ts.v_tags[v] + rand_seq(rng, 3)
            add("j_early", (rand_seq(rng, p) + jt + rand_seq(rng, n))[:n])
            add("vj_tight", (ts.v_tags[v] + rand_seq(rng, p) + jt + rand_seq(rng, n))[:n])
    # J tag right at the read end: truncated slices in the walk
    for j in range(min(nj, 6)):
        for cut in (0, 1, 2, 3, 8, 12):
            r0, info, v, _ = rr(j=j, vts=20, ins=5)
            end = info["j_tag_pos"] + len(ts.j_tags[j]) - cut
            add("j_at_end", r0[:max(0, end)])

    # 7. filters
    for _ in range(8):
        r0, info, v, j = rr(ins=12)
        p = info["v_end"] + 3
        rN = r0[:p] + "N" + r0[p + 1:]
        add("intertag_N", rN)
        add("intertag_N_allowed", rN, allow_ns=True)
        add("lenthreshold_low", r0, lenthreshold=-60)
        add("lenthreshold_neg_big", r0, lenthreshold=-1000)
    for _ in range(8):
        # impossible deletions: cut far into V (beyond the tag end) but keep the tag
        v = int(rng.integers(0, nv))
        room = ts.v_jumps[v] - len(ts.v_tags[v])
        for extra in (0, 1, 3):
            r0, info, _, _ = rr(v=v, vdel=room + extra, ins=8)
            add("imposs_vdel", r0)
        j = int(rng.integers(0, nj))
        for jd in (ts.j_jumps[j] - 1, ts.j_jumps[j], ts.j_jumps[j] + 1):
            r0, info, _, _ = rr(j=j, jdel=max(0, jd), ins=8)
            add("imposs_jdel", r0)
    # tag overlap: J tag starting inside / right after the V tag
    for _ in range(8):
        v = int(rng.integers(0, nv)); j = int(rng.integers(0, nj))
        for gap in (-5, 0, 1, 4):
            vt, jt = ts.v_tags[v], ts.j_tags[j]
            body = vt[:len(vt) + min(gap, 0)] + rand_seq(rng, max(gap, 0)) + jt
            add("overlap", (rand_seq(rng, 30) + body + rand_seq(rng, n))[:n])

    # J tag in front of the V tag: temp_start_j < -2 (walk never starts) or the walk
    # starts beyond the J germline; V-start > J-end feeds the overlap / length filters
    for _ in range(10):
        v = int(rng.integers(0, nv)); j = int(rng.integers(0, nj))
        vreg = ts.v_regions[v].upper(); off = len(vreg) - ts.v_jumps[v]
        vpiece = vreg[off:]                       # tag + germline end
        jreg = ts.j_regions[j].upper()
        for lead in (0, 10, 17, 18, 19, 20, 25, 60):
            jpiece = jreg[max(0, ts.j_jumps[j] - lead):]
            pre = rand_seq(rng, max(0, lead - ts.j_jumps[j]))
            add("j_before_v", (pre + jpiece + vpiece + rand_seq(rng, n))[:n])
            add("j_before_v_tight", (pre + jreg[max(0, ts.j_jumps[j] - lead):ts.j_jumps[j] + len(ts.j_tags[j])]
                                     + vpiece + rand_seq(rng, 4)))
            add("j_before_v_lowthr", (pre + jpiece + vpiece + rand_seq(rng, n))[:n], lenthreshold=-5)

    # 8. N / non-ACGT content
    for _ in range(6):
        r0, info, v, j = rr()
        for p in (0, info["v_tag_pos"] + 4, info["v_tag_pos"] + 15, info["v_end"] - 3,
                  info["v_end"] + 1, info["j_start"] + 4, info["j_tag_pos"] + 3, len(r0) - 1):
            if 0 <= p < len(r0):
                add("N_at", r0[:p] + "N" + r0[p + 1:])
                add("N_at_allowed", r0[:p] + "N" + r0[p + 1:], allow_ns=True)
        add("lower_n", r0[:40] + "n" + r0[41:])
        add("iupac_R", r0[:info["v_end"] + 2] + "R" + r0[info["v_end"] + 3:])
    add("all_N", "N" * n)
    add("all_N_allowed", "N" * n, allow_ns=True)

    # 9. degenerate lengths
    for L in (0, 1, 2, 3, 9, 10, 11, 19, 20, 21, 29, 30, 31, 40, 60, 100, 149, 151, 200, 251, 300):
        add("len_random", rand_seq(rng, L))
        r0, _, _, _ = rr(nn=max(L, 1), vts=20, ins=3, vdel=2, jdel=2)
        add("len_rearr", r0[:L])
    for v in range(min(nv, 4)):
        add("bare_vtag", ts.v_tags[v])
        add("bare_vtag_plus1", ts.v_tags[v] + "A")
        add("bare_vhalf1", ts.v_tags[v][:vsplit])
        add("bare_vhalf2", ts.v_tags[v][vsplit:])
        add("bare_vhalf2_padded", "AC" + ts.v_tags[v][vsplit:] + "GT")
    # homopolymers / low complexity
    for b in "ACGT":
        add("homopolymer", b * n)
    add("dinuc", "AC" * (n // 2))

    # 10. orientation
    for _ in range(12):
        r0, _, _, _ = rr()
        add("fwd_sense", r0, orientation="forward")
        add("fwd_antisense", revcomp(r0), orientation="forward")
        add("both_sense", r0, orientation="both")
        add("both_antisense", revcomp(r0), orientation="both")
        add("both_random", rand_seq(rng, n), orientation="both")
    return cases
