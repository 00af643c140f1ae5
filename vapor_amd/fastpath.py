"""The four simple SV types of a chunk of loci in array form - what vapor_simple_del_Vapor, vapor_simple_inv_Vapor,
vapor_simple_tandup_Vapor and vapor_simple_ins_Vapor (SF:1701-1745, 1895-1933, 1747-1784, 1856-1893) do for ONE locus, done
for all loci of a chunk at once when they take the drivers' straight route:

    read selection      one native call for all regions (the backend's chop_many: chop_pacbio_read_by_pos +
                        minimize_pacbio_read_list, SF:339-354, 1091-1102); a read stays a slice of its record's sequence -
                        (address, length) - and is never made a Python string
    windows             one reference window per locus read as text; the alt allele, the insertion's `ref + ins_seq` window
                        and the str.upper() twins of abs_dis_m1b are segment descriptors of it (vapor_seqset_create_derived)
    one sequence set    per chunk, one plan for every self dot plot window_size_refine looks at (SF:2030-2046), one plan
                        for the scores (the plan joins a read once against its window and the alleles derived from it)

`vapor_amd.drivers` (generators, one locus each) remains the statement of the reference's control flow and the route of every
locus that leaves the straight one: a span of 10 kb or more, a window that is refused or repetitive ('Error', a growing window
size), an inversion or duplication without enough spanning reads (their junction-window fallbacks, SF:1918-1932, 1769-1783), a
read with N or a character outside the alphabet, windows cut short by a contig end, several BAM files.  Such a locus is
answered FALLBACK here and scored by its generator - with the same result by construction; tests/test_fastpath.py compares
the two routes locus for locus.
"""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np

from . import _lib as L
from . import seqio

FALLBACK = object()
DEFAULT_MAX_SV_TEST = 10000      # SF:25-26
_KIND = {"DEL": 0, "INV": 1, "TANDUP": 2, "INS": 3}
_NOCOMP = re.compile("[^ACGTNacgtn]")       # what complementary() drops (SF:471-478)


def _is_upper(s: str) -> bool:
    return s.isupper() or s.upper() == s


def _window_traits(w: str):
    """(is upper case as abs_dis_m1b's twin rule asks, holds a character complementary() drops): one pass of C over the text for
    the usual window (upper-case ACGTN only), the exact tests only for the others."""
    rest = w.encode("ascii").translate(None, b"ACGTN")
    if not rest:
        return True, False
    return _is_upper(w), bool(rest.translate(None, b"acgtn"))


def capable(backend, bam_in, engine=None) -> bool:
    """The fast route needs a backend that selects reads for many regions at once and hands windows over as text, a
    single alignment file (bam_in_decide's per-chromosome patterns, SF:69-89, go the drivers' way), and an engine that takes
    sequences by address."""
    return (hasattr(backend, "chop_many") and hasattr(backend, "fetch_seq") and backend.isfile(bam_in)
            and (engine is None or hasattr(engine, "seqset_raw")))


def run(engine, specs: Sequence[tuple], bam_in: str, ref: str, num_reads_cff: int) -> List[object]:
    """specs: (type, chrom, start, end, ins_seq) per locus (`end` unused for INS, `start` its position).  Returns per locus the
    list of read scores the driver would return, or FALLBACK."""
    held: list = []                       # device batches of the read selection: closed on this thread on every way out
    try:
        return _run(engine, specs, bam_in, ref, num_reads_cff, held)
    finally:
        for bt in held:
            bt.close()


def _run(engine, specs, bam_in, ref, num_reads_cff, held) -> List[object]:
    import os as _os
    import time as _time
    _dbg = _os.environ.get("VAPOR_DEBUG_FASTPATH")
    _t = [_time.perf_counter()]

    def _mark(tag):
        if _dbg:
            _t.append(_time.perf_counter())
            _marks.append((tag, _t[-1] - _t[-2]))
    _marks = []
    from . import pipeline
    from .engine import _ASCII_OFF
    be = seqio.get_backend()
    n = len(specs)
    out: List[object] = [FALLBACK] * n
    if n == 0 or not (0 < _ASCII_OFF < 256):
        return out
    kind = np.fromiter((_KIND.get(sp[0], -1) for sp in specs), dtype=np.int64, count=n)
    s0 = np.fromiter((int(sp[2]) for sp in specs), dtype=np.int64, count=n)
    e0 = np.fromiter((int(sp[3]) if sp[0] != "INS" else int(sp[2]) + len(sp[4]) for sp in specs), dtype=np.int64, count=n)
    ilen = np.fromiter((len(sp[4]) if sp[0] == "INS" else 0 for sp in specs), dtype=np.int64, count=n)
    span = e0 - s0
    flank = np.where(kind == 3, np.minimum(ilen, 500), np.where(span < 500, span, 500))       # SF:794-802, 1862
    ok = (kind >= 0) & (span >= 1) & (flank >= 1) & ((kind == 3) | (span < DEFAULT_MAX_SV_TEST)) & (s0 - flank >= 1)
    # ---- reads: the region every driver hands to chop_pacbio_read_by_pos --------------------------------------------------
    r_start = s0 - flank
    r_end = np.where(kind == 0, s0 + flank, np.where(kind == 2, s0 + 2 * span + flank, e0 + flank))
    idx = np.flatnonzero(ok)
    if len(idx) == 0:
        return out
    chroms = [specs[t][1] for t in idx.tolist()]
    # reads by device address (4-bit bases in the inflated blocks of a vapor_bam_chop_device batch) where the backend and the
    # engine do that, else by host address (ASCII)
    on_device = False
    kf = None
    refw_of = {}                         # locus -> its reference window as text, read ahead while the device extracts the reads
    if hasattr(be, "chop_many_device") and hasattr(engine, "bam_chop_device"):
        # The extraction is a native call that waits for the device (it releases the interpreter lock): on a helper thread, while
        # this one reads the loci's reference windows - the text work of the loop below that does not need to know the reads.
        # (The context is still used by one thread at a time: this one does not touch the engine until the helper is back.)
        import threading
        box = {}

        def extract():
            try:
                box["got"] = be.chop_many_device(engine, bam_in, chroms, r_start[idx], r_end[idx], flank[idx])
            except BaseException as e:       # noqa: BLE001 - handed to the calling thread below
                box["err"] = e
        th = threading.Thread(target=extract)
        th.start()
        try:
            for t in idx.tolist():
                sp = specs[t]
                f, s_ = int(flank[t]), int(s0[t])
                if int(kind[t]) == 3:
                    m = len(sp[4])
                    refw_of[t] = be.fetch_seq(ref, sp[1], s_ - f, s_ + f + m if m < 5000 else s_ + f)
                else:
                    refw_of[t] = be.fetch_seq(ref, sp[1], s_ - f, int(e0[t]) + f)
        except Exception:                    # noqa: BLE001 - the loop below reads the window again and meets the same error where the drivers would
            pass
        finally:
            th.join()
        _mark("extract+windows")
        if "got" in box:
            kf, addr, q0, miss, status, keepalive = box["got"]
            held.extend(keepalive)
            on_device = True
        elif not isinstance(box.get("err"), NotImplementedError):
            raise box["err"]
    if kf is None:
        try:
            kf, addr, q0, miss, status, keepalive = be.chop_many(bam_in, chroms, r_start[idx], r_end[idx], flank[idx])
        except NotImplementedError:
            return out
    n_reads = np.diff(kf).astype(np.int64)
    # (a read that starts before its record: Python's negative slice - the drivers' way)
    neg = np.zeros(len(idx), dtype=bool)
    if len(q0):
        np.logical_or.at(neg, np.repeat(np.arange(len(idx)), n_reads), q0 < 0)
    good = (status == 0) & ~neg
    enough = n_reads > num_reads_cff
    # deletions and insertions without enough reads are done: [] (SF:1707, 1866); inversions and duplications go on to their
    # junction-window fallbacks in that case (SF:1918, 1769): the drivers' route
    for j in np.flatnonzero(good & ~enough & ((kind[idx] == 0) | (kind[idx] == 3))).tolist():
        out[int(idx[j])] = []
    live = good & enough
    sel = np.flatnonzero(live)
    if len(sel) == 0:
        return out
    # ---- windows as text (one per locus), alleles as descriptors ------------------------------------------------------------
    lit_addr: List[int] = []
    lit_len: List[int] = []
    keep: List[str] = []
    seg_rows: List[tuple] = []           # (parent literal, off, len, flags) of all derived sequences, in order
    seg_first: List[int] = [0]
    dflags: List[int] = []

    def lit(text: str) -> int:
        keep.append(text)
        lit_addr.append(id(text) + _ASCII_OFF)
        lit_len.append(len(text))
        return len(lit_addr) - 1

    def der(segs, upper=False) -> int:
        seg_rows.extend(segs)
        seg_first.append(len(seg_rows))
        dflags.append(L.SEQ_UPPER if upper else 0)
        return -len(dflags)                # (placed behind the literals below)

    loc = []                               # per live locus: what the scoring needs
    alt_texts = {}                         # TANDUP: the alt window as text (its self plot always meets the X-means branch)
    for j in sel.tolist():
        t = int(idx[j])
        sp = specs[t]
        k_, f, s_, e_ = int(kind[t]), int(flank[t]), int(s0[t]), int(e0[t])
        if k_ == 3:
            ins = sp[4]
            m = len(ins)
            w_end = s_ + f + m if m < 5000 else s_ + f
            refw = refw_of[t] if t in refw_of else be.fetch_seq(ref, sp[1], s_ - f, w_end)
            if not (type(refw) is str and refw.isascii() and type(ins) is str and ins.isascii()):
                continue
            n_x = ins.count("X")
            # both flanks of the alt allele lie inside the window just read (SF:1872): slices of it, unless a contig end cut it
            if len(refw) < 2 * f + 1 or (n_x and n_x != m):
                continue
            r_i, x_i = lit(refw), lit(ins)
            alt = der([(r_i, 0, f + 1, 0), (x_i, 0, m, 0), (r_i, f, f + 1, 0)])
            # window_size_refine(ref_seq + ins_seq) below 5 kb of insertion (which strips an all-X payload again), else (ref_seq)
            win = r_i if (n_x or m >= 5000) else der([(r_i, 0, len(refw), 0), (x_i, 0, m, 0)])
            twin = not (_window_traits(refw)[0] and _window_traits(ins)[0])
            u_ref = der([(r_i, 0, len(refw), 0)], True) if twin else r_i
            u_alt = der([(r_i, 0, f + 1, 0), (x_i, 0, m, 0), (r_i, f, f + 1, 0)], True) if twin else alt
            loc.append((j, t, 3, r_i, alt, u_ref, u_alt, win, None, len(refw), 2 * f + 2 + m))
            continue
        refw = refw_of[t] if t in refw_of else be.fetch_seq(ref, sp[1], s_ - f, e_ + f)
        lw = len(refw)
        if not (type(refw) is str and refw.isascii()) or lw != e_ - s_ + 2 * f + 1:       # (cut by a contig end: the drivers' way)
            continue
        r_i = lit(refw)
        w_upper, w_nocomp = _window_traits(refw)
        if k_ == 0:
            segs = [(r_i, 0, f, 0), (r_i, lw - f, f, 0)]                                   # ref_seq[:flank] + ref_seq[-flank:], SF:1712
            alt = der(segs)
            twin = not w_upper
            u_ref = der([(r_i, 0, lw, 0)], True) if twin else r_i
            u_alt = der(segs, True) if twin else alt
            loc.append((j, t, 0, r_i, alt, u_ref, u_alt, r_i, None, lw, 2 * f))
        elif k_ == 1:
            if lw <= 2 * f or w_nocomp:
                continue
            segs = [(r_i, 0, f, 0), (r_i, f, lw - 2 * f, L.SEG_REVCOMP), (r_i, lw - f, f, 0)]   # SF:1907
            alt = der(segs)
            twin = not w_upper
            u_ref = der([(r_i, 0, lw, 0)], True) if twin else r_i
            u_alt = der(segs, True) if twin else alt
            loc.append((j, t, 1, r_i, alt, u_ref, u_alt, r_i, alt, lw, lw))
        else:
            if lw <= 2 * f:
                continue
            segs = [(r_i, 0, f, 0), (r_i, f, lw - 2 * f, 0), (r_i, f, lw - 2 * f, 0), (r_i, lw - f, f, 0)]   # SF:1755
            alt = der(segs)
            mid = refw[f:lw - f]
            alt_texts[len(loc)] = refw[:f] + mid + mid + refw[lw - f:]
            loc.append((j, t, 2, r_i, alt, r_i, alt, r_i, alt, lw, 2 * lw - 2 * f))
    if not loc:
        return out
    _mark("alleles")
    # reads of the live loci behind the windows: slices of the records' own sequences
    n_lit_w = len(lit_addr)
    rd_first = []
    parts_a, parts_l, parts_m, parts_f = [], [], [], []
    kfl = kf.tolist()
    at = n_lit_w
    for (j, t, *_rest) in loc:
        a, b = kfl[j], kfl[j + 1]
        rd_first.append(at)
        if on_device:
            parts_a.append(addr[a:b])                    # (the record's packed bases; the read starts at base q0 of them)
            parts_f.append(q0[a:b])
        else:
            parts_a.append(addr[a:b] + q0[a:b].astype(np.uint64))
        parts_l.append((r_end[t] - r_start[t]) - miss[a:b])
        parts_m.append(miss[a:b])
        at += b - a
    all_addr = np.concatenate([np.asarray(lit_addr, dtype=np.uint64)] + parts_a)
    all_len = np.concatenate([np.asarray(lit_len, dtype=np.int64)] + parts_l)
    rd_miss = np.concatenate(parts_m)
    n_lit = len(all_addr)
    src_kind = src_first = None
    if on_device:
        src_kind = np.zeros(n_lit, dtype=np.uint8)
        src_kind[n_lit_w:] = 1
        src_first = np.concatenate([np.zeros(n_lit_w, dtype=np.int64)] + parts_f)
    segs_a = np.zeros(max(len(seg_rows), 1), dtype=L.SEG_DTYPE)
    if seg_rows:
        sr = np.asarray(seg_rows, dtype=np.int64)
        segs_a["parent"][:len(sr)], segs_a["off"][:len(sr)], segs_a["len"][:len(sr)], segs_a["flags"][:len(sr)] = sr[:, 0], sr[:, 1], sr[:, 2], sr[:, 3]

    def place(v):
        return n_lit - 1 - v if v < 0 else v
    if on_device:
        ss = engine.seqset_raw(all_addr, all_len, (np.asarray(seg_first, dtype=np.int32), segs_a, np.asarray(dflags, dtype=np.uint8)),
                               keepalive=(keep, keepalive), src_kind=src_kind, src_first=src_first)
        for bt in keepalive:                              # (the planes are made: the inflated blocks can go)
            bt.close()
    else:
        ss = engine.seqset_raw(all_addr, all_len, (np.asarray(seg_first, dtype=np.int32), segs_a, np.asarray(dflags, dtype=np.uint8)),
                               keepalive=(keep, keepalive))
    _mark("seqset")
    try:
        n_exc, n_inv, lens = ss.n_exc, ss.n_invalid, ss.lens
        # ---- window_size_refine's first self dot plot for every window (k = 10), in one plan -------------------------------
        wins: List[int] = []
        win_of = []                        # per live locus: (index of its ref-side window in `wins`, of its alt-side window or -1)
        for rec in loc:
            w1 = place(rec[7])
            a = len(wins)
            wins.append(w1)
            b = -1
            if rec[8] is not None and rec[2] == 1:              # inversion: its alt window too (a duplication's goes to refine_windows)
                b = len(wins)
                wins.append(place(rec[8]))
            win_of.append((a, b))
        wp = np.zeros(len(wins), dtype=L.PAIR_DTYPE)
        wp["seq1"] = wp["seq2"] = wins
        wp["k"] = 10
        plan = engine.plan(ss, wp)
        try:
            wst = plan.run().copy()
        finally:
            plan.close()
        _mark("window plan")
        nh, nd, nl = wst[:, L.ST_N_HITS], wst[:, L.ST_N_DIAG], wst[:, L.ST_N_LOWER]
        with np.errstate(divide="ignore", invalid="ignore"):
            frac = nl.astype(np.float64) / nh.astype(np.float64)
        w_arr = np.asarray(wins)
        # straight: accepted at the first size (no N rule in play, dots, not in the X-means band: qc = [diag, [0]] -> break)
        w_plain = (wst[:, L.ST_STATUS] == 0) & (n_exc[w_arr] <= 100) & (n_inv[w_arr] == 0)
        w_error = w_plain & (nh == 0)                                            # SF:2035, 2045: ['Error', 'Error']
        w_k10 = w_plain & (nh > 0) & ~((frac > 0.1) & (frac < 0.5))
        # duplications: the alt windows through refine_windows (their self plots meet the X-means branch, SF:1165)
        dup_k = {}
        dup_ids = [q for q, rec in enumerate(loc) if rec[2] == 2 and w_k10[win_of[q][0]]]
        if dup_ids:
            res = pipeline.refine_windows(engine, [alt_texts[q] for q in dup_ids])
            for q, r in zip(dup_ids, res):
                dup_k[q] = r
        # ---- which loci are scored, with which window size -----------------------------------------------------------------
        scored = []                        # (loc index, k)
        for q, rec in enumerate(loc):
            j, t, k_ = rec[0], rec[1], rec[2]
            a, b = win_of[q]
            if k_ in (0, 3):
                if w_error[a]:
                    out[t] = []                                                # k == 'Error': the driver returns no scores
                elif w_k10[a]:
                    scored.append((q, 10))
            elif k_ == 1:
                if w_k10[a] and w_k10[b]:
                    scored.append((q, 10))
            else:
                r = dup_k.get(q)
                if r is not None and not isinstance(r, BaseException) and r[0] != "Error":
                    scored.append((q, int(r[0])))
        # reads the library refuses (a character outside invert_base's alphabet, SF:1421) or that carry N / lower case (the
        # insertion driver's N rule, SF:1878): the drivers' route.  One reduction per column over all live loci.
        cnt = np.asarray([kfl[rec[0] + 1] - kfl[rec[0]] for rec in loc], dtype=np.int64)
        first = np.asarray(rd_first, dtype=np.int64)
        rd_inv = np.add.reduceat(n_inv[n_lit_w:n_lit].astype(np.int64), first - n_lit_w)
        rd_exc = np.add.reduceat(n_exc[n_lit_w:n_lit].astype(np.int64), first - n_lit_w)
        rd_max = np.maximum.reduceat(lens[n_lit_w:n_lit].astype(np.int64), first - n_lit_w)
        is_ins = np.asarray([rec[2] == 3 for rec in loc], dtype=bool)
        refuse = (rd_inv > 0) | (is_ins & (rd_exc > 0)) | (rd_max > L.MAX_SEQ_LEN) | (cnt <= 0)
        keep_sc = [(q, k) for q, k in scored if not refuse[q]]
        _mark("refine+select")
        if keep_sc:
            sc_lists = _score(engine, ss, loc, keep_sc, rd_first, kfl, rd_miss, n_lit_w, place)
            for (q, _k), v in zip(keep_sc, sc_lists):
                out[loc[q][1]] = v
        _mark("score")
        if _dbg:
            import sys as _sys
            print("fastpath chunk of %d: %s" % (n, "  ".join("%s %.1f" % (a, b * 1e3) for a, b in _marks)), file=_sys.stderr)
    finally:
        ss.close()
    return out


def _score(engine, ss, loc, scored, rd_first, kfl, rd_miss, n_lit_w, place):
    """One plan for every scored locus: the pair and read tables of pipeline.score_requests, from arrays."""
    import time as _time
    _ts = _time.perf_counter()
    i32 = np.int32
    # vapor_read.kind: deletion = abs_dis + within_10Perc, inversion / insertion abs_dis, duplication directed
    kinds = np.asarray([0, 1, 3, 1], dtype=np.int64)
    n_lit = ss.n_lit if hasattr(ss, "n_lit") else None
    qs = np.asarray([q for q, _k in scored], dtype=np.int64)
    rq_k = [k for _q, k in scored]
    cols = np.asarray([(loc[q][0], loc[q][2], place(loc[q][3]), place(loc[q][4]), place(loc[q][5]), place(loc[q][6]), loc[q][9], loc[q][10])
                       for q in qs.tolist()], dtype=np.int64).reshape(-1, 8)
    jj, kk, ri, ai, uri, uai, lref, lalt = (cols[:, c] for c in range(8))
    kf_a = np.asarray(kfl, dtype=np.int64)
    rq_n = (kf_a[jj + 1] - kf_a[jj]).tolist()
    rq_kind = kinds[kk].tolist()
    rq_lref, rq_lalt = lref.tolist(), lalt.tolist()
    rq_q0 = np.asarray(rd_first, dtype=np.int64)[qs]
    rq_m0 = (rq_q0 - n_lit_w).tolist()
    rq_q0 = rq_q0.tolist()
    # blocks of pairs: a deletion whose window has lower case is two blocks (upper-cased for abs_dis_m1b, as it is for
    # within_10Perc_m1b), everything else one
    two = (kk == 0) & (uri != ri)
    nblk = np.where(two, 2, 1)
    blk_a = np.concatenate(([0], np.cumsum(nblk)[:-1]))
    rq_blk_a = blk_a.tolist()
    rq_blk_b = (blk_a + two.astype(np.int64)).tolist()
    bl_rq = np.repeat(np.arange(len(qs)), nblk).tolist()
    plain = (kk == 0) | (kk == 2)
    first_ref = np.where(two, uri, np.where(plain, ri, uri))
    first_alt = np.where(two, uai, np.where(plain, ai, uai))
    first_fl = np.where(two, L.PF_C1, np.where(kk == 0, L.PF_C1 | L.PF_C2, np.where(kk == 2, L.PF_C1 | L.PF_DIR, L.PF_C1)))
    bl_ref = np.empty(int(nblk.sum()), dtype=np.int64); bl_alt = np.empty_like(bl_ref); bl_flags = np.empty_like(bl_ref)
    bl_ref[blk_a], bl_alt[blk_a], bl_flags[blk_a] = first_ref, first_alt, first_fl
    sec = blk_a[two] + 1
    bl_ref[sec], bl_alt[sec], bl_flags[sec] = ri[two], ai[two], L.PF_C2
    bl_ref, bl_alt, bl_flags = bl_ref.tolist(), bl_alt.tolist(), bl_flags.tolist()
    rq_n_a = np.asarray(rq_n, dtype=np.int64)
    n_reads_tot = int(rq_n_a.sum())
    rq_first = np.concatenate(([0], np.cumsum(rq_n_a)[:-1]))
    rq_q0_a = np.asarray(rq_q0, dtype=np.int64)
    bl_rq_a = np.asarray(bl_rq, dtype=np.int64)
    bl_n = rq_n_a[bl_rq_a]
    bl_base = 2 * np.concatenate(([0], np.cumsum(bl_n)[:-1]))
    br_blk = np.repeat(np.arange(len(bl_rq_a)), bl_n)
    br_i = np.arange(int(bl_n.sum())) - np.repeat(bl_base // 2, bl_n)
    br_rq = bl_rq_a[br_blk]
    pairs = np.zeros(2 * len(br_blk), dtype=L.PAIR_DTYPE)
    pairs["seq1"] = np.repeat((rq_q0_a[br_rq] + br_i).astype(i32), 2)
    al = np.empty(2 * len(br_blk), dtype=i32)
    al[0::2] = np.asarray(bl_ref, dtype=i32)[br_blk]
    al[1::2] = np.asarray(bl_alt, dtype=i32)[br_blk]
    pairs["seq2"] = al
    miss_of = rd_miss[np.asarray(rq_m0, dtype=np.int64)[br_rq] + br_i].astype(i32)
    pairs["off2"] = np.repeat(miss_of, 2)
    pairs["k"] = np.repeat(np.asarray(rq_k, dtype=i32)[br_rq], 2)
    pairs["flags"] = np.repeat(np.asarray(bl_flags, dtype=np.uint32)[br_blk], 2)
    rd_rq = np.repeat(np.arange(len(rq_n_a)), rq_n_a)
    rd_i = np.arange(n_reads_tot) - rq_first[rd_rq]
    pa = (bl_base[np.asarray(rq_blk_a, dtype=np.int64)][rd_rq] + 2 * rd_i).astype(i32)
    pb = (bl_base[np.asarray(rq_blk_b, dtype=np.int64)][rd_rq] + 2 * rd_i).astype(i32)
    table = np.zeros(n_reads_tot, dtype=L.READ_DTYPE)
    table["ref_a"], table["alt_a"], table["ref_b"], table["alt_b"] = pa, pa + 1, pb, pb + 1
    table["kind"] = np.asarray(rq_kind, dtype=i32)[rd_rq]
    table["locus"] = rd_rq.astype(i32)
    table["len_ref"] = np.asarray(rq_lref, dtype=i32)[rd_rq]
    table["len_alt"] = np.asarray(rq_lalt, dtype=i32)[rd_rq]
    import os as _os
    import time as _time
    _dbg = _os.environ.get("VAPOR_DEBUG_FASTPATH")
    _t0 = _time.perf_counter()
    plan = engine.plan(ss, pairs)
    try:
        _t1 = _time.perf_counter()
        plan.set_reads(table, len(scored))
        _t2 = _time.perf_counter()
        plan.run_loci(want_host=False, want_scores=True)
        _t3 = _time.perf_counter()
        sc = plan.read_scores[:n_reads_tot].tolist()
    finally:
        plan.close()
    out = []
    for x in range(len(scored)):
        a = int(rq_first[x])
        # (a read whose scorer output holds a 0 is skipped, e.g. SF:1913: NaN here)
        out.append([v for v in sc[a:a + rq_n[x]] if v == v])
    if _dbg:
        import sys as _sys
        print("  _score of %d loci, %d pairs: tables %.1f  plan %.1f  set_reads %.1f  run_loci %.1f  lists %.1f ms" % (
            len(scored), len(pairs), (_t0 - _ts) * 1e3, (_t1 - _t0) * 1e3, (_t2 - _t1) * 1e3, (_t3 - _t2) * 1e3, (_time.perf_counter() - _t3) * 1e3), file=_sys.stderr)
    return out
