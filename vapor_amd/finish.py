"""Float64 finishing steps on the host: from the device's integer statistics to per-read
scores, VaPoR_QS / VaPoR_GS / VaPoR_Rec and VaPoR_GT / VaPoR_GQ.

Each function states the reference routine it reproduces (SF = vapor_vali/Simple_function.pyx
in the reference tree).  Gates and ratios are evaluated with the same float64 operations the
reference applies to the same integers, so results are identical, not merely close.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import _lib as L


# ---------------------------------------------------------------------------
# per-read scorers from a (ref, alt) pair of statistics records
# ---------------------------------------------------------------------------

def score_abs_dis_m1b(st_ref: np.ndarray, st_alt: np.ndarray, len_ref: int, len_alt: int):
    """calcu_vapor_single_read_score_abs_dis_m1b, SF:182-203.  `st_*` are the records of
    dotdata(k, read, upper(allele)[miss:]) with VAPOR_PF_C1; len_* the full allele lengths."""
    nr, na = int(st_ref[L.ST_N_HITS]), int(st_alt[L.ST_N_HITS])
    if nr > 2 and na > 2:
        if float(nr) / min([float(len_ref), float(len_alt)]) > 0.1:
            r_ok = float(st_ref[L.ST_LAST_J] - st_ref[L.ST_FIRST_J]) / float(len_ref) > 0.6
            a_ok = float(st_alt[L.ST_LAST_J] - st_alt[L.ST_FIRST_J]) / float(len_alt) > 0.6
            if r_ok and a_ok:
                kr, ka = int(st_ref[L.ST_C1_KEPT]), int(st_alt[L.ST_C1_KEPT])
                if kr > 0 and ka > 0:
                    return [np.float64(int(st_ref[L.ST_C1_SUM_ABS])) / kr, np.float64(int(st_alt[L.ST_C1_SUM_ABS])) / ka]
                return [0, 0]
            if r_ok:
                return [1.1, 2.1]
            if a_ok:
                return [2.1, 1.1]
            return [0, 0]
        return [0, 0]
    return [0, 0]


def score_within_10Perc_m1b(st_ref, st_alt, len_ref: int, len_alt: int):
    """calcu_vapor_single_read_score_within_10Perc_m1b, SF:277-294 (alt count first).
    Records of dotdata(k, read, allele[miss:]) with VAPOR_PF_C2."""
    nr, na = int(st_ref[L.ST_N_HITS]), int(st_alt[L.ST_N_HITS])
    if max([float(nr) / float(len_ref), float(na) / float(len_alt)]) > 0.1:
        if int(st_ref[L.ST_C2_KEPT]) > 0 and int(st_alt[L.ST_C2_KEPT]) > 0:
            return [int(st_alt[L.ST_C2_COUNT10]), int(st_ref[L.ST_C2_COUNT10])]
        return [0, 0]
    return [0, 0]


def _dir_value(st):
    n = int(st[L.ST_DIR_N])
    if n == 0:
        return 0.0001
    return np.float64(int(st[L.ST_DIR_SUM2]) * 0.5) / n


def score_directed_dis_m1b_redefine_diagnal(st_ref, st_alt, len_ref: int, len_alt: int):
    """calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal, SF:241-257.
    Records with VAPOR_PF_C1 | VAPOR_PF_DIR."""
    nr, na = int(st_ref[L.ST_N_HITS]), int(st_alt[L.ST_N_HITS])
    if (float(nr) / float(len_ref) > 0.1 and float(na) / float(len_alt) > 0.1
            and float(st_ref[L.ST_LAST_J] - st_ref[L.ST_FIRST_J]) / float(len_ref) > 0.7
            and float(st_alt[L.ST_LAST_J] - st_alt[L.ST_FIRST_J]) / float(len_alt) > 0.7):
        if int(st_ref[L.ST_C1_KEPT]) > 0 and int(st_alt[L.ST_C1_KEPT]) > 0:
            return [abs(_dir_value(st_ref)), abs(_dir_value(st_alt))]
        return [0, 0]
    return [0, 0]


# ---------------------------------------------------------------------------
# the same three scorers over arrays of reads (batch pipeline, bench)
# ---------------------------------------------------------------------------

def batch_scores(kind: np.ndarray, st_ref: np.ndarray, st_alt: np.ndarray, len_ref: np.ndarray,
                 len_alt: np.ndarray):
    """Vectorised scorers.  kind[r] in {1: abs_dis_m1b, 2: within_10Perc_m1b, 3: directed_dis}.
    Returns (a, b, valid) with `valid` the drivers' `not 0 in [a, b]` test (SF:1718 etc.) and
    NaN excluded where the reference excludes it."""
    n = len(kind)
    a = np.zeros(n, dtype=np.float64)
    b = np.zeros(n, dtype=np.float64)
    lr = len_ref.astype(np.float64)
    la = len_alt.astype(np.float64)
    nr = st_ref[:, L.ST_N_HITS].astype(np.float64)
    na = st_alt[:, L.ST_N_HITS].astype(np.float64)
    span_r = (st_ref[:, L.ST_LAST_J] - st_ref[:, L.ST_FIRST_J]).astype(np.float64) / lr
    span_a = (st_alt[:, L.ST_LAST_J] - st_alt[:, L.ST_FIRST_J]).astype(np.float64) / la
    with np.errstate(divide="ignore", invalid="ignore"):
        # S1
        m1 = kind == 1
        g = m1 & (st_ref[:, L.ST_N_HITS] > 2) & (st_alt[:, L.ST_N_HITS] > 2) & (nr / np.minimum(lr, la) > 0.1)
        r_ok, a_ok = span_r > 0.6, span_a > 0.6
        both = g & r_ok & a_ok
        kept = both & (st_ref[:, L.ST_C1_KEPT] > 0) & (st_alt[:, L.ST_C1_KEPT] > 0)
        a[kept] = st_ref[kept, L.ST_C1_SUM_ABS].astype(np.float64) / st_ref[kept, L.ST_C1_KEPT]
        b[kept] = st_alt[kept, L.ST_C1_SUM_ABS].astype(np.float64) / st_alt[kept, L.ST_C1_KEPT]
        only_r = g & r_ok & ~a_ok
        a[only_r], b[only_r] = 1.1, 2.1
        only_a = g & ~r_ok & a_ok
        a[only_a], b[only_a] = 2.1, 1.1
        # S2
        m2 = kind == 2
        g2 = m2 & (np.maximum(nr / lr, na / la) > 0.1) & (st_ref[:, L.ST_C2_KEPT] > 0) & (st_alt[:, L.ST_C2_KEPT] > 0)
        a[g2] = st_alt[g2, L.ST_C2_COUNT10]
        b[g2] = st_ref[g2, L.ST_C2_COUNT10]
        # S3
        m3 = kind == 3
        g3 = (m3 & (nr / lr > 0.1) & (na / la > 0.1) & (span_r > 0.7) & (span_a > 0.7)
              & (st_ref[:, L.ST_C1_KEPT] > 0) & (st_alt[:, L.ST_C1_KEPT] > 0))
        for st, out in ((st_ref, a), (st_alt, b)):
            nn = st[:, L.ST_DIR_N]
            v = np.where(nn > 0, (st[:, L.ST_DIR_SUM2].astype(np.float64) * 0.5) / np.maximum(nn, 1), 0.0001)
            out[g3] = np.abs(v[g3])
    valid = (a != 0) & (b != 0)
    return a, b, valid


def read_scores(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """1 - b/a (SF:1719 etc.)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return 1.0 - b / a


# ---------------------------------------------------------------------------
# per-locus results
# ---------------------------------------------------------------------------

def result_organize_ins(info_list):
    """SF:1219-1231: [key, QS, GS, Rec] or [key, 'NA', 'NA', 'NA']."""
    scores = info_list[1]
    if len(scores) > 0:
        pos = [s for s in scores if float(s) > 0]
        gs = float(len(pos)) / float(len(scores))
        qs = np.mean(pos) if pos else 0
        return [info_list[0], qs, gs, ",".join([str(round(float(s), 2)) for s in scores])]
    return [info_list[0], "NA", "NA", "NA"]


def _mean_like_numpy(a):
    """np.mean of a list of fewer than 128 floats, bit for bit, without the array round trip (7 us a locus): numpy's
    add.reduce sums fewer than 8 values one by one and otherwise keeps eight strided partial sums over the first n - n % 8
    values, combines them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and adds the rest one by one (its block size is 128; the
    finish kernel sums the same way).  tests/test_host_cpu.py compares the two on random lists."""
    n = len(a)
    if n >= 128:
        return np.mean(a)
    if n < 8:
        r = 0.0
        for x in a:
            r += x
        return np.float64(r) / n
    r0, r1, r2, r3, r4, r5, r6, r7 = a[:8]
    m = n - (n % 8)
    i = 8
    while i < m:
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7]
        i += 8
    res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
    while i < n:
        res += a[i]
        i += 1
    return np.float64(res) / n


def row_tail(scores):
    """[VaPoR_QS, VaPoR_GS, VaPoR_GT, VaPoR_GQ, VaPoR_Rec] of one locus - what result_organize_ins (SF:1219-1231) followed by
    gt_estimate_log_likelihood (SF:2054-2069) inside write_output_main (SF:2084-2088) put into its row -, or five 'NA': the
    same values through one rounding of every score instead of round -> str -> split -> float (the strings parse back to the
    rounded values exactly)."""
    n = len(scores)
    if n == 0:
        return ["NA", "NA", "NA", "NA", "NA"]
    fl = [float(s) for s in scores]
    pos = [s for s in fl if s > 0]
    gs = float(len(pos)) / float(n)
    qs = _mean_like_numpy(pos) if pos else 0
    rd = [round(s, 2) for s in fl]
    l = 0
    for s in rd:
        if not s > 0:
            l += 1
    idx, gq = _gt_from_counts(n, l)
    gt = _GT_NAMES[idx]
    if gt == "0/0" and gs > .15:
        gt = "0/1"
    return [qs, gs, gt, gq, ",".join([str(s) for s in rd])]


def row_tails(scores_list) -> list:
    """row_tail for every locus of a table through one call of the library's host helper (vapor_row_tails: the rounding, the
    Rec strings, the positive scores' mean and the two counts for all loci; what stays here is a dict lookup and two str() per
    locus).  A locus the helper hands back (a score that is not finite, or huge) and a table that is not made of lists of
    numbers go through row_tail."""
    n = len(scores_list)
    if n == 0:
        return []
    import ctypes
    from itertools import chain
    try:
        lens = np.fromiter(map(len, scores_list), dtype=np.int64, count=n)
        total = int(lens.sum())
        flat = np.fromiter(chain.from_iterable(scores_list), dtype=np.float64, count=total)
    except (TypeError, ValueError):
        return [row_tail(sc) for sc in scores_list]
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    qs, n_pos, n_nonpos = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
    text = ctypes.create_string_buffer(16 * total + 16)
    text_off = np.empty(n + 1, dtype=np.int64)
    lib = L.load()
    rc = lib.vapor_row_tails(n, off.ctypes.data, flat.ctypes.data, qs.ctypes.data, n_pos.ctypes.data, n_nonpos.ctypes.data,
                             ctypes.addressof(text), len(text), text_off.ctypes.data)
    if rc == L.E_OVERFLOW:                        # (scores of 1e13 and more are not written, so this does not happen)
        text = ctypes.create_string_buffer(int(text_off[n]) + 16)
        rc = lib.vapor_row_tails(n, off.ctypes.data, flat.ctypes.data, qs.ctypes.data, n_pos.ctypes.data, n_nonpos.ctypes.data,
                                 ctypes.addressof(text), len(text), text_off.ctypes.data)
    if rc != 0:
        raise RuntimeError("vapor_row_tails: %s" % lib.vapor_bam_last_error().decode())
    recs = text.raw[:int(text_off[n])].decode("ascii")
    to = text_off.tolist()
    out = []
    for t, (k, np_, l) in enumerate(zip(lens.tolist(), n_pos.tolist(), n_nonpos.tolist())):
        if k == 0:
            out.append(["NA", "NA", "NA", "NA", "NA"])
            continue
        if l < 0:
            out.append(row_tail(scores_list[t]))
            continue
        gs = float(np_) / float(k)
        idx, gq = _gt_from_counts(k, l)
        gt = _GT_NAMES[idx]
        if gt == "0/0" and gs > .15:
            gt = "0/1"
        out.append([qs[t] if np_ else 0, gs, gt, gq, recs[to[t]:to[t + 1]]])
    return out


_GT_NAMES = ("0/0", "0/1", "1/1")
_GT_CACHE = {}


def _gt_from_counts(k: int, l: int):
    """log_likelihood_calcu (SF:2071-2077) and the arg-max / quality of SF:2058-2066 for k
    reads of which l are non-positive; depends on (k, l) only."""
    got = _GT_CACHE.get((k, l))
    if got is None:
        ll = []
        for g in (2, 1, 0):
            out = -k * np.log(2)
            for _ in range(l):
                out += np.log((2 - g) * 0.05 + g * (1 - 0.05))
            for _ in range(k - l):
                out += np.log((2 - g) * (1 - 0.05) + g * 0.05)
            ll.append(out)
        ori = [np.exp(v - max(ll)) for v in ll]
        norm = [v / sum(ori) for v in ori]
        gq = -np.log(np.median(norm)) / np.log(10)
        got = (ll.index(max(ll)), gq)
        _GT_CACHE[(k, l)] = got
    return got


def gt_estimate_log_likelihood(vapor_result):
    """SF:2054-2069: works on the rounded Rec string and on GS (vapor_result[-2])."""
    scores = [float(s) for s in vapor_result[-1].split(",")]
    k = len(scores)
    l = len([s for s in scores if not s > 0])
    idx, gq = _gt_from_counts(k, l)
    gt = _GT_NAMES[idx]
    if gt == "0/0" and vapor_result[-2] > .15:
        gt = "0/1"
    return [gt, gq]


def rounded_nonpositive(scores: np.ndarray) -> np.ndarray:
    """`not float(str(round(s, 2))) > 0` without the string round trip: round(s, 2) is
    positive exactly when s >= 0.005 (the double nearest 0.005 lies above 5e-3, its
    predecessor below), NaN counts as non-positive."""
    return ~(scores >= 0.005)


def locus_summary(scores: Sequence[float]):
    """(QS, GS, GT index, GQ) for one locus from its read scores (L2 + L3), or None when the
    list is empty ('NA' row)."""
    s = np.asarray(scores, dtype=np.float64)
    if s.size == 0:
        return None
    pos = s[s > 0]
    gs = float(pos.size) / float(s.size)
    qs = np.mean(pos) if pos.size else 0
    idx, gq = _gt_from_counts(int(s.size), int(rounded_nonpositive(s).sum()))
    if idx == 0 and gs > .15:
        idx = 1
    return qs, gs, idx, gq


def gt_name(idx: int) -> str:
    return _GT_NAMES[idx]


_GT_TABLE = None


def gt_table() -> np.ndarray:
    """(65, 65, 2) float64: genotype index and quality for k scored reads of which l round to <= 0,
    for the device-side finish (include/vapor_hip.h, vapor_plan_set_reads)."""
    global _GT_TABLE
    if _GT_TABLE is None:
        _GT_TABLE = _gt_table_rows(L.GT_TABLE_N)
    return _GT_TABLE


def _gt_table_rows(n: int) -> np.ndarray:
    """_gt_from_counts for every (k, l), k < n, with the same float64 operations in the same order, a row of l at a time
    (0.4 s of scalar loops at every process start otherwise): the log-likelihood of genotype g is -k ln 2, then l times
    + ln((2-g) 0.05 + g 0.95), then (k - l) times + ln((2-g) 0.95 + g 0.05), added one by one (SF:2071-2077) - a running sum
    over l for the first part, k steps over the whole row for the second; arg-max, normalisation, median of three and the
    quality as SF:2058-2066.  tests/test_finish_cpu.py compares the table with the scalar statement entry for entry."""
    t = np.zeros((n, n, 2), dtype=np.float64)
    ln2 = np.log(2)
    for k in range(1, n):
        ll = []
        for g in (2, 1, 0):
            a = np.log((2 - g) * 0.05 + g * (1 - 0.05))
            b = np.log((2 - g) * (1 - 0.05) + g * 0.05)
            cur = np.cumsum(np.concatenate(([-k * ln2], np.full(k, a))))       # cur[l] = (..((-k ln 2 + a) + a) ..) l times
            for step in range(1, k + 1):
                cur[:k - step + 1] += b                                         # the entries with k - l >= step
            ll.append(cur)
        ll = np.stack(ll)                                                       # (3, k + 1): rows g = 2, 1, 0
        top = ll.max(axis=0)
        ori = np.exp(ll - top)
        norm = ori / ((ori[0] + ori[1]) + ori[2])
        with np.errstate(divide="ignore"):
            gq = -np.log(np.sort(norm, axis=0)[1]) / np.log(10)
        t[k, :k + 1, 0] = np.argmax(ll, axis=0)                                 # (the first of equal maxima, as list.index)
        t[k, :k + 1, 1] = gq
    return t
