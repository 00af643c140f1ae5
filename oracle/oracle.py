"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY - see vapor_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It restates, function by function, the reference's scoring path; integer work
(k-mer hash join, gap clustering, counts) runs in vapor_oracle.c, the float64 finishing
steps are restated here with the reference's own operation order.

Parity status: PINNED against reference-generated vectors (tests/golden, produced by
oracle/gen_golden.py; checked by tests/test_oracle_golden.py).
SF = /root/reference/vapor_vali/Simple_function.pyx.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvapor_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vapor_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra",
                               "-o", _SO, src])
    return _SO


_TWIN = os.path.join(_HERE, "_build", "libvapor_cpu.so")


def build_twin(force: bool = False) -> str:
    """libvapor_cpu.so: the C ABI of include/vapor_hip.h on this oracle (cpu_twin.cpp; test infrastructure)."""
    # (vapor_bam.cpp: the BGZF/BAM host helper of the product library, plain C++ without device code, is part of the ABI)
    bam = os.path.join(os.path.dirname(_HERE), "vapor_amd", "csrc", "vapor_bam.cpp")
    srcs = [os.path.join(_HERE, "cpu_twin.cpp"), os.path.join(_HERE, "vapor_oracle.c"),
            os.path.join(os.path.dirname(_HERE), "include", "vapor_hip.h"), bam]
    if force or not os.path.exists(_TWIN) or any(os.path.getmtime(_TWIN) < os.path.getmtime(x) for x in srcs):
        os.makedirs(os.path.dirname(_TWIN), exist_ok=True)
        obj = os.path.join(_HERE, "_build", "vapor_oracle_twin.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-c", "-o", obj, srcs[1]])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread",
                               "-I" + os.path.join(os.path.dirname(_HERE), "include"), "-o", _TWIN, srcs[0], bam, obj, "-lz"])
    return _TWIN


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.vo_dotdata.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                 ctypes.c_int, i32p, ctypes.c_int64, i64p]
        L.vo_dotdata.restype = ctypes.c_int
        L.vo_clean_c1.argtypes = [i32p, ctypes.c_int64, u8p]
        L.vo_clean_c1.restype = ctypes.c_int
        L.vo_clean_c2.argtypes = [i32p, ctypes.c_int64, u8p]
        L.vo_clean_c2.restype = ctypes.c_int
        L.vo_pair_stats.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                    ctypes.c_int, i64p, i32p, ctypes.c_int64, u8p, u8p]
        L.vo_pair_stats.restype = ctypes.c_int
        _lib = L
    return _lib


def _enc(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("latin-1")


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def dotdata_array(k: int, seq1, seq2) -> np.ndarray:
    """dotdata() as an (n,2) int32 array of (j, i) rows in the reference's list order.
    Raises KeyError where the reference does (SF:1421)."""
    L = lib()
    b1, b2 = _enc(seq1), _enc(seq2)
    n = ctypes.c_int64(0)
    rc = L.vo_dotdata(k, b1, len(b1), b2, len(b2), None, 0, ctypes.byref(n))
    if rc == -3:
        raise KeyError("invert_base")
    assert rc == 0, rc
    out = np.empty((n.value, 2), dtype=np.int32)
    if n.value:
        rc = L.vo_dotdata(k, b1, len(b1), b2, len(b2), _p(out, ctypes.c_int32), n.value, ctypes.byref(n))
        assert rc == 0, rc
    return out


def dotdata(kmerlen: int, seq1, seq2):
    """SF:545-549: list of (pos_in_seq2, pos_in_seq1) tuples."""
    return [(int(a), int(b)) for a, b in dotdata_array(kmerlen, seq1, seq2)]


def clean_c1_flags(hits: np.ndarray) -> np.ndarray:
    hits = np.ascontiguousarray(hits, dtype=np.int32).reshape(-1, 2)
    keep = np.zeros(len(hits), dtype=np.uint8)
    if len(hits):
        assert lib().vo_clean_c1(_p(hits, ctypes.c_int32), len(hits), _p(keep, ctypes.c_uint8)) == 0
    return keep


def clean_c2_flags(hits: np.ndarray) -> np.ndarray:
    hits = np.ascontiguousarray(hits, dtype=np.int32).reshape(-1, 2)
    keep = np.zeros(len(hits), dtype=np.uint8)
    if len(hits):
        assert lib().vo_clean_c2(_p(hits, ctypes.c_int32), len(hits), _p(keep, ctypes.c_uint8)) == 0
    return keep


def clean_dotdata_diagnal_and_anti_diagnal(ref_dotdata):
    """SF:432-448 (empty input gives the reference's odd [[],[]])."""
    if len(ref_dotdata) == 0:
        return [[], []]
    h = np.asarray(ref_dotdata, dtype=np.int32).reshape(-1, 2)
    k = clean_c1_flags(h)
    return [ref_dotdata[t] for t in range(len(ref_dotdata)) if k[t]]


def pair_stats(k: int, seq1, seq2, want_hits: bool = False):
    """int64[16] statistics record of one (seq1, seq2) dot plot (see vapor_oracle.c),
    optionally with the hits and the C1/C2 keep flags."""
    L = lib()
    b1, b2 = _enc(seq1), _enc(seq2)
    n = ctypes.c_int64(0)
    rc = L.vo_dotdata(k, b1, len(b1), b2, len(b2), None, 0, ctypes.byref(n))
    if rc == -3:
        raise KeyError("invert_base")
    cap = max(int(n.value), 1)
    hits = np.empty((cap, 2), dtype=np.int32)
    k1 = np.zeros(cap, dtype=np.uint8)
    k2 = np.zeros(cap, dtype=np.uint8)
    st = np.zeros(16, dtype=np.int64)
    rc = L.vo_pair_stats(k, b1, len(b1), b2, len(b2), _p(st, ctypes.c_int64), _p(hits, ctypes.c_int32),
                         cap, _p(k1, ctypes.c_uint8), _p(k2, ctypes.c_uint8))
    assert rc == 0, rc
    if want_hits:
        m = int(st[0])
        return st, hits[:m], k1[:m], k2[:m]
    return st


# ---------------------------------------------------------------------------
# float64 finishing steps, restated with the reference's operation order
# ---------------------------------------------------------------------------

def eu_dis_abs_calcu(dots):
    """SF:705-708."""
    return np.mean([abs(int(j) - int(i)) for j, i in dots])


def eu_dis_dots_within_10perc(dots):
    """SF:730-733."""
    r = [abs(float(j - i) / float(j)) for j, i in dots if j > 0]
    return len([x for x in r if x < 0.16])


def eu_dis_single_dot(dot):
    """SF:710-716."""
    if dot[0] == 0:
        return abs(float(dot[0] - dot[1]) / float(dot[0] + 1))
    return abs(float(dot[0] - dot[1]) / float(dot[0]))


def eu_dis_dir_calcu(dots):
    """SF:718-722."""
    v = [d[0] - d[1] for d in dots if eu_dis_single_dot(d) > 0.1]
    if v == []:
        return 0.0001
    return np.mean(v)


def number_cluster(values, edges):
    """SF:1104-1118: values (sorted in place) go to bin b-1 for the first edge b>=1 they are
    below; whatever is left when the edges run out goes to the last bin."""
    bins = [[] for _ in edges]
    a, b = 0, 1
    values.sort()
    while True:
        if a == len(values) or b == len(edges):
            break
        if values[a] < edges[b]:
            bins[b - 1].append(values[a])
            a += 1
        else:
            b += 1
    if a < len(values):
        bins[-1] += values[a:]
    return bins


def find_longest_list(list_set):
    """SF:788-792 with unify_list SF:1483-1488."""
    m = max(len(x) for x in list_set)
    out = []
    for x in list_set:
        if len(x) == m and x not in out:
            out.append(x)
    return out


def dis_to_diagnal_most_abundant_defined(dots):
    """SF:582-591."""
    d = [int(x[1]) - int(x[0]) for x in dots]
    lo, hi = min(d), max(d)
    edges = [lo + t * float(hi - lo) / 10.0 for t in range(11)]
    kept1 = find_longest_list(number_cluster(d, edges))
    kept2 = []
    for km in kept1:
        e2 = [min(km) + t * float(max(km) - min(km)) / 10.0 for t in range(11)]
        kept2 += find_longest_list(number_cluster(km, e2))
    if len(kept2) == 1:
        return np.median(kept2[0])
    return 0


def _as_list(h):
    """(j, i) rows as Python ints (lists of two: the scorers only index them)."""
    return np.asarray(h).tolist()


def score_abs_dis_m1b(ref_seq, alt_seq, x, window_size, plots=None):
    """calcu_vapor_single_read_score_abs_dis_m1b, SF:182-203.  `plots` = (R, A): the two dotdata lists when the caller has
    filled them already (tests that score thousands of reads fill each plot once for all three scorers; only valid when
    the windows are upper-case already, which the function checks)."""
    ref_seq = ref_seq.upper()
    alt_seq = alt_seq.upper()
    if plots is None:
        R = dotdata_array(window_size, x[0], ref_seq[x[1]:])
        A = dotdata_array(window_size, x[0], alt_seq[x[1]:])
    else:
        R, A = plots
    if len(R) > 2 and len(A) > 2:
        if float(len(R)) / min([float(len(ref_seq)), float(len(alt_seq))]) > 0.1:
            r_ok = float(R[-1][0] - R[0][0]) / float(len(ref_seq)) > 0.6
            a_ok = float(A[-1][0] - A[0][0]) / float(len(alt_seq)) > 0.6
            if r_ok and a_ok:
                Rk = R[clean_c1_flags(R) > 0]
                Ak = A[clean_c1_flags(A) > 0]
                if len(Rk) > 0 and len(Ak) > 0:
                    return [eu_dis_abs_calcu(_as_list(Rk)), eu_dis_abs_calcu(_as_list(Ak))]
                return [0, 0]
            if r_ok:
                return [1.1, 2.1]
            if a_ok:
                return [2.1, 1.1]
            return [0, 0]
        return [0, 0]
    return [0, 0]


def score_within_10Perc_m1b(ref_seq, alt_seq, x, window_size, plots=None):
    """calcu_vapor_single_read_score_within_10Perc_m1b, SF:277-294 (returns alt first)."""
    if plots is None:
        R = dotdata_array(window_size, x[0], ref_seq[x[1]:])
        A = dotdata_array(window_size, x[0], alt_seq[x[1]:])
    else:
        R, A = plots
    if max([float(len(R)) / float(len(ref_seq)), float(len(A)) / float(len(alt_seq))]) > 0.1:
        Rk = R[clean_c2_flags(R) > 0]
        Ak = A[clean_c2_flags(A) > 0]
        if len(Rk) > 0 and len(Ak) > 0:
            return [eu_dis_dots_within_10perc(_as_list(Ak)), eu_dis_dots_within_10perc(_as_list(Rk))]
        return [0, 0]
    return [0, 0]


def score_directed_dis_m1b_redefine_diagnal(ref_seq, alt_seq, x, window_size, plots=None):
    """calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal, SF:241-257."""
    if plots is None:
        R = dotdata_array(window_size, x[0], ref_seq[x[1]:])
        A = dotdata_array(window_size, x[0], alt_seq[x[1]:])
    else:
        R, A = plots
    if (float(len(R)) / float(len(ref_seq)) > 0.1 and float(len(A)) / float(len(alt_seq)) > 0.1
            and float(R[-1][0] - R[0][0]) / float(len(ref_seq)) > 0.7
            and float(A[-1][0] - A[0][0]) / float(len(alt_seq)) > 0.7):
        Rk = _as_list(R[clean_c1_flags(R) > 0])
        Ak = _as_list(A[clean_c1_flags(A) > 0])
        if len(Rk) > 0 and len(Ak) > 0:
            cr = dis_to_diagnal_most_abundant_defined(Rk)
            ca = dis_to_diagnal_most_abundant_defined(Ak)
            return [abs(eu_dis_dir_calcu([[d[0] + cr, d[1]] for d in Rk])),
                    abs(eu_dis_dir_calcu([[d[0] + ca, d[1]] for d in Ak]))]
        return [0, 0]
    return [0, 0]


def qual_check_counts(hits: np.ndarray):
    """Integer part of qual_check_repetitive_region, SF:1154-1171: (n, n_diag, n_lower)."""
    j, i = hits[:, 0], hits[:, 1]
    return int(len(hits)), int(np.sum(j == i)), int(np.sum(j > i))


def result_organize_ins(info_list):
    """SF:1219-1231."""
    if len(info_list[1]) > 0:
        pos = [s for s in info_list[1] if float(s) > 0]
        neg = [s for s in info_list[1] if not float(s) > 0]
        gs = float(len(pos)) / float(len(pos) + len(neg))
        qs = np.mean(pos) if pos else 0
        return [info_list[0]] + [qs, gs, ",".join([str(round(float(s), 2)) for s in info_list[1]])]
    return [info_list[0]] + ["NA" for _ in range(3)]


def log_likelihood_calcu(k, l, m, g, err=0.05):
    """SF:2071-2077."""
    out = -k * np.log(m)
    for _ in range(l):
        out += np.log((m - g) * err + g * (1 - err))
    for _ in range(k - l):
        out += np.log((m - g) * (1 - err) + g * err)
    return out


def gt_estimate_log_likelihood(vapor_result):
    """SF:2054-2069."""
    scores = [float(s) for s in vapor_result[-1].split(",")]
    k = len(scores)
    l = len([s for s in scores if not s > 0])
    ll = [log_likelihood_calcu(k, l, 2, 2), log_likelihood_calcu(k, l, 2, 1), log_likelihood_calcu(k, l, 2, 0)]
    ori = [np.exp(v - max(ll)) for v in ll]
    norm = [v / sum(ori) for v in ori]
    gq = -np.log(np.median(norm)) / np.log(10)
    gt = ["0/0", "0/1", "1/1"][ll.index(max(ll))]
    if gt == "0/0" and vapor_result[-2] > .15:
        gt = "0/1"
    return [gt, gq]
