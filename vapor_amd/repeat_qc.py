"""Repeat quality check behind window_size_refine (SURVEY.md component #9, §8a-Q).

`qual_check_repetitive_region` (SF:1154-1171) looks at a self dot plot: the fraction of dots on
the main diagonal, and - only when the lower-triangle dots are between 10 % and 50 % of all dots
- the sizes of the repeat blocks those dots form, found by a recursive k-means / BIC split
("X-means", SF:2101-2116, 856-887, 480-517).  The integer part (dot, diagonal and lower-triangle
counts) comes from the device; the clustering is third-party float arithmetic that the reference
runs unseeded (sklearn KMeans k-means++ and scipy.cluster.vq.kmeans without a seed), so its
output is not pinned by any vector (parity unpinned for this sub-branch).

Deviation kept on purpose: SF:878 evaluates `scipy.std`, an attribute SciPy removed; the value
is never used, so this module simply does not evaluate it (on a current SciPy the reference
raises AttributeError whenever BIC prefers more than one cluster).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np


_TPC = None


def _one_thread():
    """Context limiting OpenMP to one thread; the library scan behind it is done once (it takes longer than a fit)."""
    global _TPC
    try:
        if _TPC is None:
            from threadpoolctl import ThreadpoolController
            _TPC = ThreadpoolController()
        return _TPC.limit(limits=1, user_api="openmp")
    except Exception:                      # threadpoolctl missing or too old: run as is
        import contextlib
        return contextlib.nullcontext()


def _log10(x):
    """calcu_log10, SF:155-159."""
    if x == 0:
        return 0
    with np.errstate(divide="ignore"):          # (a zero variance term gives -inf, as in the reference; not worth a warning per process)
        return np.log10(x)


def _bic(km, X) -> float:
    """compute_bic, SF:480-517 (clusters whose variance term is negative are left out)."""
    from scipy.spatial import distance
    centers = km.cluster_centers_
    labels = km.labels_
    m = km.n_clusters
    n = np.bincount(labels)
    N, d = X.shape
    var = []
    for c in range(m):
        ssq = (distance.cdist(X[np.where(labels == c)], [centers[c]], "euclidean") ** 2).sum(axis=0)
        var.append((1.0 / (n[c] - m)) * ssq if n[c] - m != 0 else float(10 ** 20) * ssq)
    bad = []
    for c, v in enumerate(var):
        v = [0.0 if x == -0.0 else x for x in v]
        var[c] = v
        if any(x < 0 for x in v):
            bad.append(c)
    n = [x for c, x in enumerate(n) if c not in bad]
    var = [x for c, x in enumerate(var) if c not in bad]
    terms = [n[c] * _log10(n[c]) - n[c] * _log10(N) - ((n[c] * d) / 2) * _log10(2 * np.pi)
             - (n[c] / 2) * _log10(var[c]) - ((n[c] - m) / 2) for c in range(len(n))]
    return np.sum(terms) - 0.5 * m * _log10(N)


class _Fit:
    """What _bic and _split_once read of a fitted sklearn KMeans."""
    __slots__ = ("labels_", "cluster_centers_", "n_clusters")

    def __init__(self, labels, centers, k):
        self.labels_, self.cluster_centers_, self.n_clusters = labels, centers, k


def _kmeans_fits_public(pts, ks, rs):
    from sklearn import cluster
    return [cluster.KMeans(n_clusters=k, init="k-means++", random_state=rs).fit(pts) for k in ks]


def _kmeans_fits_direct(pts, ks, rs):
    """The same fits - the same arithmetic on the same random draws, bit for bit - through the routines KMeans.fit itself calls
    (k-means++ seeding, one Lloyd run: n_init is 1 for this init), without what the estimator spends around them on a few
    hundred points: parameter and input validation per fit, the mean, norms and tolerance of the same points four times
    (1.5 ms of a 2 ms fit).  Private routines of scikit-learn: `_direct_ok` decides once per process whether they are there and
    answer as the estimator does."""
    from sklearn.cluster import _kmeans as K
    from sklearn.utils import check_random_state
    from sklearn.utils._openmp_helpers import _openmp_effective_n_threads
    from sklearn.utils.extmath import row_norms
    n_threads = _openmp_effective_n_threads()          # (as KMeans.fit asks: 1 in the host workers, which run with OMP_NUM_THREADS=1)
    X = np.array(pts, dtype=np.float64, order="C")
    mean = X.mean(axis=0)
    X -= mean
    norms = row_norms(X, squared=True)
    tol = K._tolerance(X, 1e-4)
    weight = np.ones(X.shape[0], dtype=X.dtype)
    fits = []
    for k in ks:
        init, _ = K._kmeans_plusplus(X, k, random_state=check_random_state(rs), x_squared_norms=norms, sample_weight=weight)
        labels, _inertia, centers, _n = K._kmeans_single_lloyd(X, weight, init, max_iter=300, verbose=0, tol=tol, n_threads=n_threads)
        fits.append(_Fit(labels, centers + mean, k))
    return fits


def _vq_kmeans_public(white, k, rs):
    from scipy.cluster.vq import kmeans
    return kmeans(white, k, seed=rs)[0] if rs is not None else kmeans(white, k)[0]


def _vq_kmeans_direct(white, k, rs):
    """scipy.cluster.vq.kmeans(white, k): twenty runs from k observations drawn at random, each until the mean distortion
    moves by 1e-5 or less, the code book of the lowest distortion - the same loop on the same draws with the same compiled
    routines (`_vq.vq`, `_vq.update_cluster_means`), without the array-namespace conversions SciPy wraps around every step
    (eight of ten parts of its time on a few hundred points).  Private module of SciPy: see `_direct_ok`."""
    from scipy._lib._util import check_random_state
    from scipy.cluster import _vq
    obs = np.ascontiguousarray(white, dtype=np.float64)
    rng = check_random_state(rs)
    best_book, best = None, np.inf
    for _ in range(20):
        book = obs[rng.choice(obs.shape[0], size=int(k), replace=False)]
        before, last = np.inf, np.inf
        while True:
            code, dist = _vq.vq(obs, book)
            before, last = last, np.add.reduce(dist) / dist.shape[0]          # (= np.mean(dist): the same sum, the same division)
            book, has = _vq.update_cluster_means(obs, code, book.shape[0])
            book = book[has]
            if not abs(before - last) > 1e-5:
                break
        if last < best:
            best_book, best = book, last
    return best_book


_DIRECT = None


def _direct_ok() -> bool:
    """Whether the two direct routes exist in the installed scikit-learn / SciPy and give what the public calls give, on seeded
    cases of the sizes met here (once per process; VAPOR_QC_DIRECT=0 keeps the public calls)."""
    global _DIRECT
    if _DIRECT is None:
        ok = os.environ.get("VAPOR_QC_DIRECT", "1") != "0"
        if ok:
            try:
                import warnings
                from scipy.cluster.vq import whiten
                rng = np.random.default_rng(20260704)
                with warnings.catch_warnings(), _one_thread():
                    warnings.simplefilter("ignore")
                    for n in (3, 40, 300):
                        pts = np.column_stack((rng.integers(0, 4000, n), rng.integers(0, 4000, n)))
                        pts[n // 2:, 0] += 3000
                        ks = list(range(1, min(5, n + 1)))
                        for a, b in zip(_kmeans_fits_public(pts, ks, 5), _kmeans_fits_direct(pts, ks, 5)):
                            ok = ok and np.array_equal(a.labels_, b.labels_) and np.array_equal(a.cluster_centers_, b.cluster_centers_)
                        for k in ks[1:]:
                            ok = ok and np.array_equal(_vq_kmeans_public(whiten(pts), k, 5), _vq_kmeans_direct(whiten(pts), k, 5))
            except Exception:              # noqa: BLE001 - another version of either library: the public calls
                ok = False
        _DIRECT = bool(ok)
    return _DIRECT


def _split_once(xs: Sequence[int], ys: Sequence[int]) -> List[List[List[int]]]:
    """k_means_cluster, SF:856-887: choose k in 1..4 by BIC, split with scipy's kmeans."""
    if not (max(xs) - min(xs) > 10 and max(ys) - min(ys) > 10):
        return [[list(xs), list(ys)]]
    from scipy.cluster.vq import vq, whiten
    seed = os.environ.get("VAPOR_QC_SEED")
    rs = int(seed) if seed else None
    pts = np.column_stack((np.asarray(xs), np.asarray(ys)))        # = np.array([[x, y] ...]) of SF:858
    ks = list(range(1, min([5, len(xs) + 1])))
    direct = _direct_ok()
    # the reference fits every k twice (SF:860-861: once for the BIC, once more only to test whether a
    # cluster came out empty); one fit serves both here - the draws are unseeded either way
    # a few hundred 2-D points: one OpenMP thread (the team start-up of sklearn's Lloyd loop costs several times the
    # fit itself at this size; the arithmetic is the same)
    with _one_thread():
        fits = (_kmeans_fits_direct if direct else _kmeans_fits_public)(pts, ks, rs)
    preds = [f.labels_ for f in fits]
    bic, bic_k = [], []
    for k in ks:
        if preds[k - 1].max() < k - 1:
            continue
        b = _bic(fits[k - 1], pts)
        if abs(b) < 10 ** 8:
            bic.append(b)
            bic_k.append(k)
    picked = bic_k[bic.index(max(bic))]
    if picked == 1:
        return [[list(xs), list(ys)]]
    white = whiten(pts)
    cent = (_vq_kmeans_direct if direct else _vq_kmeans_public)(white, picked, rs)
    idx, _ = vq(white, cent)
    return [[pts[idx == c, 0].tolist(), pts[idx == c, 1].tolist()] for c in range(picked)]


def x_means(xs: Sequence[int], ys: Sequence[int]) -> List[List[List[int]]]:
    """X_means_cluster + X_means_cluster_reformat, SF:2101-2116: split until stable."""
    parts = [p for p in _split_once(xs, ys) if not p == [[], []]]
    if parts == [[list(xs), list(ys)]]:
        return parts
    out = []
    for p in parts:
        out += x_means(p[0], p[1])
    return out


_warm_lock = None
_warm_done = False


def _warm() -> None:
    """The clustering libraries are imported on first use, inside the functions that need them - and chunks of a run are scored
    on several threads: two threads that meet their first window at the same moment would import scikit-learn / SciPy (C
    extensions, module locks taken in different orders) against each other and never come back.  The first caller imports
    everything, under a lock; afterwards the imports inside the functions are dictionary look-ups."""
    global _warm_lock, _warm_done
    if _warm_done:
        return
    import threading
    if _warm_lock is None:
        _warm_lock = _WARM_GUARD
    with _warm_lock:
        if not _warm_done:
            import scipy.cluster.vq          # noqa: F401
            import scipy.spatial.distance    # noqa: F401
            import sklearn.cluster           # noqa: F401
            try:
                import threadpoolctl         # noqa: F401
            except ImportError:
                pass
            _direct_ok()
            _warm_done = True


import threading as _threading

_WARM_GUARD = _threading.Lock()


def cluster_sizes(lower_j: Sequence[int], lower_i: Sequence[int]) -> List[float]:
    """sqrt(bounding-box area) of every repeat block (cluster_range_decide SF:372-378,
    cluster_size_decide SF:380-385)."""
    _warm()
    out = []
    for cx, cy in x_means(list(lower_j), list(lower_i)):
        out.append(np.sqrt((max(cx) - min(cx)) * (max(cy) - min(cy))))
    return out


def cluster_sizes_of_points(points: np.ndarray) -> List[float]:
    """cluster_sizes for an (n, 2) array [j, i] (what a host worker is sent: the lists are made on its side)."""
    return cluster_sizes(points[:, 0].tolist(), points[:, 1].tolist())


def qual_check_from_counts(n_hits: int, n_diag: int, n_lower: int, lower_points=None):
    """qual_check_repetitive_region, SF:1154-1171, from the device's counts.  `lower_points`
    is a callable returning (j array, i array) of the dots with j > i, evaluated only when the
    lower-triangle fraction falls in (0.1, 0.5)."""
    frac = float(n_lower) / float(n_hits)
    if n_hits > 0 and frac > 0.1 and frac < 0.5:
        lj, li = lower_points()
        sizes = cluster_sizes(lj, li)
    else:
        sizes = [0]
    return [float(n_diag) / float(n_hits), sizes]
