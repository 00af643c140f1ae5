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


def _split_once(xs: Sequence[int], ys: Sequence[int]) -> List[List[List[int]]]:
    """k_means_cluster, SF:856-887: choose k in 1..4 by BIC, split with scipy's kmeans."""
    if not (max(xs) - min(xs) > 10 and max(ys) - min(ys) > 10):
        return [[list(xs), list(ys)]]
    from scipy.cluster.vq import kmeans, vq, whiten
    from sklearn import cluster
    seed = os.environ.get("VAPOR_QC_SEED")
    rs = int(seed) if seed else None
    pts = np.column_stack((np.asarray(xs), np.asarray(ys)))        # = np.array([[x, y] ...]) of SF:858
    ks = list(range(1, min([5, len(xs) + 1])))
    # the reference fits every k twice (SF:860-861: once for the BIC, once more only to test whether a
    # cluster came out empty); one fit serves both here - the draws are unseeded either way
    # a few hundred 2-D points: one OpenMP thread (the team start-up of sklearn's Lloyd loop costs several times the
    # fit itself at this size; the arithmetic is the same)
    with _one_thread():
        fits = [cluster.KMeans(n_clusters=k, init="k-means++", random_state=rs).fit(pts) for k in ks]
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
    if rs is not None:
        cent, _ = kmeans(white, picked, seed=rs)
    else:
        cent, _ = kmeans(white, picked)
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


def cluster_sizes(lower_j: Sequence[int], lower_i: Sequence[int]) -> List[float]:
    """sqrt(bounding-box area) of every repeat block (cluster_range_decide SF:372-378,
    cluster_size_decide SF:380-385)."""
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
