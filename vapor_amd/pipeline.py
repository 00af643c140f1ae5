"""Executors for the driver generators (vapor_amd.drivers): turn their Window / Score requests
into batches for the HIP library and feed the results back.

`run_sync`  - one locus at a time (what the reference does);
`run_batch` - many loci in lockstep: all pending requests of a round become ONE sequence set
              and ONE plan on the device (reads that share an allele window share its hash table).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from . import finish, repeat_qc
from .drivers import Figure, Score, Window

_engine = None


def get_engine():
    """Process-wide default engine on the current device (LOCAL_RANK or 0)."""
    global _engine
    if _engine is None:
        import os
        from .engine import Engine
        _engine = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _engine


def set_engine(e) -> None:
    global _engine
    _engine = e


class _SeqTable:
    """Deduplicating builder of the sequence list of one device batch."""

    def __init__(self):
        self.seqs: List[str] = []
        self.upper: List[bool] = []
        self._idx: Dict[Tuple[str, bool], int] = {}

    def add(self, s: str, upper: bool = False) -> int:
        key = (s, upper)
        i = self._idx.get(key)
        if i is None:
            i = len(self.seqs)
            self._idx[key] = i
            self.seqs.append(s)
            self.upper.append(upper)
        return i


def _raise_for_status(st_row) -> None:
    code = int(st_row[L.ST_STATUS])
    if code == 0:
        return
    if code == L.E_KEYERROR:
        raise KeyError("invert_base")      # what SF:1421 raises on a base outside ATCGN/atcgn
    if code == L.E_ARG:
        raise ValueError("sequence longer than %d bases or unsupported window size" % L.MAX_SEQ_LEN)
    raise RuntimeError("libvapor_hip pair status %d" % code)


# ------------------------------------------------------------------------------------------
# window_size_refine for many sequences at once
# ------------------------------------------------------------------------------------------

def refine_windows(engine, seqs: Sequence[str], region_QC_Cff: float = 0.4) -> List[list]:
    """window_size_refine (SF:2030-2046) for every sequence; returns [[w, qc] | ['Error','Error']].
    Exceptions the reference would raise for one sequence are returned in its slot."""
    n = len(seqs)
    out: List[Optional[object]] = [None] * n
    work = []                     # (slot, seq2)
    for t, s in enumerate(seqs):
        s2 = s.replace("X", "")              # ''.join([i for i in seq2 if not i == 'X']), SF:2031
        if s2.count("N") + s2.count("n") > 100:
            out[t] = ["Error", "Error"]
        else:
            work.append((t, s2))
    k = 10
    while work:
        tab = _SeqTable()
        rows = []
        for _t, s2 in work:
            q = tab.add(s2)
            rows.append((q, q, 0, k, 0))
        ss = engine.seqset(tab.seqs)
        plan = engine.plan(ss, engine.make_pairs(rows))
        st = plan.run()
        nxt = []
        need_pts = []
        for w, (t, s2) in enumerate(work):
            code = int(st[w, L.ST_STATUS])
            if code != 0:
                try:
                    _raise_for_status(st[w])
                except Exception as e:      # noqa: BLE001
                    out[t] = e
                continue
            nh, nd, nl = int(st[w, L.ST_N_HITS]), int(st[w, L.ST_N_DIAG]), int(st[w, L.ST_N_LOWER])
            if nh == 0:
                if k == 10:
                    out[t] = ["Error", "Error"]                      # SF:2035, 2045
                else:
                    out[t] = ZeroDivisionError("float division by zero")   # SF:1171 on an empty plot
                continue
            frac = float(nl) / float(nh)
            if frac > 0.1 and frac < 0.5:
                need_pts.append(w)
        pts = {}
        if need_pts:
            hits, _f, off = plan.fetch_hits(need_pts, want_flags=False)
            for q, w in enumerate(need_pts):
                h = hits[off[q]:off[q + 1]]
                h = h[h[:, 0] > h[:, 1]]
                h = h[np.lexsort((h[:, 1], h[:, 0]))]
                pts[w] = h
        for w, (t, s2) in enumerate(work):
            if out[t] is not None:
                continue
            nh, nd, nl = int(st[w, L.ST_N_HITS]), int(st[w, L.ST_N_DIAG]), int(st[w, L.ST_N_LOWER])
            try:
                qc = repeat_qc.qual_check_from_counts(
                    nh, nd, nl, (lambda w=w: (pts[w][:, 0].tolist(), pts[w][:, 1].tolist())))
            except Exception as e:          # noqa: BLE001 - e.g. the clustering libraries' own errors
                out[t] = e
                continue
            if k > 30 or qc[0] > region_QC_Cff or sum(qc[1]) / float(len(s2)) < 0.3:
                out[t] = [k, qc]
            else:
                nxt.append((t, s2))
        plan.close()
        ss.close()
        work = nxt
        k += 10
    return out  # type: ignore[return-value]


# ------------------------------------------------------------------------------------------
# scorer requests
# ------------------------------------------------------------------------------------------
_FLAGS = {"s1": L.PF_C1, "s2": L.PF_C2, "s3": L.PF_C1 | L.PF_DIR}
_FINISH = {"s1": finish.score_abs_dis_m1b, "s2": finish.score_within_10Perc_m1b,
           "s3": finish.score_directed_dis_m1b_redefine_diagnal}


def score_requests(engine, reqs: Sequence[Score]) -> List[object]:
    """Evaluates every Score request; result per request: list of [a, b] per read ('del':
    (list_s1, list_s2)), or the exception the reference would raise."""
    tab = _SeqTable()
    rows = []
    layout = []            # per request: list of (kind, first_row) blocks, rows = 2 per read
    for r in reqs:
        blocks = []
        kinds = ["s1", "s2"] if r.kind == "del" else [r.kind]
        merged = False
        if r.kind == "del" and r.ref_seq.upper() == r.ref_seq and r.alt_seq.upper() == r.alt_seq:
            merged = True                       # upper-casing changes nothing: one fill serves both
        done_first = None
        for kind in kinds:
            if merged and done_first is not None:
                blocks.append((kind, done_first))
                continue
            up = kind == "s1"
            ri = tab.add(r.ref_seq, up and not merged)
            ai = tab.add(r.alt_seq, up and not merged)
            fl = (L.PF_C1 | L.PF_C2) if merged else _FLAGS[kind]
            first = len(rows)
            for x in r.reads:
                q = tab.add(x[0])
                rows.append((q, ri, int(x[1]), int(r.k), fl))
                rows.append((q, ai, int(x[1]), int(r.k), fl))
            blocks.append((kind, first))
            done_first = first
        layout.append(blocks)
    if not rows:
        return [([], []) if r.kind == "del" else [] for r in reqs]
    ss = engine.seqset(tab.seqs, tab.upper)
    plan = engine.plan(ss, engine.make_pairs(rows))
    st = plan.run().copy()
    plan.close()
    ss.close()
    out: List[object] = []
    for r, blocks in zip(reqs, layout):
        lr, la = len(r.ref_seq), len(r.alt_seq)
        res = []
        err = None
        for kind, first in blocks:
            lst = []
            for t in range(len(r.reads)):
                a, b = st[first + 2 * t], st[first + 2 * t + 1]
                try:
                    _raise_for_status(a)
                    _raise_for_status(b)
                    lst.append(_FINISH[kind](a, b, lr, la))
                except Exception as e:      # noqa: BLE001
                    err = e
                    break
            if err is not None:
                break
            res.append(lst)
        if err is not None:
            out.append(err)
        else:
            out.append(tuple(res) if r.kind == "del" else res[0])
    return out


# ------------------------------------------------------------------------------------------
# executors
# ------------------------------------------------------------------------------------------

def _answer(engine, reqs: Sequence[object], figure_fn) -> List[object]:
    res: List[object] = [None] * len(reqs)
    wi = [t for t, r in enumerate(reqs) if isinstance(r, Window)]
    si = [t for t, r in enumerate(reqs) if isinstance(r, Score)]
    if wi:
        for t, v in zip(wi, refine_windows(engine, [reqs[t].seq for t in wi])):
            res[t] = v
    if si:
        for t, v in zip(si, score_requests(engine, [reqs[t] for t in si])):
            res[t] = v
    for t, r in enumerate(reqs):
        if isinstance(r, Figure) and figure_fn is not None:
            figure_fn(r)
    return res


def run_sync(gen, engine=None, figure_fn: Optional[Callable] = None):
    """Drive one locus generator to completion; returns its score list."""
    engine = engine or get_engine()
    try:
        req = next(gen)
        while True:
            ans = _answer(engine, [req], figure_fn)[0]
            req = gen.throw(ans) if isinstance(ans, BaseException) else gen.send(ans)
    except StopIteration as e:
        return e.value


def run_batch(gens: Sequence, engine=None, figure_fn: Optional[Callable] = None) -> List[object]:
    """Drive many locus generators in lockstep.  Returns, per generator, its score list or the
    exception it ended with."""
    engine = engine or get_engine()
    results: List[object] = [None] * len(gens)
    pending: Dict[int, object] = {}

    def advance(t, first=False, value=None):
        g = gens[t]
        try:
            if first:
                req = next(g)
            elif isinstance(value, BaseException):
                req = g.throw(value)
            else:
                req = g.send(value)
            pending[t] = req
        except StopIteration as e:
            results[t] = e.value
            pending.pop(t, None)
        except Exception as e:              # noqa: BLE001 - recorded for the caller to re-raise in order
            results[t] = e
            pending.pop(t, None)

    for t in range(len(gens)):
        advance(t, first=True)
    while pending:
        idx = sorted(pending)
        ans = _answer(engine, [pending[t] for t in idx], figure_fn)
        for t, a in zip(idx, ans):
            advance(t, value=a)
    return results
