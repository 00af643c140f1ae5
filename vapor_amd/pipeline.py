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
from . import repeat_qc
from .drivers import Figure, Score, Window

_engine = None


def get_engine():
    """Process-wide default engine on the current device (LOCAL_RANK or 0)."""
    global _engine
    if _engine is None:
        from .dist import _device_ordinal
        from .engine import Engine
        _engine = Engine(_device_ordinal())
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
    Exceptions the reference would raise for one sequence are returned in its slot.

    All self dot plots of a round (one window size) are one plan; the statistics are read as arrays, and only a
    plot whose lower-triangle share falls into (0.1, 0.5) (SF:1165) is fetched dot by dot for the clustering."""
    n = len(seqs)
    out: List[Optional[object]] = [None] * n
    work = []                     # (slot, seq2)
    for t, s in enumerate(seqs):
        s2 = s.replace("X", "") if "X" in s else s   # ''.join([i for i in seq2 if not i == 'X']), SF:2031
        if s2.count("N") + s2.count("n") > 100:
            out[t] = ["Error", "Error"]
        else:
            work.append((t, s2))
    k = 10
    while work:
        idx_of: Dict[str, int] = {}
        useqs: List[str] = []
        which = np.empty(len(work), dtype=np.int64)
        for w, (_t, s2) in enumerate(work):
            q = idx_of.get(s2)
            if q is None:
                q = idx_of[s2] = len(useqs)
                useqs.append(s2)
            which[w] = q
        pairs = np.zeros(len(useqs), dtype=L.PAIR_DTYPE)
        pairs["seq1"] = pairs["seq2"] = np.arange(len(useqs))
        pairs["k"] = k
        ss = engine.seqset(useqs)
        plan = engine.plan(ss, pairs)
        st = plan.run()[which]
        code = st[:, L.ST_STATUS]
        nh, nd, nl = st[:, L.ST_N_HITS], st[:, L.ST_N_DIAG], st[:, L.ST_N_LOWER]
        with np.errstate(divide="ignore", invalid="ignore"):
            frac = nl.astype(np.float64) / nh.astype(np.float64)
            diag = nd.astype(np.float64) / nh.astype(np.float64)
        band = (code == 0) & (nh > 0) & (frac > 0.1) & (frac < 0.5)
        pts = {}
        need = np.flatnonzero(band)
        if len(need):
            hits, _f, off = plan.fetch_hits(which[need].tolist(), want_flags=False)
            for q, w in enumerate(need):
                h = hits[off[q]:off[q + 1]]
                h = h[h[:, 0] > h[:, 1]]
                pts[int(w)] = h[np.lexsort((h[:, 1], h[:, 0]))]
        plan.close()
        ss.close()
        nxt = []
        for w, (t, s2) in enumerate(work):
            if code[w] != 0:
                try:
                    _raise_for_status(st[w])
                except Exception as e:      # noqa: BLE001
                    out[t] = e
                continue
            if nh[w] == 0:
                # SF:2035, 2045 at the first size; later an empty plot divides by zero in SF:1171
                out[t] = ["Error", "Error"] if k == 10 else ZeroDivisionError("float division by zero")
                continue
            if band[w]:
                try:
                    qc = [float(diag[w]), repeat_qc.cluster_sizes(pts[w][:, 0].tolist(), pts[w][:, 1].tolist())]
                except Exception as e:          # noqa: BLE001 - e.g. the clustering libraries' own errors
                    out[t] = e
                    continue
            else:
                qc = [float(diag[w]), [0]]
            if k > 30 or qc[0] > region_QC_Cff or sum(qc[1]) / float(len(s2)) < 0.3:
                out[t] = [k, qc]
            else:
                nxt.append((t, s2))
        work = nxt
        k += 10
    return out  # type: ignore[return-value]


# ------------------------------------------------------------------------------------------
# scorer requests
# ------------------------------------------------------------------------------------------
_FLAGS = {"s1": L.PF_C1, "s2": L.PF_C2, "s3": L.PF_C1 | L.PF_DIR}
_KIND = {"del": 0, "s1": 1, "s2": 2, "s3": 3}


def score_requests(engine, reqs: Sequence[Score]) -> List[object]:
    """Evaluates every Score request on the device, per-read reduction included (finish_kernel): per request a list
    with one score (float) or None per read, or the exception the reference would raise (KeyError for a read with a
    base outside invert_base's alphabet, SF:1421).

    One sequence set and one plan for all requests: request t is "locus" t of the plan, its reads are the plan's
    reads in order.  The arrays are put together with numpy per request, not per read."""
    seqs: List[str] = []
    upper: List[bool] = []
    seq1, seq2, off2, kk, flg = [], [], [], [], []           # pair columns, one array per request block
    ra, aa, rb, ab, kind, locus, lref, lalt = [], [], [], [], [], [], [], []
    first_read, n_pairs = [], 0
    read_seq_first = []                                      # per request: index of its first read sequence
    n_reads_tot = 0
    for t, r in enumerate(reqs):
        n = len(r.reads)
        first_read.append(n_reads_tot)
        if n == 0:
            read_seq_first.append(len(seqs))
            continue
        # allele sequences of this request: as given, and (abs_dis_m1b, SF:183-184) upper-cased where that differs
        ri = len(seqs)
        seqs += [r.ref_seq, r.alt_seq]
        upper += [False, False]
        if r.kind in ("del", "s1") and not (_is_upper(r.ref_seq) and _is_upper(r.alt_seq)):
            ui = len(seqs)
            seqs += [r.ref_seq, r.alt_seq]
            upper += [True, True]
        else:
            ui = ri
        q0 = len(seqs)
        read_seq_first.append(q0)
        seqs += [x[0] for x in r.reads]
        upper += [False] * n
        q = np.arange(q0, q0 + n, dtype=np.int32)
        miss = np.fromiter((int(x[1]) for x in r.reads), dtype=np.int32, count=n)

        def block(allele0, fl):
            nonlocal n_pairs
            base = n_pairs
            seq1.append(np.repeat(q, 2))
            seq2.append(np.tile(np.array([allele0, allele0 + 1], dtype=np.int32), n))
            off2.append(np.repeat(miss, 2))
            kk.append(np.full(2 * n, int(r.k), dtype=np.int32))
            flg.append(np.full(2 * n, fl, dtype=np.uint32))
            n_pairs += 2 * n
            return base + 2 * np.arange(n, dtype=np.int32)

        if r.kind == "del":
            if ui == ri:
                pa = pb = block(ri, L.PF_C1 | L.PF_C2)       # upper-casing changes nothing: one fill serves both scorers
            else:
                pa = block(ui, L.PF_C1)
                pb = block(ri, L.PF_C2)
        else:
            pa = pb = block(ui if r.kind == "s1" else ri, _FLAGS[r.kind])
        ra.append(pa); aa.append(pa + 1); rb.append(pb); ab.append(pb + 1)
        kind.append(np.full(n, _KIND[r.kind], dtype=np.int32))
        locus.append(np.full(n, t, dtype=np.int32))
        lref.append(np.full(n, len(r.ref_seq), dtype=np.int32))
        lalt.append(np.full(n, len(r.alt_seq), dtype=np.int32))
        n_reads_tot += n
    if n_reads_tot == 0:
        return [[] for _ in reqs]
    pairs = np.zeros(n_pairs, dtype=L.PAIR_DTYPE)
    for name, col in (("seq1", seq1), ("seq2", seq2), ("off2", off2), ("k", kk), ("flags", flg)):
        pairs[name] = np.concatenate(col)
    table = np.zeros(n_reads_tot, dtype=L.READ_DTYPE)
    for name, col in (("ref_a", ra), ("alt_a", aa), ("ref_b", rb), ("alt_b", ab), ("kind", kind), ("locus", locus),
                      ("len_ref", lref), ("len_alt", lalt)):
        table[name] = np.concatenate(col)
    ss = engine.seqset(seqs, upper)
    try:
        plan = engine.plan(ss, pairs)
        try:
            plan.set_reads(table, len(reqs))
            plan.run_loci(want_host=False, want_scores=True)
            sc = plan.read_scores[:n_reads_tot]
        finally:
            plan.close()
        # pairs the library rejects, as the reference would have raised: a read k-mer with a base outside
        # invert_base's alphabet (SF:1421), a sequence beyond the device's 16-bit positions
        bad_inv = np.asarray(ss.n_invalid) > 0
        lens = np.asarray(ss.lens)
    finally:
        ss.close()
    out: List[object] = []
    for t, r in enumerate(reqs):
        n = len(r.reads)
        if n == 0:
            out.append([])
            continue
        q0 = read_seq_first[t]
        if k_unsupported(r.k) or max(len(r.ref_seq), len(r.alt_seq)) > L.MAX_SEQ_LEN or int(lens[q0:q0 + n].max()) > L.MAX_SEQ_LEN:
            out.append(ValueError("sequence longer than %d bases or unsupported window size" % L.MAX_SEQ_LEN))
            continue
        inv = bad_inv[q0:q0 + n] & (lens[q0:q0 + n] - int(r.k) + 1 > 0)
        if inv.any():
            out.append(KeyError("invert_base"))              # what SF:1421 raises on a base outside ATCGN/atcgn
            continue
        v = sc[first_read[t]:first_read[t] + n]
        out.append([None if x != x else x for x in v.tolist()])
    return out


def scorer_outputs(engine, kind: str, ref_seq: str, alt_seq: str, x, k):
    """[a, b] of calcu_vapor_single_read_score_{abs_dis_m1b, within_10Perc_m1b, directed_dis_m1b_redefine_diagnal}
    (kind 's1', 's2', 's3'; SF:182-203, 277-294, 241-257) for one read: the two dot plots' statistics from the
    device, the scorer's own gates and ratios on the host in float64 (vapor_amd.finish)."""
    from . import finish
    up = kind == "s1"
    ss = engine.seqset([x[0], ref_seq, alt_seq], [False, up, up])
    try:
        st = engine.score(ss, engine.make_pairs([(0, 1, int(x[1]), int(k), _FLAGS[kind]), (0, 2, int(x[1]), int(k), _FLAGS[kind])]))
    finally:
        ss.close()
    _raise_for_status(st[0])
    _raise_for_status(st[1])
    fn = {"s1": finish.score_abs_dis_m1b, "s2": finish.score_within_10Perc_m1b,
          "s3": finish.score_directed_dis_m1b_redefine_diagnal}[kind]
    return fn(st[0], st[1], len(ref_seq), len(alt_seq))


def _is_upper(s: str) -> bool:
    return s.isupper() or s.upper() == s


def k_unsupported(k) -> bool:
    return int(k) not in (10, 20, 30, 40)


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


def _prefetch_threads(n_gens: int) -> int:
    """Threads that start the loci of a batch (VAPOR_PREFETCH_THREADS; default up to 12, 1 = off)."""
    import os
    from . import seqio
    if n_gens < 16 or not getattr(seqio.get_backend(), "threads_ok", False) or os.environ.get("VAPOR_BAM_NATIVE", "1") == "0":
        return 1
    want = os.environ.get("VAPOR_PREFETCH_THREADS")
    if want is not None:
        return max(1, int(want))
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))       # ranks sharing this host's cores (torchrun)
    return max(1, min(12, _usable_cores() // ranks_here))


def _usable_cores() -> int:
    """Cores this process may use: its affinity mask, cut to the container's CPU quota where there is one."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1") and int(period) > 0:
                cores = max(1, min(cores, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


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

    # A stretch of a locus generator holds its read extraction (window from the FASTA, reads of the region from the BAM:
    # BGZF inflation in the library's host helper, which releases the GIL) - the first stretch for the deletion and
    # insertion drivers, the one after the window refinements for the others (reads are fetched only when the windows
    # pass, CLI:334-367 via SF:1512-1530).  With a backend whose handles are per thread the generators of a batch advance on
    # a few threads, so that one locus inflates while another's Python runs.
    n_thr = _prefetch_threads(len(gens))
    pool = None
    if n_thr > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=n_thr)

    def spread(work):
        """`work`: argument tuples of advance(); in slices over the pool (a future per item costs as much as a short stretch)."""
        if pool is None or len(work) < 16:
            for a in work:
                advance(*a)
            return
        step = max(1, len(work) // (n_thr * 8))

        def some(k):
            for a in work[k:k + step]:
                advance(*a)
        list(pool.map(some, range(0, len(work), step)))

    try:
        spread([(t, True) for t in range(len(gens))])
        while pending:
            idx = sorted(pending)
            ans = _answer(engine, [pending[t] for t in idx], figure_fn)
            spread([(t, False, a) for t, a in zip(idx, ans)])
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    return results
