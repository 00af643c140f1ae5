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

import threading

_engine = None
_engine_lock = threading.RLock()      # made at import: the chunk threads of cli.score_jobs reach engine_slot() together


def get_engine():
    """Process-wide default engine on the current device (LOCAL_RANK or 0)."""
    global _engine
    with _engine_lock:
        if _engine is None:
            from .dist import _device_ordinal
            from .engine import Engine
            _engine = Engine(_device_ordinal())
        return _engine


def set_engine(e) -> None:
    global _engine, _more_engines
    _engine = e
    _more_engines = []


_more_engines: list = []


def get_engines(n: int) -> list:
    """`n` engines on the current device, the default one first: one per thread that scores chunks (a library context serves
    one host thread).  An engine that is not this package's own (the tests' stand-in) is shared: it has no such rule."""
    first = get_engine()
    from .engine import Engine
    if not isinstance(first, Engine):
        return [first] * n
    from .dist import _device_ordinal
    while len(_more_engines) < n - 1:
        _more_engines.append(Engine(_device_ordinal()))
    return [first] + _more_engines[:n - 1]


def engine_slot(k: int):
    """The engine of the k-th thread that scores chunks, made on that thread's first call (the second context of a run comes
    up while the first chunk is already being prepared on the first): get_engines(k + 1)[k]."""
    first = get_engine()
    from .engine import Engine
    if k == 0 or not isinstance(first, Engine):
        return first
    with _engine_lock:
        if len(_more_engines) >= k:
            return _more_engines[k - 1]
    from .dist import _device_ordinal
    made = Engine(_device_ordinal())                   # (outside the lock: the other threads' first calls go on)
    with _engine_lock:
        while len(_more_engines) < k - 1:              # (slots are taken in order; a gap is filled by whoever comes)
            _more_engines.append(Engine(_device_ordinal()))
        if len(_more_engines) >= k:                    # another thread was faster
            made.close()
            return _more_engines[k - 1]
        _more_engines.append(made)
        return made


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

# Windows that met the X-means branch of the repeat check since the process started (SF:1165-1167), by outcome: `one_cluster`
# - BIC keeps one cluster, the reference's answer whatever its unseeded draws; `split` - more than one cluster: seed-dependent
# in the reference (and on current SciPy it raises there, SF:878); `sizes_decide` - of either kind, the windows whose diagonal
# share is at most 0.4, i.e. the only ones where the cluster sizes can change the window size at all (SF:2038).
qc_counts = {"band": 0, "one_cluster": 0, "split": 0, "sizes_decide": 0, "raised": 0}
_qc_lock = threading.Lock()


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
        # the clustering of the windows in the band (sklearn / scipy, ~1 ms each, unseeded as in the reference) goes to the
        # host worker processes when there are enough of them to pay for the pickling
        sizes_of: Dict[int, object] = {}
        if len(pts) >= 8:
            from . import hostpool
            hp = hostpool.get()
            if hp is not None:
                futs = {w: hp.submit("vapor_amd.repeat_qc", "cluster_sizes_of_points", p) for w, p in pts.items()}
                for w, f in futs.items():
                    try:
                        sizes_of[w] = f.result()
                    except hostpool.WorkerLost:
                        pass                    # (clustered below, by this process)
                    except Exception as e:      # noqa: BLE001 - e.g. the clustering libraries' own errors
                        sizes_of[w] = e
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
                    got = sizes_of[w] if w in sizes_of else repeat_qc.cluster_sizes(pts[w][:, 0].tolist(), pts[w][:, 1].tolist())
                    if isinstance(got, BaseException):
                        raise got
                    qc = [float(diag[w]), got]
                    with _qc_lock:
                        qc_counts["band"] += 1
                        qc_counts["one_cluster" if len(got) == 1 else "split"] += 1
                        qc_counts["sizes_decide"] += int(not qc[0] > region_QC_Cff)
                except Exception as e:          # noqa: BLE001 - e.g. the clustering libraries' own errors
                    with _qc_lock:
                        qc_counts["band"] += 1
                        qc_counts["raised"] += 1
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
    seqs: List[str] = []              # sequences that travel as bytes: windows, insertion payloads, reads
    derived: List[tuple] = []         # sequences described instead (drivers.Allele.segs; the str.upper() twins of abs_dis_m1b): (segments, upper)
    lit_of: Dict[int, int] = {}       # id(string) -> its index among the literals (a window is uploaded once per batch)
    nocomp_of: Dict[int, bool] = {}

    def lit(sq: str) -> int:
        q = lit_of.get(id(sq))
        if q is None:
            q = lit_of[id(sq)] = len(seqs)
            seqs.append(sq)
        return q

    derived_of: Dict[tuple, int] = {}  # (id(string), upper) -> its derived index: requests that share a window share its twins too

    def describe(sq, up: bool) -> int:
        got = derived_of.get((id(sq), up))
        if got is not None:
            return got
        got = derived_of[(id(sq), up)] = _describe(sq, up)
        return got

    def _describe(sq, up: bool) -> int:
        """Index of `sq` (upper-cased when `up`) in the set: a literal, or - (d + 1) for derived sequence d (their place behind
        the literals is known when all literals are)."""
        segs = getattr(sq, "segs", None)
        if segs is not None and len(segs) > L.MAX_SEGMENTS:
            segs = None               # (more blocks than a descriptor holds, include/vapor_hip.h: this allele travels as bytes)
        if segs is not None and segs:
            # (a reversed slice needs a parent complementary() keeps whole, SF:471-478: the library checks the same)
            for par, _o, _n, rc in segs:
                if rc:
                    bad = nocomp_of.get(id(par))
                    if bad is None:
                        bad = nocomp_of[id(par)] = _NOCOMP.search(par) is not None
                    if bad:
                        segs = None
                        break
        if segs:
            derived.append(([(lit(par), o, n, rc) for par, o, n, rc in segs], up))
            return -len(derived)
        if up:
            derived.append(([(lit(sq), 0, len(sq), False)], True))
            return -len(derived)
        return lit(sq)

    # per request: scalars only; the per-read and per-pair columns are expanded from them in one go below
    rq_n, rq_k, rq_kind, rq_lref, rq_lalt, rq_blk_a, rq_blk_b = [], [], [], [], [], [], []
    bl_rq, bl_ref, bl_alt, bl_flags = [], [], [], []         # blocks of pairs: (request, its two allele sequences, pair flags)
    miss: List[int] = []
    first_read, read_seq_first = [], []
    n_reads_tot = 0
    for t, r in enumerate(reqs):
        n = len(r.reads)
        first_read.append(n_reads_tot)
        if n == 0:
            read_seq_first.append(len(seqs))
            continue
        # allele sequences of this request: as given, and (abs_dis_m1b, SF:183-184) upper-cased where that differs
        ri, ai = describe(r.ref_seq, False), describe(r.alt_seq, False)
        if r.kind in ("del", "s1") and not (_is_upper(r.ref_seq) and _is_upper(r.alt_seq)):
            uri, uai = describe(r.ref_seq, True), describe(r.alt_seq, True)
        else:
            uri, uai = ri, ai
        read_seq_first.append(len(seqs))
        seqs += [x[0] for x in r.reads]                      # (reads are never shared between requests: no look-up)
        miss += [int(x[1]) for x in r.reads]
        k = len(rq_n)
        if r.kind == "del" and uri != ri:
            blk_a = len(bl_rq)
            bl_rq += [k, k]; bl_ref += [uri, ri]; bl_alt += [uai, ai]; bl_flags += [L.PF_C1, L.PF_C2]
            blk_b = blk_a + 1
        else:
            # (a deletion whose alleles are upper case already: one fill serves both scorers)
            blk_a = blk_b = len(bl_rq)
            bl_rq.append(k)
            plain = r.kind in ("del", "s2", "s3")
            bl_ref.append(ri if plain else uri); bl_alt.append(ai if plain else uai)
            bl_flags.append(L.PF_C1 | L.PF_C2 if r.kind == "del" else _FLAGS[r.kind])
        rq_n.append(n); rq_k.append(int(r.k)); rq_kind.append(_KIND[r.kind]); rq_lref.append(len(r.ref_seq)); rq_lalt.append(len(r.alt_seq))
        rq_blk_a.append(blk_a); rq_blk_b.append(blk_b)
        n_reads_tot += n
    if n_reads_tot == 0:
        return [[] for _ in reqs]
    i32 = np.int32
    nz = [t for t, r in enumerate(reqs) if len(r.reads)]     # requests with reads, in order (rq_* are indexed by position here)
    rq_n_a = np.asarray(rq_n, dtype=np.int64)
    rq_first = np.concatenate(([0], np.cumsum(rq_n_a)[:-1]))             # first read of a request in the read table
    rq_q0 = np.asarray([read_seq_first[t] for t in nz], dtype=np.int64)  # its first read sequence
    miss_a = np.asarray(miss, dtype=i32)
    # blocks -> pairs: block b holds, per read i of its request, the pairs (read, allele) and (read, allele + 1)
    bl_rq_a = np.asarray(bl_rq, dtype=np.int64)
    bl_n = rq_n_a[bl_rq_a]
    bl_base = 2 * np.concatenate(([0], np.cumsum(bl_n)[:-1]))            # first pair of a block
    br_blk = np.repeat(np.arange(len(bl_rq_a)), bl_n)                     # per (block, read): its block
    br_i = np.arange(int(bl_n.sum())) - np.repeat(bl_base // 2, bl_n)     # ... and the read's index in the request
    br_rq = bl_rq_a[br_blk]
    pairs = np.zeros(2 * len(br_blk), dtype=L.PAIR_DTYPE)
    pairs["seq1"] = np.repeat((rq_q0[br_rq] + br_i).astype(i32), 2)
    n_lit = len(seqs)

    def placed(v):                                            # derived sequence d: behind the literals
        v = np.asarray(v, dtype=np.int64)
        return np.where(v < 0, n_lit - 1 - v, v).astype(i32)
    al = np.empty(2 * len(br_blk), dtype=i32)
    al[0::2] = placed(bl_ref)[br_blk]
    al[1::2] = placed(bl_alt)[br_blk]
    pairs["seq2"] = al
    pairs["off2"] = np.repeat(miss_a[rq_first[br_rq] + br_i], 2)
    pairs["k"] = np.repeat(np.asarray(rq_k, dtype=i32)[br_rq], 2)
    pairs["flags"] = np.repeat(np.asarray(bl_flags, dtype=np.uint32)[br_blk], 2)
    # read table: read i of request r takes its statistics from pairs base(block) + 2 i (+ 1 for the other allele)
    rd_rq = np.repeat(np.arange(len(rq_n_a)), rq_n_a)
    rd_i = np.arange(n_reads_tot) - rq_first[rd_rq]
    pa = (bl_base[np.asarray(rq_blk_a, dtype=np.int64)][rd_rq] + 2 * rd_i).astype(i32)
    pb = (bl_base[np.asarray(rq_blk_b, dtype=np.int64)][rd_rq] + 2 * rd_i).astype(i32)
    table = np.zeros(n_reads_tot, dtype=L.READ_DTYPE)
    table["ref_a"], table["alt_a"], table["ref_b"], table["alt_b"] = pa, pa + 1, pb, pb + 1
    table["kind"] = np.asarray(rq_kind, dtype=i32)[rd_rq]
    table["locus"] = np.asarray(nz, dtype=i32)[rd_rq]
    table["len_ref"] = np.asarray(rq_lref, dtype=i32)[rd_rq]
    table["len_alt"] = np.asarray(rq_lalt, dtype=i32)[rd_rq]
    ss = engine.seqset(seqs, None, derived) if derived else engine.seqset(seqs)
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
    # per request with reads: longest read, any read the library could not have hashed (one pass over all reads)
    rd_q = rq_q0[rd_rq] + rd_i
    rd_len = lens[rd_q]
    max_read = np.maximum.reduceat(rd_len, rq_first)
    rd_inv = bad_inv[rd_q] & (rd_len - np.asarray(rq_k, dtype=np.int64)[rd_rq] + 1 > 0)
    any_inv = np.logical_or.reduceat(rd_inv, rq_first)
    vals = sc.tolist()
    if sc.size and bool(np.isnan(sc).any()):
        vals = [None if x != x else x for x in vals]
    out: List[object] = []
    j = 0
    for t, r in enumerate(reqs):
        n = len(r.reads)
        if n == 0:
            out.append([])
            continue
        if k_unsupported(r.k) or max(len(r.ref_seq), len(r.alt_seq)) > L.MAX_SEQ_LEN or int(max_read[j]) > L.MAX_SEQ_LEN:
            out.append(ValueError("sequence longer than %d bases or unsupported window size" % L.MAX_SEQ_LEN))
        elif any_inv[j]:
            out.append(KeyError("invert_base"))              # what SF:1421 raises on a base outside ATCGN/atcgn
        else:
            out.append(vals[first_read[t]:first_read[t] + n])
        j += 1
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


import re as _re

_NOCOMP = _re.compile("[^ACGTNacgtn]")       # what complementary() drops (SF:471-478)


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
    if figure_fn is not None:
        figs = [r for r in reqs if isinstance(r, Figure)]
        batch = getattr(figure_fn, "batch", None)      # (figures.make_event_figure_1: one device pass, drawing in worker processes)
        if figs and batch is not None:
            batch(figs, engine)             # (the chunk's own engine: a library context serves one host thread)
        else:
            for r in figs:
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
    finally:
        wait = getattr(figure_fn, "wait", None)
        if wait is not None:
            wait()


def _prefetch_threads(n_gens: int) -> int:
    """Threads that start the loci of a batch (VAPOR_PREFETCH_THREADS; default up to 8, 1 = off)."""
    import os
    from . import seqio
    if n_gens < 16 or not getattr(seqio.get_backend(), "threads_ok", False) or os.environ.get("VAPOR_BAM_NATIVE", "1") == "0":
        return 1
    want = os.environ.get("VAPOR_PREFETCH_THREADS")
    if want is not None:
        return max(1, int(want))
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))       # ranks sharing this host's cores (torchrun)
    return max(1, min(8, _usable_cores() // ranks_here))


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
        wait = getattr(figure_fn, "wait", None)
        if wait is not None:
            wait()                                   # the batch's figures are on disk when it returns
    return results
