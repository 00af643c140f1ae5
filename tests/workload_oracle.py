"""What the CPU oracle says about a kernel-level workload (vapor_amd/workload.py): test infrastructure.

For every read of the selected loci the two dot plots are filled ONCE by the C oracle (oracle.pair_stats: dots in dotdata's
order, all sixteen statistics), the read is scored by the oracle's restatements of the reference's scorers on those plots
(abs_dis_m1b SF:182-203, within_10Perc_m1b SF:277-294, directed_dis_m1b_redefine_diagnal SF:241-257), the per-read rule is
the drivers' (DEL: both scorers, the smaller score, SF:1718-1726; TANDUP: SF:1763-1765; INV / INS: SF:1910-1913, 1879-1882), and
the locus record comes from oracle.result_organize_ins (SF:1219-1231) and oracle.gt_estimate_log_likelihood (SF:2054-2069).
"""
import numpy as np

GT_INDEX = {"0/0": 0, "0/1": 1, "1/1": 2}


def read_score(oracle, kind, ref, alt, x, k, plots):
    """The drivers' per-read rule: a score, or None when the read is skipped (`0 in [a, b]`)."""
    def one(sc):
        return None if 0 in sc else 1 - float(sc[1]) / float(sc[0])
    if kind == 3:
        return one(oracle.score_directed_dis_m1b_redefine_diagnal(ref, alt, x, k, plots=plots))
    s1 = one(oracle.score_abs_dis_m1b(ref, alt, x, k, plots=plots))
    if kind != 0:
        return s1
    s2 = one(oracle.score_within_10Perc_m1b(ref, alt, x, k, plots=plots))
    if s1 is not None and s2 is not None:
        return min([s1, s2])
    return s1 if s1 is not None else s2


def expect(oracle, w, loci=None):
    """(stats (n_pairs, 16) int64, read_scores (n_reads,) float64 with NaN = skipped, loci (n_loci, 5) float64
    [QS, GS, GT index, GQ, reads scored] with a NaN row = 'NA') of workload `w`, by the oracle alone.  `loci`: only these
    (the other rows stay zero / NaN and `done` says which reads were looked at)."""
    n_reads = len(w.read_locus)
    stats = np.zeros((2 * n_reads, 16), dtype=np.int64)
    scores = np.full(n_reads, np.nan)
    done = np.zeros(n_reads, dtype=bool)
    want = set(range(w.n_loci)) if loci is None else set(int(t) for t in loci)
    per_locus = {li: [] for li in want}
    for r in range(n_reads):
        li = int(w.read_locus[r])
        if li not in want:
            continue
        pr, pa = w.pairs[2 * r], w.pairs[2 * r + 1]
        assert pr["seq1"] == pa["seq1"] and pr["off2"] == pa["off2"] and pr["k"] == pa["k"]
        read, ref, alt = w.seqs[pr["seq1"]], w.seqs[pr["seq2"]], w.seqs[pa["seq2"]]
        assert ref == ref.upper() and alt == alt.upper()      # (abs_dis_m1b upper-cases: the shared plots are valid for it)
        miss, k = int(pr["off2"]), int(pr["k"])
        st_r, R, _a, _b = oracle.pair_stats(k, read, ref[miss:], want_hits=True)
        st_a, A, _a, _b = oracle.pair_stats(k, read, alt[miss:], want_hits=True)
        stats[2 * r], stats[2 * r + 1] = st_r, st_a
        s = read_score(oracle, int(w.read_kind[r]), ref, alt, [read, miss, "r%d" % r], k, (R, A))
        done[r] = True
        if s is not None:
            scores[r] = s
            per_locus[li].append(s)
    rec = np.full((w.n_loci, 5), np.nan)
    for li in want:
        row = oracle.result_organize_ins(["k%d" % li, per_locus[li]])
        if row[1] == "NA":
            continue
        gt, gq = oracle.gt_estimate_log_likelihood(row)
        rec[li] = (row[1], row[2], GT_INDEX[gt], gq, len(per_locus[li]))
    return stats, scores, rec, done


def _share_main(argv):
    """Worker of expect_parallel (a fresh interpreter: `python tests/workload_oracle.py spec.json out.npz`): the oracle's
    answers for a share of the loci of a seeded workload."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import oracle as orc
    from vapor_amd import workload as wl
    job = json.load(open(argv[0]))
    orc.build()
    orc.lib()
    spec = dict(job["spec"])
    spec["svtypes"] = tuple(spec["svtypes"])
    w = wl.make_workload(job["name"], seed=job["seed"], **spec)
    st, sc, rec, done = expect(orc, w, job["loci"])
    sel = np.flatnonzero(done)
    np.savez(argv[1], loci=np.asarray(job["loci"]), sel=sel, st=st[np.sort(np.concatenate([2 * sel, 2 * sel + 1]))], sc=sc[sel],
             rec=rec[job["loci"]])


def expect_parallel(oracle, name, spec, seed, w, workers=None):
    """expect() over all loci of make_workload(name, seed, **spec) == w, shared out over `workers` fresh interpreters
    (child processes started from this file's path: they never see this process's GPU state; each rebuilds the seeded
    workload and runs the oracle on its share of the loci)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(8, workers - 1))
    if workers <= 1 or w.n_loci < 2 * workers:
        return expect(oracle, w)
    n_reads = len(w.read_locus)
    stats = np.zeros((2 * n_reads, 16), dtype=np.int64)
    scores = np.full(n_reads, np.nan)
    rec = np.full((w.n_loci, 5), np.nan)
    done = np.zeros(n_reads, dtype=bool)
    with tempfile.TemporaryDirectory(prefix="vapor_expect_") as tmp:
        procs = []
        for k in range(workers):
            job = os.path.join(tmp, "job%d.json" % k)
            json.dump({"name": name, "spec": spec, "seed": seed, "loci": list(range(k, w.n_loci, workers))}, open(job, "w"))
            out = os.path.join(tmp, "out%d.npz" % k)
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), job, out]), out))
        for p, out in procs:
            assert p.wait() == 0, "oracle worker failed"
            z = np.load(out)
            sel = z["sel"]
            stats[np.sort(np.concatenate([2 * sel, 2 * sel + 1]))] = z["st"]
            scores[sel] = z["sc"]
            rec[z["loci"]] = z["rec"]
            done[sel] = True
    assert done.all()
    return stats, scores, rec, done


def check_plan(w, got_stats, got_scores, got_loci, exp, loci=None, tag=""):
    """Device results against expect()'s: the integer statistics of every pair exactly, every per-read score and
    QS / GS within 1e-6 (north_star's tolerance; they are in fact equal), GT and the read counts exactly, GQ within 1e-6."""
    stats, scores, rec, done = exp
    sel = np.flatnonzero(done)
    psel = np.sort(np.concatenate([2 * sel, 2 * sel + 1]))
    assert (got_stats[psel, 15] == 0).all(), (tag, "status", got_stats[psel, 15][got_stats[psel, 15] != 0][:5])
    # the oracle's record holds statistics 0-9 (vapor_oracle.c); a plan computes what a pair's flags ask for: counts and spans
    # always, 3-4 with PF_C1, 5-6 and 9 with PF_C2.  The directed statistics (10-13, PF_DIR) are checked through the scores
    # of the reads they decide (directed_dis_m1b_redefine_diagnal on the oracle's own float64 restatement).
    fl = w.pairs["flags"][psel].astype(np.int64)
    mask = np.zeros((len(psel), 10), dtype=bool)
    mask[:, [0, 1, 2, 7, 8]] = True
    mask[:, 3:5] = (fl & 1)[:, None] > 0
    mask[:, [5, 6, 9]] = (fl & 2)[:, None] > 0
    bad = np.argwhere((got_stats[psel, :10] != stats[psel, :10]) & mask)
    assert len(bad) == 0, (tag, "statistics differ from the oracle's", bad[:6], got_stats[psel][bad[:3, 0]], stats[psel][bad[:3, 0]])
    if got_scores is not None:
        g, e = got_scores[sel], scores[sel]
        assert np.array_equal(np.isnan(g), np.isnan(e)), (tag, "reads skipped differ", np.flatnonzero(np.isnan(g) != np.isnan(e))[:6])
        ok = ~np.isnan(e)
        assert np.allclose(g[ok], e[ok], rtol=0, atol=1e-6), (tag, "read scores", np.abs(g[ok] - e[ok]).max())
    lsel = np.arange(w.n_loci) if loci is None else np.asarray(sorted(loci))
    g, e = got_loci[lsel], rec[lsel]
    na = np.isnan(e[:, 0])
    assert np.array_equal(np.isnan(g[:, 0]), na), (tag, "NA loci differ")
    g, e = g[~na], e[~na]
    assert np.array_equal(g[:, 2].astype(int), e[:, 2].astype(int)), (tag, "VaPoR_GT", np.flatnonzero(g[:, 2] != e[:, 2])[:6])
    assert np.array_equal(g[:, 4].astype(int), e[:, 4].astype(int)), (tag, "reads scored")
    assert np.allclose(g[:, :2], e[:, :2], rtol=0, atol=1e-6), (tag, "QS / GS", np.abs(g[:, :2] - e[:, :2]).max())
    assert np.allclose(g[:, 3], e[:, 3], rtol=0, atol=1e-6), (tag, "GQ", np.abs(g[:, 3] - e[:, 3]).max())
    return len(psel), len(lsel)


if __name__ == "__main__":
    import sys
    _share_main(sys.argv[1:])
