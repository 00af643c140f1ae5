"""Parity of the HIP path (through the C ABI) against reference-generated golden vectors and
against the CPU oracle on seeded inputs.  Bit-exact: this path is integer work throughout."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vapor_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _sha(h):
    return hashlib.sha256(np.ascontiguousarray(h, dtype=np.int32).tobytes()).hexdigest()


def test_dotdata_golden(eng):
    cases = load_golden("kmerhits.json.gz")["cases"]
    seqs, rows = [], []
    for c in cases:
        seqs += [c["seq1"], c["seq2"]]
        rows.append((len(seqs) - 2, len(seqs) - 1, 0, c["k"], 0))
    ss = eng.seqset(seqs)
    st, hits = eng.dotplots(ss, eng.make_pairs(rows))
    for t, c in enumerate(cases):
        if "error" in c:
            assert st[t, 15] == -3, c["name"]
            continue
        assert st[t, 15] == 0, c["name"]
        assert st[t, 0] == c["n_hits"], c["name"]
        assert _sha(hits[t]) == c["sha256"], c["name"]
        if "hits" in c:
            assert hits[t].tolist() == c["hits"], c["name"]


def test_cleaners_golden(eng, oracle):
    cases = load_golden("cleaners.json.gz")["cases"]
    lists = [np.asarray(c["hits"], dtype=np.int32).reshape(-1, 2) for c in cases]
    st, fl = eng.clean_hits(lists, flags=[7] * len(lists))
    from vapor_amd import finish
    for t, c in enumerate(cases):
        if "r4" in c and "ok" in c["r4"]:
            assert st[t, 10] == int(round(2 * float(c["r4"]["ok"]))), c["name"]
            assert float(finish._dir_value(st[t])) == float(c["dir"]["ok"]), c["name"]
        h = lists[t]
        got1 = h[(fl[t] & 1) > 0].tolist()
        assert got1 == c["c1"]["ok"], c["name"]
        assert sorted(map(tuple, h[(fl[t] & 2) > 0].tolist())) == sorted(map(tuple, c["c2_diag"]["ok"])), c["name"]
        assert sorted(map(tuple, h[(fl[t] & 4) > 0].tolist())) == \
            sorted(map(tuple, c.get("c2_anti_on_left", {"ok": []})["ok"])), c["name"]
        if "count10" in c:
            assert st[t, 6] == c["count10"], c["name"]
        if "meanabs" in c:
            assert float(st[t, 4]) / float(st[t, 3]) == c["meanabs"], c["name"]
        assert st[t, 0] == len(h) and st[t, 1] == h[:, 0].min() and st[t, 2] == h[:, 0].max()


def _check_stats_vs_oracle(eng, oracle, seqs, upper, rows, tag=""):
    ss = eng.seqset(seqs, upper)
    pairs = eng.make_pairs(rows)
    st = eng.score(ss, pairs)
    for t, (s1, s2, off2, k, fl) in enumerate(rows):
        a = seqs[s1].upper() if upper[s1] else seqs[s1]
        b = seqs[s2].upper() if upper[s2] else seqs[s2]
        try:
            exp = oracle.pair_stats(k, a, b[off2:])
        except KeyError:
            assert st[t, 15] == -3, (tag, t)
            continue
        assert st[t, 15] == 0, (tag, t)
        got = st[t].copy()
        if not fl & 1:
            exp[3] = exp[4] = 0
        if not fl & 2:
            exp[5] = exp[6] = exp[9] = 0
        assert got[:10].tolist() == exp[:10].tolist(), (tag, t, rows[t])
        if fl & 4 and exp[3] > 0:
            _st, h, k1, _k2 = oracle.pair_stats(k, a, b[off2:], want_hits=True)
            kept = [(int(x), int(y)) for x, y in h[k1 > 0]]
            c = oracle.dis_to_diagnal_most_abundant_defined(list(kept))
            far = [d for d in ([x + c, y] for x, y in kept) if oracle.eu_dis_single_dot(d) > 0.1]
            assert got[10] == int(round(2 * float(c))), (tag, t, "c", got[10], c)
            assert got[11] == len(far), (tag, t, "dir_n")
            assert got[12] == int(round(2 * sum(d[0] - d[1] for d in far))), (tag, t, "dir_sum")
    return st


def test_scorer_inputs_vs_oracle(eng, oracle):
    """Every (read, allele[miss:]) dot plot the three live scorers evaluate on the golden
    scorer inputs, with and without abs_dis_m1b's upper-casing."""
    cases = load_golden("scorers.json.gz")["cases"]
    seqs, upper, rows = [], [], []
    for c in cases:
        base = len(seqs)
        seqs += [c["read"], c["ref"], c["alt"], c["ref"], c["alt"]]
        upper += [False, False, False, True, True]
        for al in (1, 2, 3, 4):
            rows.append((base, base + al, c["miss"], c["k"], 7))
    _check_stats_vs_oracle(eng, oracle, seqs, upper, rows, "scorers")


def test_scorers_golden_end_to_end(eng):
    """Device statistics + host float64 finishing == the reference's three scorers, exactly."""
    from vapor_amd import finish
    cases = [c for c in load_golden("scorers.json.gz")["cases"]]
    seqs, upper, rows = [], [], []
    for c in cases:
        base = len(seqs)
        seqs += [c["read"], c["ref"], c["alt"], c["ref"], c["alt"]]
        upper += [False, False, False, True, True]
        for al in (1, 2, 3, 4):
            rows.append((base, base + al, c["miss"], c["k"], 7))
    ss = eng.seqset(seqs, upper)
    st = eng.score(ss, eng.make_pairs(rows))
    for t, c in enumerate(cases):
        r, a, ru, au = st[4 * t], st[4 * t + 1], st[4 * t + 2], st[4 * t + 3]
        lr, la = len(c["ref"]), len(c["alt"])
        if "error" in c["s1"]:
            assert r[15] == -3 and au[15] == -3
            continue
        got = {"s1": finish.score_abs_dis_m1b(ru, au, lr, la),
               "s2": finish.score_within_10Perc_m1b(r, a, lr, la),
               "s3": finish.score_directed_dis_m1b_redefine_diagnal(r, a, lr, la)}
        for key in ("s1", "s2", "s3"):
            assert [float(v) for v in got[key]] == [float(v) for v in c[key]["ok"]], (c["name"], key)
        kind = np.array([1, 2, 3])
        va, vb, valid = finish.batch_scores(kind, np.stack([ru, r, r]), np.stack([au, a, a]),
                                            np.array([lr] * 3), np.array([la] * 3))
        for q, key in enumerate(("s1", "s2", "s3")):
            assert [va[q], vb[q]] == [float(v) for v in c[key]["ok"]], (c["name"], key, "batch")
            assert bool(valid[q]) == (0 not in c[key]["ok"])


def test_selfplot_counts_golden(eng):
    cases = load_golden("window.json.gz")["cases"]
    seqs, rows, exp = [], [], []
    for c in cases:
        s = "".join(ch for ch in c["seq"] if ch != "X")
        for step, tr in enumerate(c["qc_trace"]):
            seqs.append(s)
            rows.append((len(seqs) - 1, len(seqs) - 1, 0, 10 + 10 * step, 0))
            exp.append(tr)
    ss = eng.seqset(seqs)
    st = eng.score(ss, eng.make_pairs(rows))
    assert st[:, [0, 7, 8]].tolist() == exp


@pytest.mark.parametrize("k", [10, 20, 30, 40])
def test_random_pairs_vs_oracle(eng, oracle, k):
    from vapor_amd import synth
    alleles, reads, pr = synth.make_pairs(1000 + k, 6, 4, 3000, 5000, errors=(0.002, 0.01, 0.005) if k > 10 else (0.01, 0.08, 0.04))
    seqs = alleles + reads
    rows = [(len(alleles) + r, a, (7 * r) % 50, k, 7) for r, a in pr]
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "rand%d" % k)


def test_general_mode_softmasked_both(eng, oracle):
    """Both sequences carry lower-case / N symbols -> 4-bit symbol path."""
    from vapor_amd import synth
    rng = np.random.default_rng(5)
    seqs, rows = [], []
    for t in range(6):
        a = synth.random_dna(rng, 2500)
        a = a[:400] + a[400:1100].lower() + a[1100:1500] + "N" * 30 + a[1530:2000] + "n" * 11 + a[2011:]
        r, _ = synth.mutate(rng, a[100:2300], 0.003, 0.01, 0.01)
        seqs += [a, r]
        for k in (10, 20, 30, 40):
            rows.append((len(seqs) - 1, len(seqs) - 2, 0, k, 7))
            rows.append((len(seqs) - 2, len(seqs) - 2, 0, k, 7))
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "general")


def test_bench_shape_and_multi_tile(eng, oracle):
    """BASELINE shapes: 10 kb x 20 kb, 15 kb x 20 kb, and 30 kb x 40 kb (two hash-table tiles)."""
    from vapor_amd import synth
    seqs, rows = [], []
    for seed, (lr, la) in enumerate(((10000, 20000), (15000, 20000), (30000, 40000))):
        alleles, reads, pr = synth.make_pairs(77 + seed, 2, 3, lr, la)
        base = len(seqs)
        seqs += alleles + reads
        rows += [(base + len(alleles) + r, base + a, 0, 10, 7) for r, a in pr]
    st = _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "bench")
    assert st[:, 0].min() > 1000


def test_overflow_retry_and_hit_fetch(eng, oracle):
    """Low-complexity input: far more hits than the first-guess slot -> exact-size rerun."""
    seqs = ["A" * 700, "A" * 900 + "C" * 50, "AT" * 400, "TA" * 450]
    rows = [(0, 1, 0, 10, 3), (2, 3, 5, 10, 3), (0, 3, 0, 10, 3)]
    ss = eng.seqset(seqs)
    st, hits = eng.dotplots(ss, eng.make_pairs(rows))
    for t, (s1, s2, off2, k, _f) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:])
        assert st[t, 0] == len(exp)
        assert hits[t].tolist() == exp.tolist()
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * 4, rows, "overflow")


@pytest.mark.parametrize("rpt,jt", [(1, 512), (3, 512), (64, 1), (64, 2), (64, 5), (7, 3)])
def test_task_partition_invariance(eng, oracle, rpt, jt):
    """Results do not depend on how the sorted pair list is cut into join tasks (one task may
    span several alleles and rebuild its table; one allele may be split over several tasks)."""
    from vapor_amd import synth
    alleles, reads, pr = synth.make_pairs(31, 5, 9, 1500, 2500)
    seqs = alleles + reads
    rows = [(len(alleles) + r, a, 0, 10, 7) for r, a in pr]
    eng.set_param("reads_per_task", rpt)
    eng.set_param("join_tasks", jt)
    try:
        _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "rpt%d_jt%d" % (rpt, jt))
    finally:
        eng.set_param("reads_per_task", 64)
        eng.set_param("join_tasks", 256)


def test_bad_arguments(eng):
    ss = eng.seqset(["ACGTACGTACGTACGT", "ACGT"])
    st = eng.score(ss, eng.make_pairs([(0, 5, 0, 10, 3), (0, 1, 0, 11, 3), (0, 1, 0, 10, 3), (1, 0, 0, 10, 3)]))
    assert st[0, 15] == -4 and st[1, 15] == -4
    assert st[2, 15] == 0 and st[2, 0] == 0 and st[2, 1] == -1
    assert st[3, 15] == 0 and st[3, 0] == 0


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_device_finish_matches_host_finish(eng, name):
    """finish_kernel (per-read scores, QS/GS/GT/GQ on the device) == vapor_amd.finish on the host."""
    from vapor_amd import workload as wl
    spec = dict(wl.WORKLOADS["tiny"]) if name == "tiny" else dict(n_loci=24, svtypes=("DEL", "TANDUP", "INV", "INS"),
                                                                read_len=3000, allele_len=3400, reads_per_locus=17)
    w = wl.make_workload(name, seed=5, **spec)
    ss = eng.seqset(w.seqs)
    plan = eng.plan(ss, w.pairs)
    st = plan.run().copy()
    host = wl.finish_workload(w, st)
    plan.set_reads(wl.read_table(w), w.n_loci)
    for _ in range(2):          # first call takes the full path, the second the statistics-stay-on-device path
        dev = plan.run_loci(want_scores=True).copy()
        assert np.array_equal(np.isnan(dev[:, 0]), np.isnan(host[:, 0]))
        ok = ~np.isnan(host[:, 0])
        assert ok.sum() >= 2
        assert np.array_equal(dev[ok, 1], host[ok, 1])            # GS
        assert np.array_equal(dev[ok, 2], host[ok, 2])            # GT
        assert np.array_equal(dev[ok, 3], host[ok, 3])            # GQ (same table)
        assert np.array_equal(dev[ok, 4], host[ok, 4])            # reads scored
        assert np.allclose(dev[ok, 0], host[ok, 0], rtol=0, atol=1e-12)
        # QS in numpy's own summation order: identical to np.mean over the positive scores
        sc = plan.read_scores[:len(w.read_locus)]
        for li in np.flatnonzero(ok):
            v = sc[w.read_locus == li]
            v = v[~np.isnan(v)]
            pos = [float(x) for x in v if x > 0]
            assert dev[li, 0] == (np.mean(pos) if pos else 0.0)
    plan.close()


def test_exception_paths_at_bench_shapes(eng, oracle):
    """Reads with N runs against a clean 20 kb allele (2-bit path, read k-mers masked through the
    exception plane) and against a soft-masked / N-holding allele (4-bit path, two 16 k tiles)."""
    from vapor_amd import synth
    rng = np.random.default_rng(21)
    a = synth.random_dna(rng, 20000)
    am = a[:3000] + a[3000:9000].lower() + a[9000:12000] + "N" * 60 + a[12060:17000] + a[17000:].lower()
    r, _ = synth.mutate(rng, a[2000:13000], 0.004, 0.02, 0.01)
    r = r[:10000]
    rn = r[:2000] + "N" * 25 + r[2025:6000] + "n" * 9 + r[6009:]
    rl = r[:4000] + r[4000:7000].lower() + r[7000:]
    seqs = [a, am, r, rn, rl]
    rows = []
    for rd in (2, 3, 4):
        for al in (0, 1):
            for k in (10, 20):
                rows.append((rd, al, 3 * rd, k, 7))
    rows += [(1, 1, 0, 10, 7), (1, 1, 0, 30, 7)]
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "exc_bench")


def test_run_records_expand_to_the_oracle_dots(eng, oracle):
    """The device stores runs of consecutive dots as one record.  Accurate reads (runs hit the 32-dot cap and
    the strip / tile boundaries), an inverted segment (reverse-complement dots), an allele slice (miss_bp), a
    two-tile allele and an N in the read (run forming is switched off for that pair): the expanded dot lists
    must be exactly dotdata()'s, and the flags / statistics those records produce the oracle's."""
    from vapor_amd import synth
    rng = np.random.default_rng(31)
    allele = synth.random_dna(rng, 6000)
    exact = allele[700:5200]                                            # one 4 500-dot diagonal
    nearly = synth.mutate(np.random.default_rng(5), allele[300:5600], 0.001, 0.002, 0.002)[0]
    inv = allele[500:2000] + synth.revcomp(allele[2000:3500]) + allele[3500:5000]
    with_n = exact[:2000] + "N" + exact[2001:]
    big = synth.random_dna(rng, 36000)                                  # more than one table holds: two tiles
    long_read = big[30000:35500]
    huge = synth.random_dna(rng, 47000)                                 # a longer one, read across the tile edge below
    huge_read = synth.mutate(np.random.default_rng(8), huge[22000:27500], 0.002, 0.004, 0.004)[0]   # across the tile edge
    seqs = [allele, exact, nearly, inv, with_n, big, long_read, huge, huge_read]
    rows = [(1, 0, 0, 10, 7), (2, 0, 0, 10, 7), (3, 0, 0, 10, 7), (1, 0, 913, 10, 3), (4, 0, 0, 10, 7),
            (6, 5, 0, 10, 7), (1, 0, 0, 20, 7), (2, 0, 0, 40, 3), (6, 5, 31000, 30, 1), (8, 7, 0, 10, 7), (8, 7, 21000, 20, 3)]
    ss = eng.seqset(seqs)
    plan = eng.plan(ss, eng.make_pairs(rows))
    st = plan.run().copy()
    rec = plan.record_counts()
    hits, _fl, off = plan.fetch_hits(range(len(rows)), want_flags=True)
    for t, (s1, s2, off2, k, _f) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:])
        got = hits[off[t]:off[t + 1]]
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert st[t, 0] == len(exp) and got.tolist() == exp.tolist(), (t, rows[t])
        assert 0 < rec[t] <= st[t, 0]
    assert rec[0] * 20 < st[0, 0]                 # an exact copy: 32 dots per record
    assert rec[4] * 8 < st[4, 0]                  # a read with an N forms runs as well (they end before it)
    plan.close()
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "runs")


def test_sequences_at_the_length_limit(eng, oracle):
    """VAPOR_MAX_SEQ_LEN = 65 535 bases on both sides (three allele tiles, strips up to the 16-bit position limit, the
    clean kernels' value range at its 4 096-word cap): statistics exact for every window size and scorer; one base more
    is refused per pair with VAPOR_E_ARG, the other pairs of the batch are unaffected."""
    from vapor_amd import synth
    from vapor_amd import _lib as L
    rng = np.random.default_rng(65535)
    n = L.MAX_SEQ_LEN
    allele = synth.random_dna(rng, n)
    read = (synth.mutate(np.random.default_rng(3), allele[2000:60000], 0.01, 0.03, 0.03)[0] + synth.random_dna(rng, n))[:n]
    exact_tail = allele[n - 9000:]                       # a read that ends on the allele's last base
    seqs = [allele, read, exact_tail]
    rows = [(1, 0, 0, 10, 7), (1, 0, 0, 20, 3), (1, 0, 30000, 30, 1), (1, 0, 0, 40, 3), (2, 0, 0, 10, 7), (2, 0, 50000, 10, 3)]
    assert len(read) == n and len(allele) == n
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "limit")
    too_long = allele + "A"
    seqs2 = [allele, exact_tail, too_long]
    ss = eng.seqset(seqs2)
    st = eng.score(ss, eng.make_pairs([(1, 0, 0, 10, 7), (1, 2, 0, 10, 7), (2, 0, 0, 10, 7)]))
    assert st[0, 15] == 0 and st[0, 0] > 8000
    assert st[1, 15] == L.E_ARG and st[2, 15] == L.E_ARG
    ss.close()


def test_async_steps_equal_blocking_run(eng):
    """vapor_plan_run_loci_async / vapor_plan_sync: the enqueue-only steps give the records of the blocking run,
    more steps than event sets are fine, and the timings are per-step averages."""
    from vapor_amd import workload as wl
    w = wl.make_workload("tiny", seed=5, **wl.WORKLOADS["tiny"])
    ss = eng.seqset(w.seqs)
    plan = eng.plan(ss, w.pairs)
    plan.set_reads(wl.read_table(w), w.n_loci)
    with pytest.raises(Exception):
        plan.run_loci_async()                      # needs one blocking run first
    ref = plan.run_loci().copy()
    for _ in range(70):
        plan.run_loci_async()
    got = plan.sync().copy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])
    tm = plan.timings()
    assert 0 < tm["join_ms"] < 50 and 0 < tm["clean_ms"] < 50 and tm["total_ms"] >= tm["join_ms"]
    plan.close()


def test_runs_with_a_soft_masked_allele(eng, oracle):
    """Alleles with symbols outside upper-case ACGT (lower-case stretches as in a soft-masked reference, N, IUPAC) and reads
    without: the join still forms runs - they end before the first such symbol and never start behind one - so the
    records stay a fraction of the dots, and the dots and every statistic are the oracle's, for every window size."""
    from vapor_amd import synth
    rng = np.random.default_rng(57)
    allele = synth.random_dna(rng, 9000)
    b = bytearray(allele.encode())
    for a, n in ((300, 40), (1500, 700), (2300, 1), (2301, 1), (4000, 25), (6990, 30), (8960, 40)):
        b[a:a + n] = bytes(b[a:a + n]).lower()
    for a, ch in ((3000, "N"), (3050, "R"), (5000, "N"), (5001, "N"), (5031, "n"), (7000, "Y")):
        b[a] = ord(ch)
    masked = b.decode()
    exact = allele[100:8900]                            # every k-mer of the read lies on the allele's diagonal
    noisy = synth.mutate(np.random.default_rng(9), allele[200:8800], 0.002, 0.004, 0.004)[0]
    inv = allele[500:3000] + synth.revcomp(allele[3000:5500]) + allele[5500:8000]
    seqs = [masked, exact, noisy, inv, allele]
    rows = [(1, 0, 0, 10, 7), (2, 0, 0, 10, 7), (3, 0, 0, 10, 7), (1, 0, 0, 20, 3), (2, 0, 1234, 30, 1), (1, 0, 0, 40, 3),
            (1, 4, 0, 10, 7)]
    ss = eng.seqset(seqs)
    plan = eng.plan(ss, eng.make_pairs(rows))
    st = plan.run().copy()
    rec = plan.record_counts()
    hits, _fl, off = plan.fetch_hits(range(len(rows)), want_flags=True)
    for t, (s1, s2, off2, k, _f) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:])
        got = hits[off[t]:off[t + 1]]
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert st[t, 0] == len(exp) and got.tolist() == exp.tolist(), (t, rows[t])
    assert st[0, 0] > 6000 and rec[0] * 8 < st[0, 0]      # runs despite the exceptions: a few hundred records for thousands of dots
    assert st[0, 0] < st[6, 0]                              # the masked stretches carry no dots
    plan.close()
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "soft-masked")


def test_runs_with_exception_symbols_in_the_read(eng, oracle):
    """Reads with symbols outside upper-case ACGT (an N, a lower-case stretch) against alleles without (these
    pairs stored every dot as its own record): a launch of their own (join_kernel<.., EXC = 2>) masks the positions whose
    k-mer covers such a symbol out of the lookup, ends a run before the first of them and starts none behind one.  Dots and
    statistics are the oracle's for every window size; the records stay a fraction of the dots."""
    from vapor_amd import synth
    rng = np.random.default_rng(61)
    allele = synth.random_dna(rng, 9000)

    def spoiled(seg, spots, lower=()):
        b = bytearray(seg.encode())
        for a in spots:
            b[a] = ord("N")
        for a, n in lower:
            b[a:a + n] = bytes(b[a:a + n]).lower()
        return b.decode()

    exact = spoiled(allele[100:8900], (0, 9, 10, 31, 32, 33, 500, 1013, 1014, 1023, 1024, 1025, 1055, 2047, 2048, 4000, 4001, 8799),
                    lower=((3000, 45), (6000, 1)))
    noisy = spoiled(synth.mutate(np.random.default_rng(10), allele[200:8800], 0.002, 0.004, 0.004)[0], (77, 1500, 1501, 5000, 8000))
    inv = spoiled(allele[500:3000] + synth.revcomp(allele[3000:5500]) + allele[5500:8000], (1200, 2600, 3100, 4700, 6000))
    clean_n = sum(1 for c in exact if c not in "ACGT")
    seqs = [allele, exact, noisy, inv]
    rows = [(1, 0, 0, 10, 7), (2, 0, 0, 10, 7), (3, 0, 0, 10, 7), (1, 0, 0, 20, 3), (2, 0, 1234, 30, 1), (1, 0, 0, 40, 3), (1, 0, 150, 10, 7)]
    ss = eng.seqset(seqs)
    assert ss.n_exc[0] == 0 and ss.n_exc[1] == clean_n and ss.n_exc[2] > 0 and ss.n_exc[3] > 0
    plan = eng.plan(ss, eng.make_pairs(rows))
    st = plan.run().copy()
    rec = plan.record_counts()
    hits, _fl, off = plan.fetch_hits(range(len(rows)), want_flags=True)
    for t, (s1, s2, off2, k, _f) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:])
        got = hits[off[t]:off[t + 1]]
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert st[t, 0] == len(exp) and got.tolist() == exp.tolist(), (t, rows[t])
    # runs despite the exceptions: the exact copy's diagonal is a few hundred records (32 dots each at most, cut at every
    # N), not thousands of single dots
    assert st[0, 0] > 7000 and rec[0] * 8 < st[0, 0], (st[0, 0], rec[0])
    assert rec[1] * 2 < st[1, 0] and rec[3] * 8 < st[3, 0] and rec[5] * 8 < st[5, 0]
    plan.close()
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "read exceptions")


def test_runs_on_the_four_bit_planes(eng, oracle):
    """Both sides hold symbols outside upper-case ACGT (the self plot of a soft-masked window, a lower-case read against
    a lower-case allele): the 4-bit planes, where lower case matches lower case and N matches N.  Runs are formed there
    too (up to 32 dots for window sizes 10 and 20, up to 16 for 30 and 40); the expanded dots are dotdata()'s, the statistics the
    oracle's."""
    from vapor_amd import synth
    rng = np.random.default_rng(58)
    base = synth.random_dna(rng, 7000)
    b = bytearray(base.encode())
    for a, n in ((200, 900), (2500, 40), (4100, 1500), (6950, 50)):
        b[a:a + n] = bytes(b[a:a + n]).lower()
    for a in (1500, 1501, 3000, 5000):
        b[a] = ord("N")
    masked = b.decode()
    shifted = masked[300:6500]                         # a read that carries the same lower-case stretches and Ns
    other = synth.mutate(np.random.default_rng(4), base[100:6000], 0.003, 0.004, 0.004)[0]
    other = other[:1000] + other[1000:2000].lower() + other[2000:]
    seqs = [masked, shifted, other]
    rows = [(0, 0, 0, 10, 0), (0, 0, 0, 20, 0), (1, 0, 0, 10, 7), (2, 0, 0, 10, 7), (1, 0, 777, 20, 3), (0, 0, 0, 30, 0),
            (0, 0, 0, 40, 0), (1, 0, 333, 30, 7), (2, 0, 0, 40, 3)]
    ss = eng.seqset(seqs)
    assert ss.n_exc[0] > 0 and ss.n_exc[1] > 0 and ss.n_exc[2] > 0
    plan = eng.plan(ss, eng.make_pairs(rows))
    st = plan.run().copy()
    rec = plan.record_counts()
    hits, _fl, off = plan.fetch_hits(range(len(rows)), want_flags=True)
    for t, (s1, s2, off2, k, _f) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:])
        got = hits[off[t]:off[t + 1]]
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        assert st[t, 0] == len(exp) and got.tolist() == exp.tolist(), (t, rows[t])
    assert st[0, 0] >= 6900 and rec[0] * 10 < st[0, 0]      # the self plot's diagonal in runs of up to 32
    assert rec[1] * 10 < st[1, 0]
    # window sizes 30 and 40 (they stored every dot as its own record): runs of up to 16 dots
    assert rec[5] * 10 < st[5, 0] and rec[6] * 10 < st[6, 0] and rec[7] * 8 < st[7, 0], (rec[5:8], st[5:8, 0])
    plan.close()
    _check_stats_vs_oracle(eng, oracle, seqs, [False] * len(seqs), rows, "x4 runs")


def test_async_steps_on_a_stream_of_the_callers(eng):
    """vapor_set_stream: everything - the asynchronous steps and their finish kernel too - runs on the caller's stream,
    in order with what the caller enqueues there; back on the library's own streams afterwards."""
    import torch
    from vapor_amd import workload as wl
    w = wl.make_workload("tiny", seed=13, **wl.WORKLOADS["tiny"])
    ss = eng.seqset(w.seqs)
    plan = eng.plan(ss, w.pairs)
    plan.set_reads(wl.read_table(w), w.n_loci)
    ref = plan.run_loci().copy()
    buf = torch.full((w.n_loci, 8), -1.0, dtype=torch.float64, device="cuda")   # (first: this is what initialises torch's device)
    mine = torch.cuda.Stream()
    eng.set_stream(mine.cuda_stream)
    try:
        copies = []
        for _ in range(4):
            plan.run_loci_async(device_out=buf.data_ptr())
            with torch.cuda.stream(mine):
                copies.append(buf.clone())               # same stream: ordered after the step without any event
                buf.fill_(-3.0)
        got = plan.sync().copy()
    finally:
        eng.set_stream(0)
    torch.cuda.synchronize()
    for c in copies + [torch.from_numpy(got)]:
        c = c.cpu().numpy()
        assert np.array_equal(np.isnan(c), np.isnan(ref)) and np.array_equal(c[~np.isnan(c)], ref[~np.isnan(ref)])
    for _ in range(3):
        plan.run_loci_async()
    got = plan.sync().copy()
    assert np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])
    plan.close()
    ss.close()


def test_async_steps_with_a_pair_left_to_the_big_kernel(eng):
    """An asynchronous step leaves clean_big_kernel out only when the plan's blocking run saw no pair that needs it:
    with a pair of more than 65 535 dots (tandem repeats on both sides) in the batch the steps still run it, and the
    records equal the blocking run's - which in turn equal what the statistics of a plain run give on the host."""
    from vapor_amd import workload as wl
    w = wl.make_workload("tiny", seed=9, **wl.WORKLOADS["tiny"])
    unit = "ACGTTGCAAGGCTTAACCGGATCGATTACGGCATCGTAGCTAGGCTAACGT"
    r = 0                                            # read 0 of locus 0 and its ref window become tandem repeats
    ref_idx, read_idx = int(w.pairs["seq2"][2 * r]), int(w.pairs["seq1"][2 * r])
    w.seqs[ref_idx] = unit * 60
    w.seqs[read_idx] = unit * 40
    w.len_ref[w.read_locus == w.read_locus[r]] = len(w.seqs[ref_idx])
    ss = eng.seqset(w.seqs)
    plan = eng.plan(ss, w.pairs)
    st = plan.run().copy()
    assert st[2 * r, 0] > 65535 and st[2 * r, 15] == 0    # dots of the repeat pair: the 16-bit counters of clean_kernel cannot hold them
    want = wl.finish_workload(w, st)
    plan.set_reads(wl.read_table(w), w.n_loci)
    ref = plan.run_loci().copy()
    for _ in range(5):
        plan.run_loci_async()
    got = plan.sync().copy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])
    ok = ~np.isnan(want[:, 0])
    assert np.array_equal(ok, ~np.isnan(ref[:, 0])) and np.array_equal(ref[ok, :3], want[ok, :3])
    plan.close()
    ss.close()


def test_randomised_sweep_vs_oracle(eng):
    """A few seconds of tools/fuzz_parity.py (odd lengths, every k, slices, soft-masked / N / repeat sequences,
    exact copies, alleles around the tile size): dots, statistics and directed statistics exact."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    msg = mod.run(6.0, 12345, engine=eng)
    assert msg.startswith("fuzz ok")


def test_async_steps_report_overflow(eng):
    """A pair that cannot get a big enough record slot (max_pair_cap) makes vapor_plan_sync fail loudly instead of
    handing back records computed from a truncated dot list."""
    from vapor_amd import _lib as L
    from vapor_amd import workload as wl
    w = wl.make_workload("tiny", seed=6, **wl.WORKLOADS["tiny"])
    eng.set_param("max_pair_cap", 16)
    try:
        ss = eng.seqset(w.seqs)
        plan = eng.plan(ss, w.pairs)
        plan.set_reads(wl.read_table(w), w.n_loci)
        st = plan.run()
        assert (st[:, 15] == -2).any()                 # slots capped at 16 records: overflow stays
        plan.run_loci()
        plan.run_loci_async()
        with pytest.raises(L.VaporHipError):
            plan.sync()
        plan.close()
    finally:
        eng.set_param("max_pair_cap", 1 << 28)


def test_queue_edge(eng, oracle):
    """Candidate totals of 127, 129 and 129 on consecutive filled positions of one strip (tests/queue_case.py): the
    queue's fast path must not take a total it cannot always hold.  Dot for dot against the oracle."""
    import queue_case
    read, allele, _km = queue_case.build()
    assert queue_case.candidate_totals(read, allele)[:9] == [127, 0, 0, 0, 129, 0, 0, 0, 129]
    seqs = [read, allele]
    ss = eng.seqset(seqs)
    st, hits = eng.dotplots(ss, eng.make_pairs([(0, 1, 0, 10, 7)]))
    exp, h, _k1, _k2 = oracle.pair_stats(10, read, allele, want_hits=True)
    assert st[0, 15] == 0 and st[0, :10].tolist() == exp[:10].tolist()
    eh = np.asarray(h, dtype=np.int32).reshape(-1, 2)
    assert hits[0].tolist() == eh[np.lexsort((eh[:, 1], eh[:, 0]))].tolist()
    assert st[0, 0] >= 127 + 129 + 129


def test_deep_loci_device_finish_vs_reference(eng):
    """BASELINE configs[4]'s concordance leg: loci of 33-64 reads through join -> clean -> finish_kernel; per-read
    scores, VaPoR_QS, VaPoR_GS, VaPoR_GT and VaPoR_GQ against what the reference's scorers, result_organize_ins
    (SF:1219-1231) and gt_estimate_log_likelihood (SF:2054-2077) returned for the same reads."""
    import deep_cases as dc
    seqs, rows, table, n_loci = dc.build()
    ss = eng.seqset(seqs)
    plan = eng.plan(ss, eng.make_pairs(rows))
    plan.set_reads(table, n_loci)
    for _ in range(2):
        loci = plan.run_loci(want_scores=True).copy()
        sc = plan.read_scores[:len(table)].copy()
        for li, c in enumerate(dc.DEEP):
            scores, qs, gs, gt, gq = dc.expected(c)
            got = sc[table["locus"] == li]
            got = got[~np.isnan(got)]
            assert got.tolist() == scores, c["name"]                   # float for float (tolerance 1e-6 not needed)
            assert loci[li, 0] == qs and loci[li, 1] == gs, c["name"]  # np.mean's own summation order
            assert int(loci[li, 2]) == gt and loci[li, 3] == gq, c["name"]
            assert int(loci[li, 4]) == len(scores)
    plan.close()
    ss.close()


def test_two_plans_in_flight_and_stream_ordering(eng):
    """vapor_plan_run_loci_async on two plans (one library stream each), with a stream of the caller's ordered against
    them by vapor_plan_then / vapor_plan_after: the records every pass leaves in the caller's buffers equal the
    blocking run's, and vapor_plan_sync reports per-kernel times."""
    import torch
    from vapor_amd import workload as wl
    ws = [wl.make_workload("tiny", seed=s, **wl.WORKLOADS["tiny"]) for s in (21, 22)]
    sets = [eng.seqset(w.seqs) for w in ws]
    plans, want, bufs = [], [], []
    for w, ss in zip(ws, sets):
        p = eng.plan(ss, w.pairs)
        p.set_reads(wl.read_table(w), w.n_loci)
        want.append(p.run_loci().copy())
        plans.append(p)
        bufs.append(torch.full((w.n_loci, 8), -1.0, dtype=torch.float64, device="cuda"))
    mine = torch.cuda.Stream()
    copies = []
    for i in range(8):
        p, b = plans[i % 2], bufs[i % 2]
        p.run_loci_async(device_out=b.data_ptr())
        p.then(mine.cuda_stream)                     # my stream reads the records after this pass ...
        with torch.cuda.stream(mine):
            copies.append((i % 2, b.clone()))
            b.fill_(-2.0)                            # ... and scribbles over the buffer
        p.after(mine.cuda_stream)                    # the plan's next pass overwrites it only after that
    for p in plans:
        p.sync()
    torch.cuda.synchronize()
    for k, c in copies:
        got = c.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want[k])) and np.array_equal(got[~np.isnan(got)], want[k][~np.isnan(want[k])])
    tm = plans[0].timings()
    assert tm["join_ms"] > 0 and tm["clean_ms"] > 0
    for p in plans:
        p.close()
    for ss in sets:
        ss.close()


def test_join_tasks_in_whole_rounds_when_the_reads_per_task_limit_binds(eng, oracle):
    """Round 5: when the limit of reads per task asks for more join tasks than `join_tasks` (= CUs), the partition aims for whole
    rounds of them (BASELINE configs[2]: 625 -> 768 tasks).  The same rule at toy size - 3 "CUs", 5 reads per task, 60 reads of 6
    windows - gives the statistics of the default partition and the oracle's dots."""
    from vapor_amd import _lib as L
    from vapor_amd import synth
    rng = np.random.default_rng(66)
    wins = [synth.random_dna(rng, 1800 + 100 * t) for t in range(6)]
    reads, rows = [], []
    for wi, wn in enumerate(wins):
        for t in range(10):
            a = int(rng.integers(0, 600))
            reads.append(synth.mutate(rng, wn[a:a + 1100], 0.01, 0.05, 0.03)[0])
            rows.append((len(wins) + len(reads) - 1, wi, (0, 11)[t % 2], (10, 20)[t % 3 == 0], L.PF_C1 | L.PF_C2 | L.PF_DIR))
    seqs = wins + reads
    pairs = eng.make_pairs(rows)
    ss = eng.seqset(seqs)
    want = eng.score(ss, pairs)
    eng.set_param("join_tasks", 3)
    eng.set_param("reads_per_task", 5)
    try:
        st, dots = eng.dotplots(ss, pairs)
    finally:
        eng.set_param("join_tasks", 256)
        eng.set_param("reads_per_task", 64)
    ss.close()
    assert np.array_equal(st, want)
    for t in range(0, len(rows), 7):
        s1, s2, off2, k, _f = rows[t]
        assert np.array_equal(dots[t], oracle.dotdata_array(k, seqs[s1], seqs[s2][off2:]).reshape(-1, 2)), t
