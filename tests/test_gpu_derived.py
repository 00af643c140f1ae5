"""Derived sequences (vapor_seqset_create_derived) and the shared joins a plan builds on them, on the GPU.

(1) The planes the device assembles from segment descriptors equal the planes of the same text uploaded as bytes - for every
    way the reference's drivers build an allele from a window it has read: a deletion's ref[:f] + ref[-f:] (SF:1712), a tandem
    duplication's ref[:f] + mid + mid + ref[-f:] (SF:1755), an inversion's ref[:f] + reverse(complementary(mid)) + ref[-f:]
    (SF:1907), an insertion's flank + ins_seq + flank (SF:1872), the block structures of DEL_INV / DUP_INV / DISDUP
    (SF:1557-1665: 'aba^', 'b^ab', 'aa^', 'bab'), the str.upper() twins of abs_dis_m1b (SF:183-184).
(2) A plan over such a set joins every read ONCE against a window and the alleles derived from it (remap_kernel): dots and
    statistics of every pair equal the oracle's dotdata(read, allele[miss:]) and those of the same plan with one join per pair.

The bodies take an engine, so tests/test_cpu_twin.py runs them on the CPU twin as well."""
import numpy as np
import pytest

from vapor_amd import _lib as L
from vapor_amd import seqio, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vapor_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _rc(s):
    return seqio.reverse(seqio.complementary(s))


def allele_cases(rng):
    """[(name, literals, [(segments, upper, expected text)])]: every derived sequence with the text the reference would build."""
    out = []
    g = synth.random_dna(rng, 2400)
    f = 500

    def soft(s, a, b):
        return s[:a] + s[a:b].lower() + s[b:]
    for tag, ref in (("plain", g), ("softmasked", soft(soft(g, 300, 420), 1500, 1530)), ("with_N", g[:800] + "N" * 7 + g[807:1900] + "n" * 3 + g[1903:])):
        n = len(ref)
        mid = ref[f:n - f]
        der = [
            ([(0, 0, f, 0), (0, n - f, f, 0)], False, ref[:f] + ref[-f:]),                                  # DEL
            ([(0, 0, f, 0), (0, f, n - 2 * f, 0), (0, f, n - 2 * f, 0), (0, n - f, f, 0)], False, ref[:f] + mid + mid + ref[-f:]),   # TANDUP
            ([(0, 0, n - f, 0), (0, f, n - 2 * f, 0), (0, n - f, f, 0)], False, ref[:n - f] + mid + ref[-f:]),  # the same, first two merged
            ([(0, 0, f, 0), (0, f, n - 2 * f, 1), (0, n - f, f, 0)], False, ref[:f] + _rc(mid) + ref[-f:]),  # INV
            ([(0, 0, f, 0), (1, 0, 333, 0), (0, f, n - f, 0)], False, ref[:f] + "%INS%" + ref[f:]),         # INS (payload = literal 1)
            ([(0, 0, n, 0)], True, ref.upper()),                                                             # upper twin of the window
            ([(0, 0, f, 0), (0, n - f, f, 0)], True, (ref[:f] + ref[-f:]).upper()),                           # upper twin of the DEL allele
        ]
        a, b = ref[f:f + 400], ref[f + 400:f + 900]
        pre, post = ref[:f], ref[f + 900:f + 900 + f]
        A, B = (0, f, 400), (0, f + 400, 500)
        P, Q = (0, 0, f, 0), (0, f + 900, f, 0)
        der += [
            ([P, A + (0,), B + (0,), A + (1,), Q], False, pre + a + b + _rc(a) + post),                     # DUP_INV 'aba^'
            ([P, B + (1,), A + (0,), B + (0,), Q], False, pre + _rc(b) + a + b + post),                     # DUP_INV 'b^ab'
            ([P, A + (0,), A + (1,), Q], False, pre + a + _rc(a) + post),                                   # DUP_INV 'aa^'
            ([P, B + (0,), A + (0,), B + (0,), Q], False, pre + b + a + b + post),                          # DISDUP 'bab'
            ([P, B + (1,), Q], False, pre + _rc(b) + post),                                                  # DEL_INV: a deleted, b inverted
            ([(0, 0, f, 0), (0, f + 12, n - f - 12, 0)], False, ref[:f] + ref[f + 12:]),                      # a 12 bp deletion (shorter than a run)
            ([(0, 0, f, 0), (0, f, 7, 0), (0, f, 7, 0), (0, f + 7, n - f - 7, 0)], False, ref[:f] + ref[f:f + 7] * 2 + ref[f + 7:]),   # 7 bp duplication (< k)
            ([(1, 0, 333, 0)], False, "%INS%"),                                                              # nothing of the window at all
            ([(0, 5, 9, 0)], False, ref[5:14]),                                                              # shorter than any k
        ]
        ins = synth.random_dna(rng, 333)
        if tag == "with_N":
            ins = ins[:100] + "NN" + ins[102:]
        out.append((tag, [ref, ins], [(sg, up, txt.replace("%INS%", ins.upper() if up else ins)) for sg, up, txt in der]))
    return out


def check_planes(eng):
    rng = np.random.default_rng(808)
    for tag, lits, der in allele_cases(rng):
        ss = eng.seqset(lits, derived=[([(p, o, n, bool(rc)) for p, o, n, rc in sg], up) for sg, up, _t in der])
        ref = eng.seqset(lits + [t for _sg, _u, t in der])
        try:
            assert ss.n == ref.n == len(lits) + len(der)
            assert ss.lens.tolist() == ref.lens.tolist(), tag
            assert ss.n_exc.tolist() == ref.n_exc.tolist() and ss.n_invalid.tolist() == ref.n_invalid.tolist(), tag
            for t in range(ss.n):
                for a, b, what in zip(ss.planes(t), ref.planes(t), ("p2", "e1", "x4")):
                    assert np.array_equal(a, b), (tag, t, what)
        finally:
            ss.close()
            ref.close()


def test_derived_planes_equal_the_planes_of_the_same_text(eng):
    check_planes(eng)


def test_revcomp_of_a_window_with_iupac_codes_is_refused(eng):
    """complementary() drops what is not ATGCN / atgcn (SF:471-478): a descriptor cannot say that, so the library refuses a
    reverse-complemented slice of such a window (the caller uploads the allele as bytes) - and takes forward slices of it."""
    from vapor_amd._lib import VaporHipError
    w = "ACGTACGTRACGTTTGACCAGGTTAACCAGT" * 3
    with pytest.raises(VaporHipError) as ei:
        eng.seqset([w], derived=[([(0, 3, 40, True)], False)])
    assert ei.value.code == L.E_ARG
    ss = eng.seqset([w], derived=[([(0, 3, 40, False)], False)])
    ref = eng.seqset([w[3:43]])
    assert all(np.array_equal(a, b) for a, b in zip(ss.planes(1), ref.planes(0)))
    ss.close(); ref.close()
    for bad in ([(1, 0, 4, False)], [(0, 0, len(w) + 1, False)], [(0, -1, 4, False)]):
        with pytest.raises(VaporHipError):
            eng.seqset([w], derived=[(bad, False)])
    # a derived sequence is slices of its parents' BYTES - not upper-cased itself, it cannot be cut from a parent that
    # was upper-cased at upload (whose planes hold the upper-cased text); its upper-cased twin can
    soft = "acgtTTGACCAGGTTAACCAGTacgt" * 3
    with pytest.raises(VaporHipError) as ei:
        eng.seqset([soft], upper=[True], derived=[([(0, 2, 30, False)], False)])
    assert ei.value.code == L.E_ARG
    ss = eng.seqset([soft], upper=[True], derived=[([(0, 2, 30, False)], True)])
    ref = eng.seqset([soft[2:32].upper()])
    assert all(np.array_equal(a, b) for a, b in zip(ss.planes(1), ref.planes(0)))
    ss.close(); ref.close()


def _plots(eng, ss, pairs):
    plan = eng.plan(ss, pairs)
    try:
        st = plan.run().copy()
        hits, _f, off = plan.fetch_hits(range(plan.n), want_flags=False)
        tm = plan.timings()
    finally:
        plan.close()
    dots = []
    for t in range(len(off) - 1):
        h = hits[off[t]:off[t + 1]]
        dots.append(h[np.lexsort((h[:, 1], h[:, 0]))])
    return st, dots, tm


def check_shared_joins(eng, oracle, ks=(10, 20, 30, 40), want_shared=True):
    rng = np.random.default_rng(909)
    total_served = 0
    for tag, lits, der in allele_cases(rng):
        ref = lits[0]
        n_lit = len(lits)
        # reads from the window and from the alleles (PacBio-like errors), one with an N, one with a lower-case stretch
        reads = []
        for src in (ref, der[0][2], der[1][2], der[3][2], der[4][2], der[7][2], der[12][2]):
            r, _ = synth.mutate(rng, src[:1800], 0.01, 0.05, 0.03)
            reads.append(r)
        reads.append(reads[1][:300] + "N" + reads[1][301:])
        reads.append(reads[2][:200] + reads[2][200:260].lower() + reads[2][260:])
        reads.append(ref[:1500])                                 # a perfect read: long runs
        seqs = lits + reads
        first_read = n_lit
        first_der = n_lit + len(reads)
        # a group shares the joins of its window and at most three derived alleles: the alleles in threes (every structure
        # gets its turn), the two upper-cased twins as a group of their own
        plain = [d for d in range(len(der)) if not der[d][1]]
        chunks = [plain[a:a + 3] for a in range(0, len(plain), 3)] + [[5, 6]]
        for k in ks:
            for ch in chunks:
                sub = [der[d] for d in ch]
                derived = [([(p, o, n, bool(rc)) for p, o, n, rc in sg], up) for sg, up, _t in sub]
                texts = seqs + [t for _sg, _u, t in sub]              # the text of every sequence of the set, by index
                upper = sub[0][1]
                rows = []
                for r in range(len(reads)):
                    miss = (0, 37, 0, 5)[r % 4]
                    for a in ([] if upper else [0]) + [first_der + d for d in range(len(sub))]:
                        rows.append((first_read + r, a, miss, k, L.PF_C1 if upper else (L.PF_C1 | L.PF_C2 | L.PF_DIR)))
                pairs = eng.make_pairs(rows)
                ss = eng.seqset(seqs, derived=derived)
                try:
                    st, dots, tm = _plots(eng, ss, pairs)
                    eng.set_param("shared_join", 0)
                    try:
                        ss0 = eng.seqset(seqs, derived=derived)
                        st0, dots0, tm0 = _plots(eng, ss0, pairs)
                        ss0.close()
                    finally:
                        eng.set_param("shared_join", 1)
                finally:
                    ss.close()
                assert tm0["shared_joins"] == 0
                if want_shared and len(rows) // len(reads) >= 2 and min(len(t) for _s, _u, t in sub) >= k + 40:
                    assert tm["shared_joins"] > 0 and tm["pairs_served_by_shared_joins"] >= 2 * tm["shared_joins"], (tag, k, ch, tm)
                total_served += tm["pairs_served_by_shared_joins"]
                assert np.array_equal(st, st0), (tag, k, ch, np.argwhere(st != st0)[:5])
                for t, (s1, s2, off2, kk, _fl) in enumerate(rows):
                    assert np.array_equal(dots[t], dots0[t]), (tag, k, ch, t)
                    exp = oracle.dotdata_array(kk, texts[s1], texts[s2][off2:])
                    assert np.array_equal(dots[t], exp.reshape(-1, 2)), (tag, k, ch, t, len(dots[t]), len(exp))
                    assert st[t, 0] == len(exp)
    return total_served


def test_shared_joins_give_the_dots_of_separate_joins(eng, oracle):
    assert check_shared_joins(eng, oracle) > 1000


@pytest.mark.parametrize("route", [0, 2])
def test_shared_joins_cut_up_by_either_route(eng, oracle, route):
    # remap_in_clean 0: remap_kernel cuts every target's records out of the shared dot plots before the cleaning starts;
    # 2: the clean workgroup of each target does (1, the default, picks by the plan's size); both give the dots of separate joins
    eng.set_param("remap_in_clean", route)
    try:
        assert check_shared_joins(eng, oracle, ks=(10, 30)) > 300
    finally:
        eng.set_param("remap_in_clean", 1)


def test_shared_joins_through_the_loci_path(eng):
    """The bench shapes with derived alt windows: per-locus records of run_loci equal those of the same batch with every
    sequence uploaded as bytes (one join per pair), for DEL / TANDUP / INV / INS - and the plan shares its joins."""
    from vapor_amd import workload as wl
    spec = dict(wl.WORKLOADS["tiny"])
    spec["n_loci"] = 16
    w = wl.make_workload("tiny", seed=4242, **spec)
    assert w.derived and w.n_lit < len(w.seqs)
    recs = []
    for shared in (True, False):
        ss = w.upload(eng) if shared else eng.seqset(w.seqs)
        plan = eng.plan(ss, w.pairs)
        plan.set_reads(wl.read_table(w), w.n_loci)
        recs.append((plan.run_loci(want_scores=True).copy(), plan.read_scores.copy(), plan.run().copy(), plan.timings()))
        # the asynchronous steps give the same records
        plan.run_loci()
        plan.run_loci_async(); plan.run_loci_async()
        again = plan.sync().copy()
        assert np.array_equal(np.isnan(again), np.isnan(recs[-1][0])) and np.array_equal(again[~np.isnan(again)], recs[-1][0][~np.isnan(recs[-1][0])])
        plan.close(); ss.close()
    (a, sa, sta, ta), (b, sb, stb, tb) = recs
    # (a read shorter than the window size has no dot plot to share: the tiny shape has a few)
    n_joined = sum(1 for p in w.pairs[0::2] if len(w.seqs[p["seq1"]]) >= 10)
    assert n_joined > 64
    assert ta["shared_joins"] == n_joined and ta["pairs_served_by_shared_joins"] == 2 * n_joined and tb["shared_joins"] == 0
    assert np.array_equal(sta, stb)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    assert np.array_equal(np.isnan(sa), np.isnan(sb)) and np.array_equal(sa[~np.isnan(sa)], sb[~np.isnan(sb)])


def test_shared_join_overflow_is_resized(eng, oracle):
    """A shared dot plot that outgrows its first slot (a low-complexity window: far more dots than min(n1, n2)) is counted,
    resized and rerun like any pair; the targets' dots are the oracle's."""
    unit = "ACGGTCATTG"
    ref = "TTGACCAGTCCATGGACTAGC" * 10 + unit * 150 + "GGATCCATTGACGTTAGCATC" * 10
    n = len(ref)
    alt = ref[:200] + ref[400:]
    read = (unit * 120)[:1100]
    ss = eng.seqset([ref, read], derived=[([(0, 0, 200, False), (0, 400, n - 400, False)], False)])
    pairs = eng.make_pairs([(1, 0, 0, 10, 3), (1, 2, 0, 10, 3)])
    st, dots, tm = _plots(eng, ss, pairs)
    ss.close()
    assert tm["shared_joins"] == 1
    for t, a in enumerate((ref, alt)):
        exp = oracle.dotdata_array(10, read, a)
        assert st[t, 15] == 0 and st[t, 0] == len(exp) and np.array_equal(dots[t], exp.reshape(-1, 2))
    assert st[0, 0] > 20000


@pytest.mark.parametrize("route", [0, 2])
def test_served_pairs_on_the_big_path_equal_pairs_joined_on_their_own(eng, oracle, route):
    """Dot plots too large for a clean workgroup's LDS (more than 65 535 dots: clean_big_kernel) that are SERVED by a shared join,
    cut out by either route: every statistic and every per-dot flag equals those of the same texts uploaded as bytes and joined
    pair by pair, the dots equal the oracle's - for a deletion allele and for one with a reversed stretch."""
    unit = "ACGGTCATTG"
    ref = "TTGACCAGTCCATGGACTAGC" * 10 + unit * 300 + "GGATCCATTGACGTTAGCATC" * 10
    n = len(ref)
    del_segs = [(0, 0, 400, False), (0, 900, n - 900, False)]
    inv_segs = [(0, 0, 500, False), (0, 500, 700, True), (0, 1200, n - 1200, False)]
    del_txt = ref[:400] + ref[900:]
    inv_txt = ref[:500] + _rc(ref[500:1200]) + ref[1200:]
    reads = [(unit * 200)[:1900], ref[100:2100], _rc(ref[300:1700])]
    rows_d, rows_l = [], []
    for r in range(len(reads)):
        for a in range(3):                                   # window, deletion allele, inversion allele
            miss = (0, 13)[r % 2]
            rows_d.append((1 + r, 0 if a == 0 else 3 + a, miss, 10, L.PF_C1 | L.PF_C2 | L.PF_DIR))
            rows_l.append((1 + r, 0 if a == 0 else 3 + a, miss, 10, L.PF_C1 | L.PF_C2 | L.PF_DIR))
    texts = [ref] + reads + [del_txt, inv_txt]

    def run(ss, rows):
        plan = eng.plan(ss, eng.make_pairs(rows))
        try:
            st = plan.run().copy()
            hits, fl, off = plan.fetch_hits(range(plan.n), want_flags=True)
            tm = plan.timings()
            again = plan.run().copy()                         # (a second pass over sized slots gives the same)
        finally:
            plan.close()
        assert np.array_equal(st, again)
        per = []
        for t in range(len(off) - 1):
            h, f = hits[off[t]:off[t + 1]], fl[off[t]:off[t + 1]]
            o = np.lexsort((f, h[:, 1], h[:, 0]))             # (a dot can lie on both strands: its two flags in a fixed order)
            per.append((h[o], f[o]))
        return st, per, tm

    eng.set_param("remap_in_clean", route)
    try:
        ss = eng.seqset([ref] + reads, derived=[(del_segs, False), (inv_segs, False)])
        try:
            st, per, tm = run(ss, rows_d)
        finally:
            ss.close()
    finally:
        eng.set_param("remap_in_clean", 1)
    assert tm["shared_joins"] == len(reads) and tm["pairs_served_by_shared_joins"] == 3 * len(reads)
    assert tm["remap_in_clean"] == (1 if route == 2 else 0)
    lit = eng.seqset(texts)
    try:
        st0, per0, tm0 = run(lit, rows_l)
    finally:
        lit.close()
    assert tm0["shared_joins"] == 0
    assert int(st0[:, 0].max()) > 65535                       # (the case is on the big path)
    assert np.array_equal(st, st0), np.argwhere(st != st0)[:8]
    for t, (s1, s2, off2, k, _f) in enumerate(rows_l):
        assert np.array_equal(per[t][0], per0[t][0]) and np.array_equal(per[t][1], per0[t][1]), t
        exp = oracle.dotdata_array(k, texts[s1], texts[s2][off2:]).reshape(-1, 2)
        assert np.array_equal(per[t][0], exp), (t, len(per[t][0]), len(exp))


def check_random_structures(eng, oracle, n_windows=6, seed=1234):
    """Random segment lists - two to six slices of a window (some reversed, some a few bases short, some overlapping, some
    from a second text) - as derived alleles: the dots of every (read, allele) pair equal the oracle's on the text the segments
    spell, with the joins shared.  Lower-case stretches and N in windows and reads take the pairs through every symbol mode."""
    rng = np.random.default_rng(seed)
    served = 0
    for wi in range(n_windows):
        n = int(rng.integers(600, 5000))
        win = synth.random_dna(rng, n)
        other = synth.random_dna(rng, int(rng.integers(50, 700)))
        if wi % 3 == 1:
            a = int(rng.integers(0, n - 200))
            win = win[:a] + win[a:a + 150].lower() + win[a + 150:]
        if wi % 3 == 2:
            a = int(rng.integers(0, n - 50))
            win = win[:a] + "N" * 5 + win[a + 5:]
        lits = [win, other]
        der, texts = [], []
        for _ in range(3):
            segs, txt = [], ""
            for _s in range(int(rng.integers(2, 7))):
                par = 0 if rng.random() < 0.85 else 1
                plen = len(lits[par])
                ln = int(rng.choice([int(rng.integers(1, 25)), int(rng.integers(25, 400)), int(rng.integers(400, max(401, plen)))]))
                ln = min(ln, plen)
                off = int(rng.integers(0, plen - ln + 1))
                rc = bool(rng.random() < 0.3)
                piece = lits[par][off:off + ln]
                segs.append((par, off, ln, rc))
                txt += _rc(piece) if rc else piece
            der.append((segs, False))
            texts.append(txt)
        reads = []
        for src in [win] + texts:
            r, _ = synth.mutate(rng, src[:3000], 0.01, 0.05, 0.03)
            reads.append(r)
        reads.append(reads[0][:120] + "n" * 3 + reads[0][123:])
        seqs = lits + reads
        fr, fd = len(lits), len(lits) + len(reads)
        allt = seqs + texts
        for k in (10, 20, 40):
            rows = []
            for r in range(len(reads)):
                miss = int(rng.integers(0, 60)) if r % 2 else 0
                for a in [0] + [fd + d for d in range(len(der))]:
                    rows.append((fr + r, a, miss, k, L.PF_C1 | L.PF_C2 | L.PF_DIR))
            ss = eng.seqset(seqs, derived=der)
            try:
                st, dots, tm = _plots(eng, ss, eng.make_pairs(rows))
            finally:
                ss.close()
            served += tm["pairs_served_by_shared_joins"]
            # every statistic (both cleaners, the counts, the directed statistics) equals that of the same texts uploaded as
            # bytes and joined pair by pair - the path tests/test_gpu_parity.py pins on the oracle
            lit = eng.seqset(allt)
            try:
                st_lit, _d, tm_lit = _plots(eng, lit, eng.make_pairs(rows))
            finally:
                lit.close()
            assert tm_lit["shared_joins"] == 0
            assert np.array_equal(st, st_lit), (wi, k, np.argwhere(st != st_lit)[:6])
            for t, (s1, s2, off2, kk, _fl) in enumerate(rows):
                exp = oracle.dotdata_array(kk, allt[s1], allt[s2][off2:]).reshape(-1, 2)
                assert st[t, 15] == 0 and st[t, 0] == len(exp) and np.array_equal(dots[t], exp), (wi, k, t, der, len(dots[t]), len(exp))
    return served


def test_shared_joins_on_random_segment_structures(eng, oracle):
    assert check_random_structures(eng, oracle) > 100        # (pairs that were served by shared joins: random structures do not all qualify)


def test_shared_joins_fuzz_at_length(eng, oracle):
    """The same check at the length asked for: VAPOR_FUZZ_WINDOWS windows for each seed of VAPOR_FUZZ_SEEDS (comma-separated),
    under both routes of the cutting; by default one more seed of six windows per route.  Every (read, allele) pair's dots equal
    the oracle's dotdata on the text the segments spell, and all sixteen statistics those of the same texts uploaded as bytes.  profiles/r04_fuzz_shared_joins.txt is the log of a long run."""
    import os
    n_win = int(os.environ.get("VAPOR_FUZZ_WINDOWS", "6"))
    seeds = [int(x) for x in os.environ.get("VAPOR_FUZZ_SEEDS", "77").split(",")]
    for route in (0, 2):
        eng.set_param("remap_in_clean", route)
        try:
            for seed in seeds:
                served = check_random_structures(eng, oracle, n_windows=n_win, seed=seed)
                print("fuzz: route %d seed %d: %d windows x 3 derived alleles x 3 k, %d pairs served by shared joins, all dots = oracle's, all statistics = the byte upload's" % (route, seed, n_win, served), flush=True)
        finally:
            eng.set_param("remap_in_clean", 1)


def test_shared_plot_beyond_max_pair_cap_marks_every_served_pair(eng, oracle):
    """a SHARED dot plot that needs more records than max_pair_cap cannot grow; the pairs cut from it would be
    cut from a truncated plot although their own slots do not overflow.  Every pair it serves keeps VAPOR_E_OVERFLOW
    (include/vapor_hip.h, "max_pair_cap") through vapor_plan_run and vapor_plan_run_loci - which used to call itself without
    end here - under both routes; with the default cap the same plan gives the oracle's dots."""
    unit = "ACGGTCATTG"
    ref = "TTGACCAGTCCATGGACTAGC" * 10 + unit * 150 + "GGATCCATTGACGTTAGCATC" * 10
    n = len(ref)
    a0, a1 = 215, 210 + 1495                                   # the deletion takes nearly all of the repeat
    alt = ref[:a0] + ref[a1:]
    read = (unit * 120)[:1100]
    rows = [(1, 0, 0, 10, 3), (1, 2, 0, 10, 3)]
    table = np.zeros(1, dtype=L.READ_DTYPE)
    table["ref_a"] = table["ref_b"] = 0
    table["alt_a"] = table["alt_b"] = 1
    table["kind"], table["len_ref"], table["len_alt"] = 1, n, len(alt)
    exp_alt = oracle.dotdata_array(10, read, alt)
    for route in (0, 2):
        eng.set_param("remap_in_clean", route)
        eng.set_param("max_pair_cap", 3000)
        try:
            ss = eng.seqset([ref, read], derived=[([(0, 0, a0, False), (0, a1, n - a1, False)], False)])
            plan = eng.plan(ss, eng.make_pairs(rows))
            plan.set_reads(table, 1)
            st = plan.run().copy()
            tm = plan.timings()
            assert tm["shared_joins"] == 1
            assert len(exp_alt) < 3000                            # (the alt pair's own slot is large enough ...)
            assert st[:, 15].tolist() == [L.E_OVERFLOW, L.E_OVERFLOW], (route, st[:, 15])     # (... its source is not)
            loci = plan.run_loci(want_scores=True).copy()         # terminates, and scores nothing from a truncated plot
            assert np.isnan(loci[0, 0]) and np.isnan(plan.read_scores[0])
            again = plan.run_loci().copy()
            assert np.isnan(again[0, 0])
            plan.close(); ss.close()
        finally:
            eng.set_param("max_pair_cap", 1 << 28)
            eng.set_param("remap_in_clean", 1)
    ss = eng.seqset([ref, read], derived=[([(0, 0, a0, False), (0, a1, n - a1, False)], False)])
    st, dots, tm = _plots(eng, ss, eng.make_pairs(rows))
    ss.close()
    assert tm["shared_joins"] == 1 and st[:, 15].tolist() == [0, 0]
    assert np.array_equal(dots[1], exp_alt.reshape(-1, 2)) and st[0, 0] > 20000


def test_a_second_upper_twin_of_a_window_takes_no_member_slot(eng, oracle):
    """several requests of a batch may each describe the upper-cased twin of one window.  The first is the
    group's identity; a second one must not take one of the three member slots (it would push a real allele out of the shared
    join) - its pairs are joined on their own - and every pair's dots stay the oracle's."""
    rng = np.random.default_rng(31)
    g = synth.random_dna(rng, 2600)
    win = g[:900] + g[900:1000].lower() + g[1000:]
    n, f = len(win), 500
    der = [([(0, 0, n, False)], True), ([(0, 0, n, False)], True),                                     # two whole upper twins
           ([(0, 0, f, False), (0, n - f, f, False)], True),                                            # DEL, upper
           ([(0, 0, f, False), (0, f, n - 2 * f, True), (0, n - f, f, False)], True),                   # INV, upper
           ([(0, 0, n - f, False), (0, f, n - 2 * f, False), (0, n - f, f, False)], True)]              # TANDUP, upper
    up = win.upper()
    texts = [up, up, up[:f] + up[-f:], up[:f] + _rc(up[f:n - f]) + up[-f:], up[:n - f] + up[f:n - f] + up[-f:]]
    reads = [synth.mutate(rng, t[:1900], 0.01, 0.05, 0.03)[0] for t in texts[1:]]
    seqs = [win] + reads
    fd = len(seqs)
    rows = [(1 + r, fd + d, (0, 9)[r % 2], 10, L.PF_C1) for r in range(len(reads)) for d in range(len(der))]
    ss = eng.seqset(seqs, derived=der)
    st, dots, tm = _plots(eng, ss, eng.make_pairs(rows))
    ss.close()
    # every read shares one join for the identity twin and the three alleles; the second twin's pairs run joins of their own
    assert tm["shared_joins"] == len(reads) and tm["pairs_served_by_shared_joins"] == 4 * len(reads), tm
    for t, (s1, s2, off2, k, _fl) in enumerate(rows):
        exp = oracle.dotdata_array(k, seqs[s1], texts[s2 - fd][off2:]).reshape(-1, 2)
        assert st[t, 15] == 0 and np.array_equal(dots[t], exp), (t, len(dots[t]), len(exp))
