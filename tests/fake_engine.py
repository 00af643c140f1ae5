"""TEST INFRASTRUCTURE: an object with the vapor_amd.engine.Engine interface whose numbers come
from the CPU oracle.  It lets the CPU-only suite exercise the *host* logic (drivers, batching,
finishing, CLI, sharding) against reference-generated vectors without a GPU.  It lives under
tests/ and nothing in vapor_amd/ can reach it: the product has no CPU path."""
import numpy as np

from vapor_amd import _lib as L


_VALID = set("ACGTNRYSWKMBDHVacgtnryswkmbdhv")       # what pack_kernel's sym_code accepts (IUPAC folds to N, SF:908-949)


def _materialise(seqs, derived):
    """Derived sequences (vapor_amd.engine.SeqSet's `derived`: [(segments of (parent, off, len, revcomp), upper)]) as text, built
    the way the reference builds them: slices, reverse(complementary()) of slices (SF:471-478), str.upper()."""
    from vapor_amd import seqio
    out = []
    for segs, up in derived or ():
        t = "".join(seqio.reverse(seqio.complementary(seqs[p][o:o + n])) if rc else seqs[p][o:o + n] for p, o, n, rc in segs)
        out.append(t.upper() if up else t)
    return out


class _SeqSet:
    def __init__(self, seqs, upper, derived=None):
        self.seqs = [s.upper() if u else s for s, u in zip(seqs, upper)] + _materialise(seqs, derived)
        self.n = len(self.seqs)
        self.n_derived = len(derived or ())
        self.lens = np.array([len(s) for s in self.seqs], dtype=np.int32)
        self.n_invalid = np.array([sum(1 for ch in set(s) if ch not in _VALID) for s in self.seqs], dtype=np.int32)
        self.n_exc = np.array([sum(1 for ch in s if ch not in "ACGT") for s in self.seqs], dtype=np.int32)

    def close(self):
        pass


class _Plan:
    def __init__(self, orc, ss, pairs):
        self.orc, self.ss, self.pairs, self.n = orc, ss, pairs, len(pairs)
        self.stats = np.zeros((max(self.n, 1), 16), dtype=np.int64)
        self._hits = {}

    def run(self):
        o = self.orc
        for t, p in enumerate(self.pairs):
            k, fl = int(p["k"]), int(p["flags"])
            s1, s2 = self.ss.seqs[p["seq1"]], self.ss.seqs[p["seq2"]][int(p["off2"]):]
            row = np.zeros(16, dtype=np.int64)
            row[1] = row[2] = -1
            if len(s1) > L.MAX_SEQ_LEN or len(self.ss.seqs[p["seq2"]]) > L.MAX_SEQ_LEN or k not in (10, 20, 30, 40):
                row[15] = L.E_ARG
                self.stats[t] = row
                continue
            try:
                st, h, k1, k2 = o.pair_stats(k, s1, s2, want_hits=True)
            except KeyError:
                row[15] = L.E_KEYERROR
                self.stats[t] = row
                continue
            row[:10] = st[:10]
            if not fl & 1:
                row[3] = row[4] = 0
            if not fl & 2:
                row[5] = row[6] = row[9] = 0
            if fl & 4 and fl & 1 and st[3] > 0:
                kept = [(int(a), int(b)) for a, b in h[k1 > 0]]
                c = o.dis_to_diagnal_most_abundant_defined(list(kept))
                far = [d for d in ([a + c, b] for a, b in kept) if o.eu_dis_single_dot(d) > 0.1]
                row[10] = int(round(2 * float(c)))
                row[11] = len(far)
                row[12] = int(round(2 * sum(d[0] - d[1] for d in far)))
            self.stats[t] = row
            self._hits[t] = h
        return self.stats[:self.n]

    def set_reads(self, reads, n_loci):
        self.reads, self.n_loci = reads, n_loci
        self.read_scores = np.zeros(max(len(reads), 1), dtype=np.float64)

    def run_loci(self, device_out=0, want_host=True, want_scores=False):
        """The per-read reduction finish_kernel does on the device, with the host functions of vapor_amd.finish
        (which restate SF:182-294 and are themselves checked against the reference's vectors)."""
        from vapor_amd import finish
        st = self.run()
        fn = {1: finish.score_abs_dis_m1b, 2: finish.score_within_10Perc_m1b, 3: finish.score_directed_dis_m1b_redefine_diagnal}

        def ratio(ab):
            return None if 0 in ab else 1 - float(ab[1]) / float(ab[0])
        for t, r in enumerate(self.reads):
            lr, la = int(r["len_ref"]), int(r["len_alt"])
            if int(r["kind"]) == 0:
                s1 = ratio(finish.score_abs_dis_m1b(st[r["ref_a"]], st[r["alt_a"]], lr, la))
                s2 = ratio(finish.score_within_10Perc_m1b(st[r["ref_b"]], st[r["alt_b"]], lr, la))
                v = min([s1, s2]) if s1 is not None and s2 is not None else s1 if s1 is not None else s2
            else:
                v = ratio(fn[int(r["kind"])](st[r["ref_a"]], st[r["alt_a"]], lr, la))
            self.read_scores[t] = np.nan if v is None else v
        return None

    def fetch_hits(self, idx, want_flags=True):
        idx = list(idx)
        hs = [self._hits.get(int(t), np.zeros((0, 2), np.int32)) for t in idx]
        off = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum([len(h) for h in hs], out=off[1:])
        allh = np.concatenate(hs) if hs else np.zeros((0, 2), np.int32)
        return allh, (np.zeros(len(allh), np.uint8) if want_flags else None), off

    def close(self):
        pass


class FakeEngine:
    def __init__(self, orc):
        self.orc = orc
        self.batches = []        # (n_seqs, n_pairs) per plan, for batching assertions

    def seqset(self, seqs, upper=None, derived=None):
        self.derived_seen = getattr(self, "derived_seen", 0) + len(derived or ())
        return _SeqSet(list(seqs), list(upper) if upper is not None else [False] * len(seqs), derived)

    def seqset_raw(self, addr, lens, derived=None, keepalive=None):
        """Sequences given by address (vapor_amd.engine.SeqSet.from_addresses): read back as text."""
        import ctypes
        seqs = [ctypes.string_at(int(a), int(n)).decode("ascii") for a, n in zip(addr, lens)]
        der = None
        if derived is not None:
            seg_first, segs, dflags = derived
            der = [([(int(g["parent"]), int(g["off"]), int(g["len"]), bool(g["flags"] & 1)) for g in segs[seg_first[d]:seg_first[d + 1]]],
                    bool(dflags[d] & 1)) for d in range(len(seg_first) - 1)]
        self.raw_sets = getattr(self, "raw_sets", 0) + 1
        return self.seqset(seqs, None, der)

    def plan(self, ss, pairs):
        self.batches.append((ss.n, len(pairs)))
        return _Plan(self.orc, ss, pairs)

    @staticmethod
    def make_pairs(rows):
        a = np.zeros(len(rows), dtype=L.PAIR_DTYPE)
        for t, r in enumerate(rows):
            a[t] = tuple(r)
        return a

    def score(self, ss, pairs):
        return self.plan(ss, pairs).run().copy()

    def dotplots(self, ss, pairs):
        p = self.plan(ss, pairs)
        st = p.run().copy()
        return st, [p._hits.get(t, np.zeros((0, 2), np.int32)) for t in range(p.n)]

    def clean_hits(self, lists, flags=None):
        st = np.zeros((len(lists), 16), dtype=np.int64)
        out = []
        for t, h in enumerate(lists):
            h = np.asarray(h, dtype=np.int32).reshape(-1, 2)
            k1 = self.orc.clean_c1_flags(h)
            k2 = self.orc.clean_c2_flags(h)
            out.append(((k1 > 0) * 1 + (k2 == 1) * 2 + (k2 == 2) * 4).astype(np.uint8))
        return st, out
