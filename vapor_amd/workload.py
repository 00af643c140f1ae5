"""Kernel-level synthetic workloads at the BASELINE.json shapes and the
vectorised per-locus finish used by bench.py and the batch pipeline.

A workload is a set of loci; each locus has a ref and an alt allele window and a list of
reads, every read is scored against both windows with the scorer(s) its SV type uses in the
reference's drivers:  DEL -> abs_dis_m1b and within_10Perc_m1b (min of the
two, SF:1718-1726), TANDUP -> directed_dis_m1b_redefine_diagnal (SF:1761), INV / INS ->
abs_dis_m1b (SF:1910, SF:1879).
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence

import numpy as np

from . import _lib as L
from . import finish, synth

SCORER_FLAGS = {"DEL": L.PF_C1 | L.PF_C2, "TANDUP": L.PF_C1 | L.PF_DIR, "INV": L.PF_C1, "INS": L.PF_C1}


@dataclasses.dataclass
class Workload:
    name: str
    seqs: List[str]              # alleles and reads, one device sequence each
    pairs: np.ndarray            # PAIR_DTYPE, ref pair then alt pair for every read
    read_locus: np.ndarray       # locus index of every read
    read_kind: np.ndarray        # 0 DEL (S1+S2), 1 S1 only, 3 S3
    len_ref: np.ndarray          # per read: full length of its locus' ref / alt window
    len_alt: np.ndarray
    n_loci: int
    svtypes: List[str]
    # The alt windows as the reference's drivers build them - by string surgery on the ref window (SF:1712, 1755, 1907, 1872) -
    # described instead of uploaded (include/vapor_hip.h, vapor_seqset_create_derived): seqs[:n_lit] travel as bytes,
    # seqs[n_lit + d] is `derived[d]` = (segments of (parent, off, len, revcomp), upper) and is assembled on the device.
    # derived = None: every sequence as bytes (the same strings, the same indices).
    n_lit: int = 0
    derived: list = None

    def upload(self, eng):
        """The batch's sequence set on `eng`."""
        if self.derived:
            arr = self.__dict__.get("_derived_arrays")
            if arr is None:
                # (the descriptors as the arrays the library takes, made once: a caller that streams batches builds them
                # with numpy, as pipeline / fastpath do - not segment by segment on every upload)
                seg_first = np.zeros(len(self.derived) + 1, dtype=np.int32)
                np.cumsum([len(sg) for sg, _u in self.derived], out=seg_first[1:])
                segs = np.zeros(max(int(seg_first[-1]), 1), dtype=L.SEG_DTYPE)
                rows = [(p, o, n, L.SEG_REVCOMP if rc else 0) for sg, _u in self.derived for p, o, n, rc in sg]
                if rows:
                    segs[:len(rows)] = rows
                arr = self.__dict__["_derived_arrays"] = (seg_first, segs, np.asarray([L.SEQ_UPPER if u else 0 for _sg, u in self.derived], dtype=np.uint8))
                self.__dict__["_lits"] = self.seqs[:self.n_lit]
            return eng.seqset(self.__dict__["_lits"], derived=arr)
        return eng.seqset(self.seqs)


def make_workload(name: str, seed: int, n_loci: int, svtypes: Sequence[str], read_len: int, allele_len: int,
                  reads_per_locus: int, k: int = 10, derived: bool = True) -> Workload:
    rng = np.random.default_rng(seed)
    seqs: List[str] = []             # in creation order; ("alt", n) entries are moved behind the literals below
    is_alt: List[bool] = []
    segs_of = {}
    rows = []
    read_locus, read_kind, len_ref, len_alt, types = [], [], [], [], []
    for li in range(n_loci):
        t = svtypes[li % len(svtypes)]
        types.append(t)
        ref = synth.random_dna(rng, allele_len)
        span = int(rng.integers(300, 3000))
        s = int(rng.integers(allele_len // 4, allele_len // 2))
        ri = len(seqs)
        seqs.append(ref); is_alt.append(False)
        span = min(span, allele_len - s)                  # (what the slices below take on a short window)
        if t == "DEL":
            alt = ref[:s] + ref[s + span:]
            sg = [(ri, 0, s, False), (ri, s + span, allele_len - s - span, False)]
        elif t == "TANDUP":
            alt = ref[:s + span] + ref[s:s + span] + ref[s + span:]
            sg = [(ri, 0, s + span, False), (ri, s, span, False), (ri, s + span, allele_len - s - span, False)]
        elif t == "INV":
            alt = ref[:s] + synth.revcomp(ref[s:s + span]) + ref[s + span:]
            sg = [(ri, 0, s, False), (ri, s, span, True), (ri, s + span, allele_len - s - span, False)]
        else:
            ins = synth.random_dna(rng, span)
            alt = ref[:s] + ins + ref[s:]
            xi = len(seqs)
            seqs.append(ins); is_alt.append(False)          # (the inserted bytes are a sequence of their own, as ins_seq is, SF:1856)
            sg = [(ri, 0, s, False), (xi, 0, span, False), (ri, s, allele_len - s, False)]
        ai = len(seqs)
        seqs.append(alt); is_alt.append(True)
        segs_of[ai] = sg
        kind = {"DEL": 0, "TANDUP": 3}.get(t, 1)
        for _ in range(reads_per_locus):
            hap = alt if rng.random() < 0.5 else ref
            lo = max(0, s + span + 500 - read_len)
            hi = max(lo, min(s - 200, len(hap) - read_len))
            st = int(rng.integers(lo, hi + 1))
            rd, _c = synth.mutate(rng, hap[st:st + read_len + read_len // 8], cigar=False)
            seqs.append(rd[:read_len]); is_alt.append(False)
            q = len(seqs) - 1
            rows.append((q, ri, 0, k, SCORER_FLAGS[t]))
            rows.append((q, ai, 0, k, SCORER_FLAGS[t]))
            read_locus.append(li)
            read_kind.append(kind)
            len_ref.append(len(ref))
            len_alt.append(len(alt))
    pairs = np.array(rows, dtype=np.int64).view(np.int64).reshape(-1, 5)
    pa = np.zeros(len(rows), dtype=L.PAIR_DTYPE)
    for c, name in enumerate(("seq1", "seq2", "off2", "k", "flags")):
        pa[name] = pairs[:, c]
    pairs = pa
    n_lit, der = len(seqs), None
    if derived:
        # literals first (their order kept), the alt windows behind them as derived sequences
        lit = [t for t in range(len(seqs)) if not is_alt[t]]
        alts = [t for t in range(len(seqs)) if is_alt[t]]
        new = np.empty(len(seqs), dtype=np.int64)
        new[lit] = np.arange(len(lit))
        new[alts] = len(lit) + np.arange(len(alts))
        der = [([(int(new[p]), o, n, rc) for p, o, n, rc in segs_of[t]], False) for t in alts]
        seqs = [seqs[t] for t in lit] + [seqs[t] for t in alts]
        pairs["seq1"] = new[pairs["seq1"]]
        pairs["seq2"] = new[pairs["seq2"]]
        n_lit = len(lit)
    return Workload(name, seqs, pairs, np.asarray(read_locus), np.asarray(read_kind), np.asarray(len_ref),
                    np.asarray(len_alt), n_loci, types, n_lit, der)


WORKLOADS = {
    # BASELINE.json configs[1]: 100 simple DEL/DUP loci, 10 kb reads at 20x, 20 kb windows
    "cfg2": dict(n_loci=100, svtypes=("DEL", "TANDUP"), read_len=10000, allele_len=20000, reads_per_locus=20),
    # configs[2]: 1000 mixed DEL/DUP/INV/INS loci, 15 kb reads at 40x
    "cfg3": dict(n_loci=1000, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), read_len=15000, allele_len=20000,
                 reads_per_locus=40),
    # the largest shape BASELINE names (configs[4]: 30 kb reads, 40 kb windows, 60x), a single-GPU slice of it
    "cfg5s": dict(n_loci=200, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), read_len=30000, allele_len=40000,
                  reads_per_locus=60),
    # a shape whose clean workgroup needs ~5 KB of LDS (experiments on what runs beside a join workgroup)
    "small": dict(n_loci=2000, svtypes=("DEL", "TANDUP"), read_len=2000, allele_len=4000, reads_per_locus=20),
    "tiny": dict(n_loci=8, svtypes=("DEL", "TANDUP", "INV", "INS"), read_len=1500, allele_len=3000, reads_per_locus=6),
}


def read_table(w: Workload) -> np.ndarray:
    """READ_DTYPE rows for Plan.set_reads: read r is scored by pairs 2r (ref) and 2r+1 (alt)."""
    n = len(w.read_locus)
    t = np.zeros(n, dtype=L.READ_DTYPE)
    t["ref_a"] = t["ref_b"] = 2 * np.arange(n)
    t["alt_a"] = t["alt_b"] = 2 * np.arange(n) + 1
    t["kind"] = w.read_kind
    t["locus"] = w.read_locus
    t["len_ref"] = w.len_ref
    t["len_alt"] = w.len_alt
    return t


def finish_workload(w: Workload, stats: np.ndarray) -> np.ndarray:
    """Per-locus records [QS, GS, GT index, GQ, n_scored] (float64; NaN row = 'NA') from the
    statistics of one pass (L1 -> L2 -> L3)."""
    st_ref, st_alt = stats[0::2], stats[1::2]
    kind = w.read_kind
    k1 = np.where(kind == 3, 3, 1)
    a1, b1, v1 = finish.batch_scores(k1, st_ref, st_alt, w.len_ref, w.len_alt)
    s1 = finish.read_scores(a1, b1)
    is_del = kind == 0
    score = s1.copy()
    valid = v1.copy()
    if is_del.any():
        a2, b2, v2 = finish.batch_scores(np.where(is_del, 2, 0), st_ref, st_alt, w.len_ref, w.len_alt)
        s2 = finish.read_scores(a2, b2)
        both = is_del & v1 & v2
        score[both] = np.minimum(s1[both], s2[both])
        only2 = is_del & ~v1 & v2
        score[only2] = s2[only2]
        valid = np.where(is_del, v1 | v2, v1)
    out = np.full((w.n_loci, 5), np.nan, dtype=np.float64)
    loc = w.read_locus[valid]
    sc = score[valid]
    n = np.bincount(loc, minlength=w.n_loci)
    pos = sc > 0
    npos = np.bincount(loc[pos], minlength=w.n_loci)
    spos = np.bincount(loc[pos], weights=sc[pos], minlength=w.n_loci)
    nonpos_r = np.bincount(loc[finish.rounded_nonpositive(sc)], minlength=w.n_loci)
    for li in np.flatnonzero(n):
        gs = float(npos[li]) / float(n[li])
        qs = spos[li] / npos[li] if npos[li] else 0.0
        idx, gq = finish._gt_from_counts(int(n[li]), int(nonpos_r[li]))
        if idx == 0 and gs > .15:
            idx = 1
        out[li] = (qs, gs, idx, gq, n[li])
    return out


# ------------------------------------------------------------------------------------------
# BASELINE.json configs[3] / configs[4] as whole-CLI worlds (tools/run_at_size.py, bench.py's strong-scaling record)
# ------------------------------------------------------------------------------------------
# the complex records of a VCF that reach a scorer in the reference (the defects it dies of - str > int for a DISDUP with
# spanning reads, SF:1801; blocks >= 100 bp apart, SF:1585 - are pinned by tests/golden/locus_complex.json.gz, not run here)
CX = [dict(type="DUP_INV", a=200, gap=1400), dict(type="DUP_INV", a=260, gap=1900), dict(type="DUP_INV", a=180, gap=-1300),
      dict(type="DUP_INV", a=320, gap=10400), dict(type="DUP_INV", a=280, xchrom=1000),
      dict(type="DISDUP", a=300, gap=11000), dict(type="DISDUP", a=260, xchrom=900), dict(type="DISDUP", a=260, xchrom=15000),
      dict(type="DEL_INV", a=300, b=420), dict(type="DEL_INV", a=350, b=300, order="inv,del"), dict(type="DEL_INV", a=700, b=9600),
      dict(type="OTHER", a=420, b=520, other=("ab/ab", "b/b^")), dict(type="OTHER", a=380, b=460, other=("ab/ab", "a/ab")),
      dict(type="OTHER", a=300, b=2400, other=("ab/ab", "aba/ab")), dict(type="OTHER", a=400, b=500, other=("ab/ab", "ba^/ab"))]

AT_SIZE = {
    "cfg4": dict(mode="vcf", loci=10000, base=300, n_reads=40, read_len=15000, seed=404),
    "cfg5": dict(mode="bed", loci=50000, base=250, n_reads=60, read_len=30000, seed=505),
}


def at_size_base_world(cfg, base, span_dist=None):
    """The distinct loci of an at-size world: cfg5 simple types for `vapor bed`; cfg4 two thirds simple types and a third
    complex records for `vapor vcf`.  Returns (simple world, complex world or None).
    span_dist = "simulate": the simple types' spans follow the reference's simulated truth sets (synth.simulate_span_tables:
    50 bp - 100 kb, median ~2.8 kb, 7-10 % of the deletions and inversions >= 10 kb - the drivers' junction-window branch,
    SF:1706 / 1728) instead of the uniform 50 bp - 11 kb of rounds 3 and 4."""
    sp = AT_SIZE[cfg]
    if cfg == "cfg5":
        w = synth.make_world(seed=sp["seed"], n_loci=base, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), span_range=(50, 11000),
                             read_len=sp["read_len"], n_reads=sp["n_reads"], span_dist=span_dist)
        return w, None
    n_cx = base // 3                                         # a third complex records
    w = synth.make_world(seed=sp["seed"], n_loci=base - n_cx, svtypes=("DEL", "INV", "INS", "DEL"), span_range=(60, 11000),
                         read_len=sp["read_len"], n_reads=sp["n_reads"], span_dist=span_dist)
    rng = np.random.default_rng(sp["seed"] + 1)
    specs = []
    for t in range(n_cx):
        s = dict(CX[t % len(CX)])
        s["a"] = int(s["a"] * rng.uniform(0.9, 1.3))         # sizes vary from record to record
        specs.append(s)
    cx = synth.make_complex_world(sp["seed"] + 2, specs, n_reads=sp["n_reads"])
    return w, cx


def at_size_tile(w, n_total, text_of):
    """Alias contigs `<name>_t<k>` sharing the base strings and record lists; returns the tiled world and its input text."""
    base_names = list(w.contigs)
    per = len(w.loci)
    big = synth.SynthWorld()
    lines = []
    k = 0
    while len(big.loci) < n_total:
        sub = synth.SynthWorld()
        for c in base_names:
            big.contigs["%s.t%d" % (c, k)] = w.contigs[c]
            big.reads["%s.t%d" % (c, k)] = w.reads.get(c, [])
        for l in w.loci[:n_total - len(big.loci)]:
            extra = dict(l.extra) if l.extra else None
            if extra and "insert_chrom" in extra:
                extra["insert_chrom"] = "%s.t%d" % (extra["insert_chrom"], k)
            sub.loci.append(synth.Locus("%s.t%d" % (l.chrom, k), l.svtype, l.start, l.end, "%s.t%d" % (l.svid, k), l.ins_seq, extra))
        big.loci += sub.loci
        lines.append(text_of(sub))
        k += 1
    return big, "".join(lines), per


def at_size_input(cfg: str, n_total: int, base: int, distinct: bool = False, cap: int = 2048, span_dist=None):
    """(world, input text, records) of an at-size run: the tiled world and its BED (cfg5) or header-less VCF (cfg4).
    distinct: every tile mutated on its own and made when it is reached (synth.DistinctTilesWorld): as many distinct loci as
    records, without the world ever lying in memory."""
    w, cx = at_size_base_world(cfg, base, span_dist)
    if distinct:
        return _at_size_distinct(cfg, n_total, w, cx, cap)
    if cfg == "cfg5":
        big, text, _ = at_size_tile(w, n_total, synth.bed_text)
    else:
        n_cx_total = n_total // 3
        b1, t1, _ = at_size_tile(w, n_total - n_cx_total, lambda s: synth.vcf_text(s, header=False))
        b2, t2, _ = at_size_tile(cx, n_cx_total, lambda s: synth.complex_vcf_text(s, header=False))
        big = b1
        big.contigs.update(b2.contigs); big.reads.update(b2.reads); big.loci += b2.loci
        text = t1 + t2
    return big, text, text.count("\n")


def _at_size_distinct(cfg, n_total, w, cx, cap):
    seed = AT_SIZE[cfg]["seed"]
    worlds = [(w, n_total if cfg == "cfg5" else n_total - n_total // 3, synth.bed_text if cfg == "cfg5" else (lambda s: synth.vcf_text(s, header=False)))]
    if cfg != "cfg5":
        worlds.append((cx, n_total // 3, lambda s: synth.complex_vcf_text(s, header=False)))
    big = synth.SynthWorld()
    parts = []
    texts = []
    for base_w, want, text_of in worlds:
        per = len(base_w.loci)
        n_tiles = -(-want // per)
        dw = synth.DistinctTilesWorld(base_w, n_tiles, seed, cap)
        parts.append(dw)
        left = want
        for k in range(n_tiles):
            sub = synth.SynthWorld()
            sub.loci = [dw.tile_locus(l, k) for l in base_w.loci[:left]]
            left -= len(sub.loci)
            big.loci += sub.loci
            texts.append(text_of(sub))
    if len(parts) == 1:
        parts[0].loci = big.loci
        return parts[0], "".join(texts), "".join(texts).count("\n")
    merged = _MergedWorld(parts)
    merged.loci = big.loci
    text = "".join(texts)
    return merged, text, text.count("\n")


class _MergedWorld(synth.SynthWorld):
    """Two lazily tiled worlds behind one set of contig names (cfg4: simple and complex records)."""
    cache_ok = False

    class _Both:
        def __init__(self, maps):
            self.maps = maps

        def __contains__(self, k):
            return any(k in m for m in self.maps)

        def __getitem__(self, k):
            for m in self.maps:
                if k in m:
                    return m[k]
            raise KeyError(k)

        def get(self, k, d=None):
            for m in self.maps:
                if k in m:
                    return m[k]
            return d

        def __len__(self):
            return sum(len(m) for m in self.maps)

    def __init__(self, parts):
        super().__init__()
        self.parts = parts
        self.contigs = self._Both([p.contigs for p in parts])
        self.reads = self._Both([p.reads for p in parts])

    def fai_rows(self):
        return [r for p in self.parts for r in p.fai_rows()]
