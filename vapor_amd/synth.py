"""Seeded synthetic genomes, SV loci and PacBio-like reads.

Everything here is the build's own generator: i.i.d. uniform ACGT contigs, SVs
implanted by string surgery, reads sampled from the ref or alt haplotype with
CLR-like errors (1 % sub / 8 % ins / 4 % del by default) and a CIGAR that is
exact up to the left edge of the SV (all the reference ever walks, see
vapor_vali/Simple_function.pyx:309-337).  The records are served to host code
through `vapor_amd.seqio.MemorySamtools`, which speaks the two samtools text
formats the reference parses (SF:339-354, SF:1203-1217).

No reference code is used or needed here.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = {i: None for i in range(256)}
_COMP.update({ord(a): b for a, b in zip("ACGTNacgtn", "TGCANtgcan")})


def random_dna(rng: np.random.Generator, n: int) -> str:
    return _ACGT[rng.integers(0, 4, size=n)].tobytes().decode("ascii")


def revcomp(seq: str) -> str:
    """Reverse complement over ACGTN/acgtn; other characters are dropped, as the
    reference's `complementary` does (SF:471-478)."""
    return seq.translate(_COMP)[::-1]


def _rle_cigar(ops: np.ndarray) -> str:
    # ops: uint8 array of b'M', b'I', b'D'
    if ops.size == 0:
        return "*"
    change = np.flatnonzero(ops[1:] != ops[:-1]) + 1
    starts = np.concatenate(([0], change))
    ends = np.concatenate((change, [ops.size]))
    # "<run length><op>" for every run, formatted by numpy rather than one str.__mod__ per run
    return "".join(np.char.add((ends - starts).astype("U"), ops[starts].view("S1").astype("U")).tolist())


def mutate(rng: np.random.Generator, seg: str, sub: float = 0.01, ins: float = 0.08,
           dele: float = 0.04, cigar: bool = True) -> Tuple[str, str]:
    """Apply CLR-like errors to `seg`; returns (read, cigar relative to seg; "" when `cigar` is off - the random
    stream is the same either way).

    The first base is always kept as a match so that POS is the first aligned base."""
    n = len(seg)
    if n == 0:
        return "", "*"
    b = np.frombuffer(seg.encode("ascii"), dtype=np.uint8)
    u = rng.random(n)
    is_del = u < dele
    is_sub = (u >= dele) & (u < dele + sub)
    is_ins = rng.random(n) < ins
    is_del[0] = False
    is_sub[0] = False
    # substituted bases: rotate within ACGT when the base is ACGT, else keep
    code = np.full(256, 255, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    c = code[b]
    rot = rng.integers(1, 4, size=n).astype(np.uint8)
    subbed = np.where((c < 4) & is_sub, _ACGT[(c + rot) & 3], b)
    ins_b = _ACGT[rng.integers(0, 4, size=n)]
    keep = ~is_del
    # output layout: for each position p: [base if kept][inserted base if is_ins]
    cnt = keep.astype(np.int64) + is_ins.astype(np.int64)
    off = np.concatenate(([0], np.cumsum(cnt)))
    out = np.empty(off[-1], dtype=np.uint8)
    out[off[:-1][keep]] = subbed[keep]
    ins_pos = off[:-1] + keep.astype(np.int64)
    out[ins_pos[is_ins]] = ins_b[is_ins]
    # cigar ops: per position M or D, then I
    if not cigar:
        return out.tobytes().decode("ascii"), ""
    opcnt = 1 + is_ins.astype(np.int64)
    ooff = np.concatenate(([0], np.cumsum(opcnt)))
    ops = np.empty(ooff[-1], dtype=np.uint8)
    ops[ooff[:-1]] = np.where(keep, ord("M"), ord("D"))
    ops[(ooff[:-1] + 1)[is_ins]] = ord("I")
    return out.tobytes().decode("ascii"), _rle_cigar(ops)


@dataclasses.dataclass
class SamRecord:
    qname: str
    rname: str
    pos: int          # 1-based leftmost aligned base
    cigar: str
    seq: str
    ref_span: int     # reference bases the alignment is taken to cover (for region overlap)

    def line(self) -> str:
        return "\t".join([self.qname, "0", self.rname, str(self.pos), "60", self.cigar,
                          "*", "0", "0", self.seq, "*"])


@dataclasses.dataclass
class Locus:
    """One SV call in a private contig (0-based half-open internals, 1-based text outside)."""
    chrom: str
    svtype: str                   # DEL | TANDUP | INV | INS | DISDUP | DUP_INV | DEL_INV
    start: int                    # as written to BED/VCF column 2
    end: int                      # column 3
    svid: str
    ins_seq: Optional[str] = None
    extra: Optional[dict] = None  # insert_point etc. for the complex types


class SynthWorld:
    """Contigs + aligned reads, addressable the way samtools addresses them."""

    def __init__(self) -> None:
        self.contigs: Dict[str, str] = {}
        self.reads: Dict[str, List[SamRecord]] = {}
        self.loci: List[Locus] = []

    # -- samtools-like accessors -------------------------------------------------
    def fetch(self, chrom: str, start: int, end: int) -> str:
        """1-based inclusive, clipped to the contig like `samtools faidx`."""
        s = self.contigs[chrom]
        start = max(int(start), 1)
        end = min(int(end), len(s))
        if end < start:
            return ""
        return s[start - 1:end]

    def overlapping(self, chrom: str, start: int, end: int) -> List[SamRecord]:
        out = []
        for r in self.reads.get(chrom, ()):
            if r.pos <= end and r.pos + r.ref_span - 1 >= start:
                out.append(r)
        return out


def apply_sv(ref: str, svtype: str, s: int, e: int, ins_seq: Optional[str] = None,
             ins_point: Optional[int] = None) -> str:
    """Alt haplotype of a whole contig. `s`,`e` are the BED columns; the SV block is
    contig[s:e] in 0-based half-open terms (the reference treats faidx s..e loosely;
    the generator only has to be self-consistent)."""
    blk = ref[s:e]
    if svtype == "DEL":
        return ref[:s] + ref[e:]
    if svtype in ("TANDUP", "DUP"):
        return ref[:e] + blk + ref[e:]
    if svtype == "INV":
        return ref[:s] + revcomp(blk) + ref[e:]
    if svtype == "INS":
        return ref[:s] + (ins_seq or "") + ref[s:]
    if svtype == "DISDUP":
        p = int(ins_point)
        return ref[:p] + blk + ref[p:]
    if svtype == "DUP_INV":
        p = int(ins_point)
        return ref[:p] + revcomp(blk) + ref[p:]
    if svtype == "DEL_INV":
        # delete [s, m) and invert [m, e), m carried in ins_point
        m = int(ins_point)
        return ref[:s] + revcomp(ref[m:e]) + ref[e:]
    raise ValueError(svtype)


_SPAN_TABLES = None


def simulate_span_tables() -> dict:
    """The SV span distribution of the reference's simulated truth sets (simulate/Structural_Variants_het: spans 50 bp - 100 kb,
    median ~2.8 kb, 7-10 % of the deletions and inversions >= 10 kb; tandem duplications below 5 kb; insertion lengths from the
    element names) as quantile tables - vapor_amd/data/simulate_spans.json, written by oracle/gen_span_dist.py in the build
    container (data only)."""
    global _SPAN_TABLES
    if _SPAN_TABLES is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "simulate_spans.json")) as f:
            _SPAN_TABLES = json.load(f)
    return _SPAN_TABLES


def draw_span(rng, table: dict) -> int:
    """One value from a quantile table: inverse CDF, linear between quantiles."""
    q = table["quantiles"]
    x = float(rng.random()) * (len(q) - 1)
    lo = min(int(x), len(q) - 2)
    return int(round(q[lo] + (q[lo + 1] - q[lo]) * (x - lo)))


def make_world(seed: int, n_loci: int, svtypes: Sequence[str] = ("DEL", "TANDUP"),
               span_range: Tuple[int, int] = (200, 3000), read_len: int = 6000,
               n_reads: int = 12, alt_fraction: float = 0.5, lead: int = 300,
               contig_pad: int = 2000, errors: Tuple[float, float, float] = (0.01, 0.08, 0.04),
               chrom_prefix: str = "c", ins_len_range: Tuple[int, int] = (100, 600), span_dist: str = None,
               span_max: int = 100000, spans: Sequence[int] = None) -> SynthWorld:
    """Build `n_loci` independent loci, one contig each.

    Reads start 1..`lead` bases left of the scored window's left edge (SV start minus the
    500 bp flank) so that the reference's POS<=start filter keeps them, and are
    `read_len` haplotype bases long before errors.
    span_dist = "simulate": spans (and insertion lengths) are drawn from the distribution of the reference's simulated truth
    sets per SV type (simulate_span_tables) instead of uniformly from `span_range`; a span above `span_max` is drawn again."""
    rng = np.random.default_rng(seed)
    w = SynthWorld()
    tables = simulate_span_tables() if span_dist == "simulate" else None
    if span_dist not in (None, "simulate"):
        raise ValueError("span_dist: None or 'simulate'")
    for li in range(n_loci):
        svtype = svtypes[li % len(svtypes)]
        if spans is not None:
            span = int(spans[li % len(spans)])             # (given locus by locus)
        elif tables is not None:
            tb = tables["simple"].get(svtype) or tables["simple"]["DEL"]
            span = draw_span(rng, tb)
            while span > span_max or span < 1:
                span = draw_span(rng, tb)
        else:
            span = int(rng.integers(span_range[0], span_range[1] + 1))
        flank = min(500, span)
        left = flank + lead + 50
        # (a span the drivers score by its junction windows only - 10 kb and more, SF:1706 - needs no room for the alt
        # haplotype's doubled block behind it)
        clen = left + (3 if (span < 10000 or svtype not in ("DEL", "INV")) else 1) * span + read_len + contig_pad
        chrom = "%s%d" % (chrom_prefix, li + 1)
        ref = random_dna(rng, clen)
        s = left
        e = s + span
        ins_seq = None
        ins_point = None
        extra = None
        if svtype == "INS":
            ilen = draw_span(rng, tables["insertion_length"]) if tables is not None else int(rng.integers(ins_len_range[0], ins_len_range[1] + 1))
            ilen = max(ilen, 1)
            ins_seq = random_dna(rng, ilen)
            e = s + 1
            flank = min(500, ilen)
        elif svtype in ("DISDUP", "DUP_INV"):
            ins_point = e + int(rng.integers(50, max(51, span)))
            extra = {"insert_point": ins_point}
        elif svtype == "DEL_INV":
            ins_point = s + span // 2
            extra = {"mid": ins_point}
        alt = apply_sv(ref, svtype, s, e, ins_seq, ins_point)
        w.contigs[chrom] = ref
        recs: List[SamRecord] = []
        win_left = s - flank  # 1-based coordinate the reference uses as window start
        for ri in range(n_reads):
            from_alt = rng.random() < alt_fraction
            hap = alt if from_alt else ref
            a = win_left - 1 - int(rng.integers(1, lead + 1))
            a = max(a, 0)
            b = min(a + read_len, len(hap))
            read, cigar = mutate(rng, hap[a:b], *errors)
            recs.append(SamRecord("r%d_%d%s" % (li + 1, ri, "a" if from_alt else "r"), chrom,
                                  a + 1, cigar, read, b - a))
        w.reads[chrom] = recs
        w.loci.append(Locus(chrom, svtype, s, e, "sv%d" % (li + 1), ins_seq, extra))
    return w


def bed_text(world: SynthWorld) -> str:
    """5+ column BED as `bed_info_readin` expects it today (vapor_vali/vapor:22-50)."""
    rows = []
    for l in world.loci:
        t = {"TANDUP": "DUP"}.get(l.svtype, l.svtype)
        if l.svtype == "INS":
            rows.append("\t".join([l.chrom, str(l.start), str(l.end), l.svid, "INS", l.ins_seq]))
        elif l.svtype in ("DEL", "TANDUP", "INV"):
            rows.append("\t".join([l.chrom, str(l.start), str(l.end), l.svid, t]))
    return "\n".join(rows) + "\n"


def vcf_text(world: SynthWorld, header: bool = True) -> str:
    """Minimal VCF with the INFO keys `vcf_list_readin` reads (vapor_vali/vapor:127-202,
    README.md:79-82 for the complex types)."""
    out = ["##fileformat=VCFv4.1", "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of SV\">",
           "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End\">", "##source=vapor_amd.synth",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"] if header else []
    for l in world.loci:
        info = "SVTYPE=%s;END=%d" % ({"TANDUP": "DUP"}.get(l.svtype, l.svtype), l.end)
        alt = "<%s>" % l.svtype
        if l.svtype == "INS":
            info = "SVTYPE=INS;END=%d;SVLEN=%d;SEQ=%s" % (l.end, len(l.ins_seq), l.ins_seq)
            alt = "<INS>"
        elif l.svtype in ("DISDUP", "DUP_INV"):
            info += ";insert_point=%s:%d" % (l.chrom, l.extra["insert_point"])
        elif l.svtype == "DEL_INV":
            m = l.extra["mid"]
            info += ";del=%s:%d-%d;inv=%s:%d-%d" % (l.chrom, l.start, m, l.chrom, m, l.end)
        out.append("\t".join([l.chrom, str(l.start), l.svid, "N", alt, ".", "PASS", info,
                              "GT", "0/1"]))
    return "\n".join(out) + "\n"


# ---------------------------------------------------------------------------
# kernel-level synthetic shapes (bench / parity at BASELINE sizes)
# ---------------------------------------------------------------------------

def make_pairs(seed: int, n_alleles: int, reads_per_allele: int, read_len: int, allele_len: int,
               errors: Tuple[float, float, float] = (0.01, 0.08, 0.04),
               sv: bool = True) -> Tuple[List[str], List[str], List[Tuple[int, int]]]:
    """`n_alleles` windows of `allele_len` bases, each with `reads_per_allele` noisy reads of
    ~`read_len` bases drawn from inside the window (half of the alleles carry a deletion
    relative to the haplotype the reads come from when `sv`), returning
    (alleles, reads, [(read_idx, allele_idx)...])."""
    rng = np.random.default_rng(seed)
    alleles: List[str] = []
    reads: List[str] = []
    pairs: List[Tuple[int, int]] = []
    for ai in range(n_alleles):
        a = random_dna(rng, allele_len)
        alleles.append(a)
        hap = a
        if sv and (ai & 1):
            cut = int(rng.integers(allele_len // 4, allele_len // 2))
            ln = int(rng.integers(50, 2000))
            hap = a[:cut] + a[cut + ln:]
        for _ in range(reads_per_allele):
            st = int(rng.integers(0, max(1, len(hap) - read_len)))
            r, _c = mutate(rng, hap[st:st + read_len], *errors)
            reads.append(r[:read_len])
            pairs.append((len(reads) - 1, ai))
    return alleles, reads, pairs


def world_from_json(d: dict) -> SynthWorld:
    """Inverse of the fixture layout written by oracle/gen_golden.py (world_to_json)."""
    w = SynthWorld()
    w.contigs = dict(d["contigs"])
    w.reads = {c: [SamRecord(q, c, pos, cig, seq, span) for q, pos, cig, seq, span in rs]
               for c, rs in d["reads"].items()}
    w.loci = [Locus(*l) for l in d["loci"]]
    return w


class VirtualContig:
    """A chromosome-sized i.i.d. ACGT sequence that is never materialised: 4 kb blocks are generated
    on demand from (seed, block index).  Supports len() and slicing like a str."""
    BLOCK = 4096

    def __init__(self, seed: int, length: int):
        self.seed, self.length = int(seed), int(length)
        self._cache: Dict[int, str] = {}

    def __len__(self) -> int:
        return self.length

    def _block(self, b: int) -> str:
        s = self._cache.get(b)
        if s is None:
            s = random_dna(np.random.default_rng([self.seed, b]), self.BLOCK)
            if len(self._cache) > 4096:
                self._cache.clear()
            self._cache[b] = s
        return s

    def __getitem__(self, sl) -> str:
        if not isinstance(sl, slice):
            raise TypeError("VirtualContig supports slices only")
        a, b, _ = sl.indices(self.length)
        if b <= a:
            return ""
        parts = [self._block(k) for k in range(a // self.BLOCK, (b - 1) // self.BLOCK + 1)]
        s = "".join(parts)
        off = a - (a // self.BLOCK) * self.BLOCK
        return s[off:off + (b - a)]


def make_world_from_bed(rows: Sequence[Sequence], seed: int, contig_len: int = 135534747, n_reads: int = 10,
                        alt_fraction: float = 0.5, lead: int = 250,
                        errors: Tuple[float, float, float] = (0.01, 0.08, 0.04)) -> SynthWorld:
    """Loci at given genome coordinates (rows of chrom, start, end, TYPE) on virtual contigs, with reads
    sampled around every locus - BASELINE.json configs[0]: the reference's vapor_test.bed needs a BAM
    and hg19 that are not bundled, so the plumbing runs on a stand-in genome."""
    rng = np.random.default_rng(seed)
    w = SynthWorld()
    for li, row in enumerate(rows):
        chrom, s, e, svtype = row[0], int(row[1]), int(row[2]), {"DUP": "TANDUP"}.get(row[3], row[3])
        if chrom not in w.contigs:
            w.contigs[chrom] = VirtualContig(seed * 1000003 + len(w.contigs), contig_len)
            w.reads[chrom] = []
        span = e - s
        flank = min(500, span)
        r0 = s - flank - lead - 50                       # 0-based start of the local window
        read_len = 2 * flank + 2 * span + lead + 400
        local = w.contigs[chrom][r0:r0 + read_len + 3 * span + 2000]
        ls, le = s - r0, e - r0
        alt = apply_sv(local, svtype, ls, le)
        for ri in range(n_reads):
            from_alt = rng.random() < alt_fraction
            hap = alt if from_alt else local
            a = (s - flank) - r0 - 1 - int(rng.integers(1, lead + 1))
            b = min(a + read_len, len(hap))
            read, cigar = mutate(rng, hap[a:b], *errors)
            w.reads[chrom].append(SamRecord("v%d_%d%s" % (li + 1, ri, "a" if from_alt else "r"), chrom,
                                            r0 + a + 1, cigar, read, b - a))
        w.loci.append(Locus(chrom, svtype, s, e, "sv%d" % (li + 1)))
    return w


# ---------------------------------------------------------------------------
# complex SV types of the VCF path (BASELINE.json configs[3]): DISDUP, DUP_INV, DEL_INV, Other=
# ---------------------------------------------------------------------------

def _add_reads(rng, w: SynthWorld, chrom: str, hap_ref: str, hap_alt: str, anchor: int, read_len: int, n_reads: int,
               alt_fraction: float, lead: int, errors, tag: str) -> None:
    """`n_reads` reads that start 1..`lead` bases left of the 1-based window start `anchor`."""
    recs = w.reads.setdefault(chrom, [])
    for ri in range(n_reads):
        from_alt = rng.random() < alt_fraction
        hap = hap_alt if from_alt else hap_ref
        a = max(anchor - 1 - int(rng.integers(1, lead + 1)), 0)
        b = min(a + read_len, len(hap))
        read, cigar = mutate(rng, hap[a:b], *errors)
        recs.append(SamRecord("%s_%d%s" % (tag, ri, "a" if from_alt else "r"), chrom, a + 1, cigar, read, b - a))


def make_complex_world(seed: int, specs: Sequence[dict], n_reads: int = 8, alt_fraction: float = 0.5, lead: int = 250,
                       errors: Tuple[float, float, float] = (0.01, 0.08, 0.04), chrom_prefix: str = "x") -> SynthWorld:
    """One private contig (two for an inter-contig duplication) per spec.  Spec keys:
      type   DUP_INV | DISDUP | DEL_INV | OTHER
      a      length of the duplicated / first block
      gap    DUP_INV / DISDUP: distance from the block's end to the insert point (negative: the insert point lies
             that far left of the block's start); `inside`: insert point this far inside the block
      b      DEL_INV: length of the second block; OTHER: length of block b
      order  DEL_INV: "del,inv" or "inv,del"; `apart`: bases between the two blocks (the reference wants < 100)
      other  OTHER: (ref structure, alt structure), e.g. ("ab/ab", "b/b^")
      xchrom DISDUP / DUP_INV: insert point on a contig of its own, at this coordinate
      n_reads / alt_fraction override the defaults.
    Duplicated blocks are kept short against the distance to their copy so that the self dot plot of the alt
    window stays below the 10 % lower-triangle share at which the reference starts its unseeded X-means
    (qual_check_repetitive_region, SF:1165)."""
    rng = np.random.default_rng(seed)
    w = SynthWorld()
    for li, sp in enumerate(specs):
        t = sp["type"]
        chrom = "%s%d" % (chrom_prefix, li + 1)
        a = int(sp["a"])
        nr = int(sp.get("n_reads", n_reads))
        af = float(sp.get("alt_fraction", alt_fraction))
        left = 500 + lead + 80
        tag = "q%d" % (li + 1)
        if t in ("DUP_INV", "DISDUP"):
            s, e = left, left + a
            flank = min(500, a)
            rcb = t == "DUP_INV"
            if "xchrom" in sp:
                # the copy lands on another contig
                p = int(sp["xchrom"])
                ref = random_dna(rng, e + 3000)
                xc = chrom + "i"
                xref = random_dna(rng, p + a + 4000)
                blk = ref[s:e]
                xalt = xref[:p] + (revcomp(blk) if rcb else blk) + xref[p:]
                w.contigs[chrom] = ref
                w.contigs[xc] = xref
                w.reads.setdefault(chrom, [])
                _add_reads(rng, w, xc, xref, xalt, p - flank, 2 * flank + a + lead + 600, nr, af, lead, errors, tag)
                w.loci.append(Locus(chrom, t, s, e, "cx%d" % (li + 1), None, {"insert_point": p, "insert_chrom": xc}))
                continue
            gap = int(sp.get("gap", 1200))
            if sp.get("inside"):
                p = s + int(sp["inside"])
            elif gap >= 0:
                p = e + gap
            else:
                left = left - gap
                s, e = left, left + a
                p = s + gap
            hi = max(e, p)
            ref = random_dna(rng, hi + 2 * a + 3000 + lead)
            blk = ref[s:e]
            alt = ref[:p] + (revcomp(blk) if rcb else blk) + ref[p:]
            w.contigs[chrom] = ref
            far = hi - min(s, p) >= 10000
            if far:
                anchor, rl = p - flank, 2 * flank + a + lead + 600
            else:
                anchor, rl = min(s, p) - flank, (hi - min(s, p)) + a + 2 * flank + lead + 500
            _add_reads(rng, w, chrom, ref, alt, anchor, rl, nr, af, lead, errors, tag)
            w.loci.append(Locus(chrom, t, s, e, "cx%d" % (li + 1), None, {"insert_point": p, "insert_chrom": chrom}))
        elif t == "DEL_INV":
            b = int(sp["b"])
            apart = int(sp.get("apart", 0))
            s = left
            m1 = s + a                    # end of the first block
            m0 = m1 + apart               # start of the second
            e = m0 + b
            ref = random_dna(rng, e + a + b + 3000 + lead)
            first, second = sp.get("order", "del,inv").split(",")
            blocks = [(s, m1, first), (m0, e, second)]
            def piece(x0, x1, kind):
                return revcomp(ref[x0:x1]) if kind == "inv" else ""

            alt = ref[:s] + piece(*blocks[0]) + ref[m1:m0] + piece(*blocks[1])
            alt += ref[e:]
            w.contigs[chrom] = ref
            flank = min(500, e - s)
            if e - s >= 10000:
                anchor, rl = s - flank, 2 * flank + lead + 700
            else:
                anchor, rl = s - flank, (e - s) + 2 * flank + lead + 500
            _add_reads(rng, w, chrom, ref, alt, anchor, rl, nr, af, lead, errors, tag)
            w.loci.append(Locus(chrom, t, s, e, "cx%d" % (li + 1), None,
                                {"blocks": [[x0, x1, kind] for (x0, x1, kind) in blocks]}))
        elif t == "OTHER":
            b = int(sp["b"])
            s = left
            m = s + a
            e = m + b
            ref = random_dna(rng, e + a + b + 3000 + lead)
            ref_s, alt_s = sp["other"]
            blk = {"a": ref[s:m], "b": ref[m:e]}
            alleles = [x for x in alt_s.split("/") if x not in ref_s.split("/")] or [ref_s.split("/")[0]]
            haps = []
            for al in alleles:
                mid, prev = "", None
                for ch in al:
                    if ch == "^":
                        mid = mid[:len(mid) - len(blk[prev])] + revcomp(blk[prev])
                    else:
                        mid += blk[ch]
                        prev = ch
                haps.append(ref[:s] + mid + ref[e:])
            w.contigs[chrom] = ref
            flank = min(500, e - s)
            rl = 2 * (e - s) + 2 * flank + lead + 500
            recs = w.reads.setdefault(chrom, [])
            for ri in range(nr):
                from_alt = rng.random() < af
                hap = haps[ri % len(haps)] if from_alt else ref
                a0 = max(s - flank - 1 - int(rng.integers(1, lead + 1)), 0)
                b0 = min(a0 + rl, len(hap))
                read, cigar = mutate(rng, hap[a0:b0], *errors)
                recs.append(SamRecord("%s_%d%s" % (tag, ri, "a" if from_alt else "r"), chrom, a0 + 1, cigar, read, b0 - a0))
            w.loci.append(Locus(chrom, t, s, e, "cx%d" % (li + 1), None, {"other": [ref_s, alt_s], "bps": [s, m, e]}))
        else:
            raise ValueError(t)
    return w


def complex_vcf_text(world: SynthWorld, header: bool = False) -> str:
    """VCF records for make_complex_world's loci with the INFO keys vapor_vali/vapor:87-125, 176-202 read:
    insert_point=chrom:pos (DISDUP, DUP_INV), del=/inv= (DEL_INV), Other=ref_alt_chrom:bp:bp:bp."""
    out = ["##fileformat=VCFv4.1", "##source=vapor_amd.synth",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"] if header else []
    for l in world.loci:
        if l.svtype in ("DISDUP", "DUP_INV"):
            info = "SVTYPE=%s;END=%d;insert_point=%s:%d" % (l.svtype, l.end, l.extra["insert_chrom"], l.extra["insert_point"])
        elif l.svtype == "DEL_INV":
            info = "SVTYPE=DEL_INV;END=%d;" % l.end + ";".join("%s=%s:%d-%d" % (k, l.chrom, x0, x1) for x0, x1, k in l.extra["blocks"])
        elif l.svtype == "OTHER":
            r, a = l.extra["other"]
            info = "SVTYPE=CPLX;END=%d;Other=%s_%s_%s:%s" % (l.end, r, a, l.chrom, ":".join(str(x) for x in l.extra["bps"]))
        else:
            raise ValueError(l.svtype)
        out.append("\t".join([l.chrom, str(l.start), l.svid, "N", "<%s>" % l.svtype, ".", "PASS", info, "GT", "0/1"]))
    return "\n".join(out) + "\n"


def write_world_files(world: SynthWorld, directory: str, block_size: int = 8192, qual_seed=None) -> Tuple[str, str]:
    """FASTA + .fai and coordinate-sorted BAM + .bai of a synthetic world, written by this package alone
    (vapor_amd.bamio); returns (fasta path, bam path).  Reads keep their order inside one start position."""
    import os
    from . import bamio
    fa = os.path.join(directory, "ref.fa")
    names = list(world.contigs)
    with open(fa, "w") as f, open(fa + ".fai", "w") as fi:
        off = 0
        for n in names:
            seq = world.contigs[n]
            seq = seq[0:len(seq)] if not isinstance(seq, str) else seq
            hdr = ">" + n + "\n"
            f.write(hdr)
            off += len(hdr)
            fi.write("%s\t%d\t%d\t60\t61\n" % (n, len(seq), off))
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
            off += len(seq) + (len(seq) + 59) // 60
    recs = [(r.qname, names.index(c), r.pos - 1, r.cigar, r.seq) for c, rs in world.reads.items() for r in rs]
    bam = os.path.join(directory, "reads.bam")
    bamio.write_bam(bam, [(n, len(world.contigs[n])) for n in names], recs, block_size=block_size, qual_seed=qual_seed)
    return fa, bam


# ---------------------------------------------------------------------------
# at-size worlds of DISTINCT loci without holding them: tiles of a base world, every tile mutated on its own
# ---------------------------------------------------------------------------
_NEXT_BASE = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"CGTAcgta"):
    _NEXT_BASE[_a] = _b


def _mutated(text: str, rng, rate: float = 0.002) -> str:
    """`text` with one base in five hundred replaced by the next one of A -> C -> G -> T -> A (case kept, other symbols
    kept): another sequence of the same length and structure."""
    a = np.frombuffer(text.encode("ascii"), dtype=np.uint8).copy()
    if len(a):
        pos = rng.integers(0, len(a), size=max(1, int(len(a) * rate)))
        a[pos] = _NEXT_BASE[a[pos]]
    return a.tobytes().decode("ascii")


class _LazyTiles:
    """Mapping `<base name>.t<k>` -> the base world's entry, mutated for tile k when it is asked for; the last `cap` answers are
    kept (a chunk's loci are neighbours).  Stands in for SynthWorld.contigs / .reads."""

    def __init__(self, base: dict, n_tiles: int, make, cap: int):
        from collections import OrderedDict
        self.base, self.n_tiles, self.make, self.cap = base, n_tiles, make, cap
        self.lru = OrderedDict()
        import threading
        self.lock = threading.Lock()
        self.made = 0
        self.seconds = 0.0

    def _split(self, name):
        b, _, k = name.rpartition(".t")
        return (b, int(k)) if b in self.base and k.isdigit() and int(k) < self.n_tiles else (None, -1)

    def __contains__(self, name):
        return self._split(name)[0] is not None

    def __getitem__(self, name):
        with self.lock:
            got = self.lru.get(name)
            if got is not None:
                self.lru.move_to_end(name)
                return got
        b, k = self._split(name)
        if b is None:
            raise KeyError(name)
        import time
        t0 = time.perf_counter()
        got = self.make(b, k)
        with self.lock:
            self.seconds += time.perf_counter() - t0
            self.made += 1
            self.lru[name] = got
            while len(self.lru) > self.cap:
                self.lru.popitem(last=False)
        return got

    def get(self, name, default=None):
        try:
            return self[name]
        except KeyError:
            return default

    def __len__(self):
        return len(self.base) * self.n_tiles


class DistinctTilesWorld(SynthWorld):
    """`n_tiles` copies of a base world under the contig names `<c>.t<k>`, every copy with its own substitutions in contigs,
    reads and insertion payloads (seeded by tile and contig): as many DISTINCT loci as records, generated when a locus is
    reached and dropped again, so that a 50 000-locus world of 30 kb reads never lies in memory (BASELINE configs[3], [4])."""
    cache_ok = False                   # (backends must not keep per-contig caches of this world: they would keep the world)

    def __init__(self, base: SynthWorld, n_tiles: int, seed: int, cap: int = 1024):
        super().__init__()
        import zlib
        self.base_world, self.n_tiles, self.seed = base, n_tiles, seed

        def rng_of(name, k, what):
            return np.random.default_rng([seed, k, zlib.crc32(name.encode()), what])

        def contig(b, k):
            return _mutated(base.contigs[b], rng_of(b, k, 0))

        def reads(b, k):
            rng = rng_of(b, k, 1)
            return [SamRecord(r.qname, "%s.t%d" % (r.rname, k), r.pos, r.cigar, _mutated(r.seq, rng), r.ref_span) for r in base.reads.get(b, [])]
        self.contigs = _LazyTiles(base.contigs, n_tiles, contig, cap)
        self.reads = _LazyTiles(base.reads, n_tiles, reads, cap)
        self._rng_of = rng_of

    def tile_locus(self, l: Locus, k: int) -> Locus:
        extra = dict(l.extra) if l.extra else None
        if extra and "insert_chrom" in extra:
            extra["insert_chrom"] = "%s.t%d" % (extra["insert_chrom"], k)
        ins = _mutated(l.ins_seq, self._rng_of(l.chrom, k, 2)) if l.ins_seq and "X" not in l.ins_seq else l.ins_seq
        return Locus("%s.t%d" % (l.chrom, k), l.svtype, l.start, l.end, "%s.t%d" % (l.svid, k), ins, extra)

    def fai_rows(self):
        return [("%s.t%d" % (c, k), len(v)) for k in range(self.n_tiles) for c, v in self.base_world.contigs.items()]
