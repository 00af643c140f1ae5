"""Locus drivers: allele-string construction, read selection and scorer choice per SV type
(component #2; reference SF:1490-1933).

Each driver is a generator.  It does its host-side string work itself and *yields* the two
kinds of device work the reference does inline -

    Window(seq)                             -> window_size_refine(seq)            (SF:2030-2046)
    Score(kind, ref_seq, alt_seq, reads, k) -> per read x the drivers' own reduction of
                                               calcu_vapor_single_read_score_*(ref, alt, x, k): the score
                                               1 - b/a, or None where the reference skips the read
                                               (`0 in [a, b]`, e.g. SF:1913); 'del' takes the smaller of the
                                               abs_dis and within_10Perc scores (SF:1718-1726)

- and receives the results back through send().  Run one generator at a time
(`pipeline.run_sync`) and it behaves like the reference's function of the same name; run many
in lockstep (`pipeline.run_batch`) and the pending requests of all loci go to the GPU as one
batch.  The control flow, fallbacks and quirks of the reference are kept, including the places
where it raises (cited inline).
"""
from __future__ import annotations

from typing import List

from . import seqio

default_flank_length = 500       # SF:21-22
default_max_sv_test = 10000      # SF:25-26


class Window:
    __slots__ = ("seq",)

    def __init__(self, seq: str):
        self.seq = seq


class Score:
    """kind: 's1' abs_dis_m1b, 's2' within_10Perc_m1b, 's3' directed_dis_m1b_redefine_diagnal,
    'del' = s1 and s2 for the same reads.  Result: one score (float) or None per read."""
    __slots__ = ("kind", "ref_seq", "alt_seq", "reads", "k")

    def __init__(self, kind, ref_seq, alt_seq, reads, k):
        self.kind, self.ref_seq, self.alt_seq, self.reads, self.k = kind, ref_seq, alt_seq, reads, k


class Figure:
    """make_event_figure_1 (SF:1072-1089) request; executors may render it or drop it."""
    __slots__ = ("scores", "best_read", "k", "ref_seq", "alt_seq", "name")

    def __init__(self, scores, best_read, k, ref_seq, alt_seq, name):
        self.scores, self.best_read, self.k = scores, best_read, k
        self.ref_seq, self.alt_seq, self.name = ref_seq, alt_seq, name


def _collect(results, reads, scores: List[float], keep=None):
    """The per-read loop every driver repeats (e.g. SF:1909-1915): a read counts when neither scorer output is 0
    (`keep`: a flag per read, for the reads that were handed to the executor at all; the executor
    hands such a read over as None, the others as 1 - b/a; the scorers never produce NaN, so
    vapor_dup_inv_VapoR's extra isnan test, SF:1629, changes nothing); returns the read with the best score so
    far (ties: the later read)."""
    best = ""
    it = iter(results)
    top = max(scores) if scores else None
    for t, x in enumerate(reads):
        if keep is not None and not keep[t]:
            continue
        s = next(it)
        if s is None:
            continue
        scores.append(s)
        if top is None or s >= top:          # (`scores[-1] == max(scores)` of the reference, without the scan)
            top = s
            best = x
    return best


def _window(seq):
    res = yield Window(seq)
    return res[0]


def _rc(seq: str) -> str:
    return seqio.reverse(seqio.complementary(seq))


class Allele(str):
    """An allele string that remembers how the driver built it: `segs` = [(parent string, off, len, revcomp), ...] - slices of
    windows the driver has read (and of an insertion's sequence), in order.  It IS the string (every consumer that wants
    text - figures, the window check, the reference-named functions - sees a str); the executors hand the segments to the
    library instead of the bytes (include/vapor_hip.h, vapor_seqset_create_derived): the device assembles the allele from
    the window it already has, and a read is joined once against the window and the alleles derived from it.
    `segs` is None when the text is not a concatenation of slices (complementary() drops characters outside ATGCN / atgcn,
    SF:471-478: a reversed slice that lost one is uploaded as bytes)."""
    segs = None


def _cat(*parts) -> Allele:
    """''.join of the parts, each (s, a, b) = s[a:b] or (s, a, b, True) = reverse(complementary(s[a:b])) with Python's slice
    rules (negative and None bounds), as an Allele that knows its segments."""
    text, segs = [], []
    for part in parts:
        src, a, b = part[0], part[1], part[2]
        i0, i1, _ = slice(a, b).indices(len(src))
        n = max(0, i1 - i0)
        piece = src[i0:i1]
        rc = len(part) > 3 and part[3]
        if rc:
            piece = _rc(piece)
            if len(piece) != n:
                segs = None                 # complementary() dropped something: not a slice of anything any more
        text.append(piece)
        if segs is not None and n:
            base = getattr(src, "segs", None)
            if base is not None and not rc and len(base) == 1 and not base[0][3]:
                seg = (base[0][0], base[0][1] + i0, n, False)             # (a slice of a one-slice allele: of its parent)
            elif type(src) is str:
                seg = (src, i0, n, rc)
            else:
                segs = None
                continue
            last = segs[-1] if segs else None
            if last is not None and not seg[3] and not last[3] and last[0] is seg[0] and last[1] + last[2] == seg[1]:
                segs[-1] = (last[0], last[1], last[2] + n, False)         # (two forward slices that lie end to end: one slice)
            else:
                segs.append(seg)
    out = Allele("".join(text))
    out.segs = segs
    return out


def _within(window: str, w0: int, piece: str, a: int):
    """The _cat part for `piece` = the reference from coordinate `a` on, given a window that starts at coordinate `w0`: a
    slice of the window where the window really holds those bases (the usual case: the drivers fetch a block's bases again
    although they lie inside the window they already hold, e.g. SF:1809-1813), else the piece as text of its own."""
    off = a - w0
    if off >= 0 and len(piece) and window[off:off + len(piece)] == piece:
        return (window, off, off + len(piece))
    return (piece, None, None)


def _rcpart(part):
    return (part[0], part[1], part[2], True)


# ------------------------------------------------------------------------------------------
def vapor_simple_del(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_simple_del_Vapor, SF:1701-1745."""
    flank = seqio.flank_length_calculate(sv_info)
    scores: List[float] = []
    if sv_info[2] - sv_info[1] < default_max_sv_test:
        reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, sv_info, flank)
        if len(reads) > num_reads_cff:
            ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[2] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                alt_seq = _cat((ref_seq, None, flank), (ref_seq, -flank, None))      # ref_seq[:flank] + ref_seq[-flank:], SF:1712
                res = yield Score("del", ref_seq, alt_seq, reads, k)     # min of the two scorers' scores, SF:1718-1726
                best = _collect(res, reads, scores)
                yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    else:
        reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, sv_info, flank)
        if len(reads) > num_reads_cff:
            ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[1] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                left = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[1])
                alt_seq = _cat(_within(ref_seq, sv_info[1] - flank, left, sv_info[1] - flank),
                               (seqio.ref_seq_readin(ref, sv_info[0], sv_info[2], sv_info[2] + flank), None, None))
                k = yield from _window(alt_seq)
                if not k == "Error":
                    res = yield Score("s2", ref_seq, alt_seq, reads, k)
                    best = _collect(res, reads, scores)
                    yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_simple_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_simple_inv_Vapor, SF:1895-1933."""
    flank = seqio.flank_length_calculate(sv_info)
    scores: List[float] = []
    if sv_info[2] - sv_info[1] < default_max_sv_test:
        ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[2] + flank)
        k = yield from _window(ref_seq)
        if not k == "Error":
            # ref_seq[:flank] + reverse(complementary(ref_seq[flank:-flank])) + ref_seq[-flank:], SF:1907
            alt_seq = _cat((ref_seq, None, flank), (ref_seq, flank, -flank, True), (ref_seq, -flank, None))
            k = yield from _window(alt_seq)
            if not k == "Error":
                reads = seqio.simple_chop_pacbio_read_simple_short(bam_in, sv_info, flank)
                if len(reads) > num_reads_cff:
                    res = yield Score("s1", ref_seq, alt_seq, reads, k)
                    best = _collect(res, reads, scores)
                    yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
                    return scores
    ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[1] + flank)
    k = yield from _window(ref_seq)
    if not k == "Error":
        alt_seq = _cat((ref_seq, None, flank), (seqio.ref_seq_readin(ref, sv_info[0], sv_info[2] - flank, sv_info[2], "TRUE"), None, None))
        k = yield from _window(alt_seq)
        if not k == "Error":
            reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, sv_info, flank)
            if len(reads) > num_reads_cff:
                res = yield Score("s2", ref_seq, alt_seq, reads, k)
                best = _collect(res, reads, scores)
                yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_simple_tandup(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_simple_tandup_Vapor, SF:1747-1784."""
    flank = seqio.flank_length_calculate(sv_info)
    scores: List[float] = []
    if sv_info[2] - sv_info[1] < default_max_sv_test:
        ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[1] - flank, sv_info[2] + flank)
        k = yield from _window(ref_seq)
        if not k == "Error":
            # ref_seq[:flank] + mid + mid + ref_seq[-flank:] with mid = ref_seq[flank:-flank], SF:1755
            alt_seq = _cat((ref_seq, None, flank), (ref_seq, flank, -flank), (ref_seq, flank, -flank), (ref_seq, -flank, None))
            k = yield from _window(alt_seq)
            if not k == "Error":
                reads = seqio.simple_chop_pacbio_read_simple_short(
                    bam_in, sv_info[:2] + [sv_info[1] + 2 * (sv_info[2] - sv_info[1])], flank)
                if len(reads) > num_reads_cff:
                    res = yield Score("s3", ref_seq, alt_seq, reads, k)
                    best = _collect(res, reads, scores)
                    yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
                    return scores
    ref_seq = seqio.ref_seq_readin(ref, sv_info[0], sv_info[2] - flank, sv_info[2] + flank)
    k = yield from _window(ref_seq)
    if not k == "Error":
        left = seqio.ref_seq_readin(ref, sv_info[0], sv_info[2] - flank, sv_info[2])
        alt_seq = _cat(_within(ref_seq, sv_info[2] - flank, left, sv_info[2] - flank),
                       (seqio.ref_seq_readin(ref, sv_info[0], sv_info[1], sv_info[1] + flank), None, None))
        k = yield from _window(alt_seq)
        if not k == "Error":
            reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, [sv_info[0], sv_info[2]], flank)
            if len(reads) > num_reads_cff:
                res = yield Score("s2", ref_seq, alt_seq, reads, k)
                best = _collect(res, reads, scores)
                yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_simple_ins(num_reads_cff, plt_li, bam_in, ref, ins_pos, ins_seq, out_figure_name, POLARITY):
    """vapor_simple_ins_Vapor, SF:1856-1893.  ins_pos is 'chrom_pos'."""
    if POLARITY == "+":
        ins_seq_2 = ins_seq
    elif POLARITY == "-":
        ins_seq_2 = _rc(ins_seq)
    else:
        raise UnboundLocalError("ins_seq_2")            # SF:1860-1861 leave it unbound
    flank = default_flank_length if len(ins_seq) > default_flank_length else len(ins_seq)
    chrom = "_".join(ins_pos.split("_")[:-1])
    pos_s = ins_pos.split("_")[-1]
    pos = int(pos_s)
    scores: List[float] = []
    reads = seqio.simple_chop_pacbio_read_simple_short(bam_in, [chrom, pos_s] + [pos + len(ins_seq)], flank)
    if len(reads) > num_reads_cff:
        if len(ins_seq) < 5000:
            ref_seq = seqio.ref_seq_readin(ref, chrom, pos - flank, pos + flank + len(ins_seq))
            k = yield from _window(ref_seq + ins_seq)
        else:
            ref_seq = seqio.ref_seq_readin(ref, chrom, pos - flank, pos + flank)
            k = yield from _window(ref_seq)
        if not k == "Error":
            # flank + ins_seq + flank, SF:1872 (both flanks lie inside the window just read)
            alt_seq = _cat(_within(ref_seq, pos - flank, seqio.ref_seq_readin(ref, chrom, pos - flank, pos), pos - flank),
                           (ins_seq_2, None, None),
                           _within(ref_seq, pos - flank, seqio.ref_seq_readin(ref, chrom, pos, pos + flank), pos))

            def few_n(x):                                   # SF:1878
                return float(x[0].count("N") + x[0].count("n")) / float(len(x[0])) < 0.1

            kept = [few_n(x) for x in reads]
            used = [x for x, f in zip(reads, kept) if f]
            res = (yield Score("s1", ref_seq, alt_seq, used, k)) if used else []
            best = _collect(res, reads, scores, keep=kept)
            if ins_seq_2.count("X") == len(ins_seq_2):
                yield Figure(scores, best, k, ref_seq, ref_seq[2:flank], out_figure_name)
            else:
                yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_simple_disdup(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_simple_disdup_Vapor, SF:1786-1854.  sv_info = [chrom, s, e, ins_chrom, ins_pos]."""
    sv_info[1:3] = [int(i) for i in sv_info[1:3]]
    dup_block = sv_info[:3]
    ins_point = [sv_info[3], int(sv_info[4])]
    flank = seqio.flank_length_calculate(dup_block)
    scores: List[float] = []
    bp = sorted([int(i) for i in sv_info[1:3] + [sv_info[4]]])
    ran = False
    if sv_info[0] == sv_info[3] and max(bp) - min(bp) < default_max_sv_test:
        ref_seq = seqio.ref_seq_readin(ref, sv_info[0], min(bp) - flank, max(bp) + flank)
        k = yield from _window(ref_seq)
        if not k == "Error":
            reads = seqio.simple_chop_pacbio_read_simple_short(
                bam_in, [sv_info[0]] + bp + [int(bp[-1]) + sv_info[2] - sv_info[1]], flank)
            if len(reads) > num_reads_cff:
                ran = True
                if sv_info[4] > sv_info[2]:
                    structure = ["a", "b", "a"]
                elif sv_info[4] < sv_info[1]:
                    structure = ["b", "a", "b"]
                else:
                    raise UnboundLocalError("alt_structure")  # SF:1803-1804: insert point inside the block
                w0 = min(bp) - flank                   # (every block lies inside the window just read: slices of it)
                parts = [_within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], min(bp) - flank, min(bp)), w0)]
                a_seq = _within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], bp[0], bp[1]), bp[0])
                b_seq = _within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], bp[1], bp[2]), bp[1])
                for x in structure:
                    parts.append(a_seq if x == "a" else b_seq)
                parts.append(_within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], max(bp), max(bp) + flank), max(bp)))
                alt_seq = _cat(*parts)
                k = yield from _window(alt_seq)
                if not k == "Error":
                    res = yield Score("s3", ref_seq, alt_seq, reads, k)
                    best = _collect(res, reads, scores)
                    yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    if not ran:
        short = max(bp) - min(bp) < default_max_sv_test
        reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, ins_point, flank)
        if len(reads) > num_reads_cff:
            ref_seq = seqio.ref_seq_readin(ref, ins_point[0], ins_point[1] - flank, ins_point[1] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                if short:
                    alt_seq = _cat((ref_seq, None, flank), (seqio.ref_seq_readin(ref, dup_block[0], dup_block[1], dup_block[2]), None, None),
                                   (ref_seq, -flank, None))
                else:
                    alt_seq = _cat((ref_seq, None, flank), (seqio.ref_seq_readin(ref, dup_block[0], dup_block[1], dup_block[1] + flank), None, None))
                k = yield from _window(alt_seq)
                if not k == "Error":
                    res = yield Score("s1" if short else "s2", ref_seq, alt_seq, reads, k)
                    best = _collect(res, reads, scores)
                    yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_dup_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_dup_inv_VapoR, SF:1595-1665."""
    sv_info[1:3] = [int(i) for i in sv_info[1:3]]
    dup_block = sv_info[:3]
    ins_point = [sv_info[3], int(sv_info[4])]
    flank = seqio.flank_length_calculate(dup_block)
    scores: List[float] = []
    if sv_info[0] == sv_info[3]:
        bp = sorted(sv_info[1:3] + [sv_info[4]])
        ran = False
        if max(bp) - min(bp) < default_max_sv_test:
            ref_seq = seqio.ref_seq_readin(ref, sv_info[0], min(bp) - flank, max(bp) + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                ran = True
                if sv_info[4] > sv_info[2]:
                    structure = ["a", "b", "a^"]
                elif sv_info[4] < sv_info[1]:
                    structure = ["b^", "a", "b"]
                else:
                    structure = ["a", "a^"]
                reads = seqio.simple_chop_pacbio_read_simple_short(
                    bam_in, [sv_info[0]] + bp + [bp[-1] + sv_info[2] - sv_info[1]], flank)
                if len(reads) > num_reads_cff:
                    w0 = min(bp) - flank
                    parts = [_within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], min(bp) - flank, min(bp)), w0)]
                    a_seq = _within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], bp[0], bp[1]), bp[0])
                    b_seq = _within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], bp[1], bp[2]), bp[1])
                    for x in structure:
                        parts.append({"a": a_seq, "a^": _rcpart(a_seq), "b": b_seq, "b^": _rcpart(b_seq)}[x])
                    parts.append(_within(ref_seq, w0, seqio.ref_seq_readin(ref, sv_info[0], max(bp), max(bp) + flank), max(bp)))
                    alt_seq = _cat(*parts)
                    k = yield from _window(alt_seq)
                    if not k == "Error":
                        res = yield Score("s3", ref_seq, alt_seq, reads, k)
                        best = _collect(res, reads, scores)
                        yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
        if not ran:
            short = max(bp) - min(bp) < default_max_sv_test
            ref_seq = seqio.ref_seq_readin(ref, ins_point[0], ins_point[1] - flank, ins_point[1] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, ins_point, flank)
                if len(reads) > num_reads_cff:
                    if short:
                        alt_seq = _cat((ref_seq, None, flank), (seqio.ref_seq_readin(ref, dup_block[0], dup_block[1], dup_block[2]), None, None, True),
                                       (ref_seq, -flank, None))
                    else:
                        alt_seq = _cat((ref_seq, None, flank),
                                       (seqio.ref_seq_readin(ref, dup_block[0], dup_block[2] - flank, dup_block[2]), None, None, True))
                    k = yield from _window(alt_seq)
                    if not k == "Error":
                        res = yield Score("s1" if short else "s2", ref_seq, alt_seq, reads, k)
                        best = _collect(res, reads, scores)
                        yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_long_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_long_del_inv, SF:1667-1688.  sv_info = [[chrom, s, e, 'del'], [chrom, s, e, 'inv']]."""
    scores: List[float] = []
    flank = 500
    ref_seq = seqio.ref_seq_readin(ref, sv_info[0][0], sv_info[0][1] - flank, sv_info[1][1] + flank)
    k = yield from _window(ref_seq)
    if not k == "Error":
        alt_seq = _cat((ref_seq, None, flank), (seqio.ref_seq_readin(ref, sv_info[1][0], sv_info[1][2] - flank, sv_info[1][2]), None, None, True))
        k = yield from _window(alt_seq)
        if not k == "Error":
            reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, sv_info[0], flank)
            if len(reads) > num_reads_cff:
                res = yield Score("s2", ref_seq, alt_seq, reads, k)
                best = _collect(res, reads, scores)
                yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
    return scores


def vapor_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_del_inv_Vapor, SF:1557-1593.  sv_info = ordered [[chrom, s, e, 'del'|'inv'], ...]."""
    sv_block = [sv_info[0][0], sv_info[0][1], sv_info[-1][2]]
    flank = seqio.flank_length_calculate(sv_block)
    scores: List[float] = []
    if sv_info[1][1] - sv_info[0][2] < 100:
        if sv_block[2] - sv_block[1] < default_max_sv_test:
            ref_seq = seqio.ref_seq_readin(ref, sv_block[0], sv_block[1] - flank, sv_block[2] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                parts = [(ref_seq, None, flank)]
                for x in sv_info:
                    if x[-1] == "del":
                        continue
                    elif x[-1] == "inv":
                        parts.append(_rcpart(_within(ref_seq, sv_block[1] - flank, seqio.ref_seq_readin(ref, x[0], x[1], x[2]), x[1])))
                parts.append((ref_seq, -flank, None))
                alt_seq = _cat(*parts)
                k = yield from _window(alt_seq)
                if not k == "Error":
                    reads = seqio.simple_chop_pacbio_read_simple_short(
                        bam_in, sv_block[:2] + [sv_block[1] + len(alt_seq) - 2 * flank], flank)
                    if len(reads) > num_reads_cff:
                        res = yield Score("s1", ref_seq, alt_seq, reads, k)
                        best = _collect(res, reads, scores)
                        yield Figure(scores, best, k, ref_seq, alt_seq, out_figure_name)
                    else:
                        if len(sv_info) == 2 and [i[-1] for i in sv_info] == ["del", "inv"]:
                            # SF:1585 calls vapor_long_del_inv with four arguments
                            raise TypeError("vapor_long_del_inv() missing 2 required positional arguments: "
                                            "'sv_info' and 'out_figure_name'")
        else:
            if len(sv_info) == 2 and [i[-1] for i in sv_info] == ["del", "inv"]:
                scores = yield from vapor_long_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name)
    else:
        for sub in sv_info:
            if "del" in sub or "inv" in sub:
                # SF:1591-1592 call the simple drivers with four arguments
                raise TypeError("vapor_simple_%s_Vapor() missing 2 required positional arguments: "
                                "'sv_info' and 'out_figure_name'" % ("del" if "del" in sub else "inv"))
    return scores


# ------------------------------------------------------------------------------------------
# complex structures written as letter strings (SVelter style): 'ab_ab' -> 'b_b^' etc.
# ------------------------------------------------------------------------------------------

def letter_split(let):
    """SF:1013-1019: 'c^ba' -> ['c^', 'b', 'a']."""
    out = []
    for x in let:
        if not x == '^':
            out.append(x)
        else:
            out[-1] += x
    return out


def list_unify(items):
    """SF:1021-1025."""
    out = []
    for i in items:
        if i not in out:
            out.append(i)
    return out


def block_subsplot(bp_list, chromos):
    """SF:147-153: ['chr1', '10', '20', 'chr2', '5', '9'] -> [['chr1', 10, 20], ['chr2', 5, 9]]."""
    out = []
    for x in bp_list:
        if x not in chromos:
            out[-1].append(int(x))
        else:
            out.append([x])
    return out


def bp_to_chr_hash(bps, chromos, flank_length=500):
    """SF:98-114: letters a, b, ... for consecutive blocks, '-' / '+' for the flanks (with the
    reference's mix of int and str coordinates)."""
    groups = []
    for i in bps:
        if i in chromos:
            groups.append([i])
        else:
            groups[-1].append(i)
    out = {}
    rec = -1
    for k1 in groups:
        for k2 in range(len(k1[2:])):
            rec += 1
            out[chr(97 + rec)] = [k1[0], k1[k2 + 1], k1[k2 + 2]]
    last = out[sorted(out.keys())[-1]]
    out['+'] = [last[0], last[2], str(int(last[2]) + flank_length)]
    out['-'] = [out['a'][0], str(int(out['a'][1]) - flank_length), int(out['a'][1])]
    return out


def block_around_check(alt_allele, ref_allele):
    """SF:91-96: junctions of the alt allele that the ref allele does not have."""
    al = ['-'] + letter_split(alt_allele) + ['+']
    rl = ['-'] + letter_split(ref_allele) + ['+']
    n = len(letter_split(alt_allele)) + 1
    alt_j = [al[j:j + 2] for j in range(n)]
    ref_j = [rl[j:j + 2] for j in range(n)]
    return [i for i in alt_j if i not in ref_j]


def vapor_cannot_classify(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    """vapor_CANNOT_CLASSIFY_VapoR, SF:1490-1555.
    sv_info = ['ab_ab', 'b_b^', 'chr7', '70955990', '70961199', '70973901']."""
    ref_sv = sv_info[0].split('_')
    alt_sv = list_unify([i for i in sv_info[1].split('_') if i not in ref_sv])
    chromos = seqio.chromos_readin(ref)
    bp_info = block_subsplot(sv_info[2:], chromos)
    flank = max([seqio.flank_length_calculate(i) for i in bp_info])
    scores: List[float] = []
    ran = False
    if len(bp_info) == 1:
        b0 = bp_info[0]
        if b0[-1] - b0[1] < default_max_sv_test:
            ref_seq = seqio.ref_seq_readin(ref, b0[0], b0[1] - flank, b0[-1] + flank)
            k = yield from _window(ref_seq)
            if not k == "Error":
                reads = seqio.simple_chop_pacbio_read_simple_short(bam_in, b0, flank)
                let_hash = bp_to_chr_hash(b0, chromos, flank)
                if len(reads) > num_reads_cff:
                    ran = True
                    let_seq = {}
                    for i in list(let_hash.keys()):
                        let_seq[i] = seqio.ref_seq_readin(ref, let_hash[i][0], int(let_hash[i][1]), int(let_hash[i][-1]))
                    for alt_allele in alt_sv:
                        parts = [(ref_seq, None, flank)]
                        for i in letter_split(alt_allele):
                            blk = _within(ref_seq, b0[1] - flank, let_seq[i[0]], int(let_hash[i[0]][1]))
                            parts.append(blk if '^' not in i else _rcpart(blk))
                        parts.append((ref_seq, -flank, None))
                        alt_seq = _cat(*parts)
                        k = yield from _window(alt_seq)
                        if not k == "Error":
                            repeated = max([alt_allele.count(i) for i in alt_allele] + [0]) > 1
                            res = yield Score("s3" if repeated else "s1", ref_seq, alt_seq, reads, k)
                            best = _collect(res, reads, scores)
                            parts = out_figure_name.split('.')
                            yield Figure(scores, best, k, ref_seq, alt_seq,
                                         '.'.join(parts[:-1] + [ref_sv[0] + '.vs.' + alt_allele, parts[-1]]))
        if not ran:
            for alt_allele in alt_sv:
                juncs = block_around_check(alt_allele, ref_sv[0])
                let_hash = bp_to_chr_hash(b0, chromos, flank)
                for jun in juncs:
                    ha, hb = let_hash[jun[0][0]], let_hash[jun[1][0]]
                    if '^' not in jun[0]:
                        ref_a = seqio.ref_seq_readin(ref, ha[0], ha[2] - flank, ha[2] + flank)
                    else:
                        ref_a = _rc(seqio.ref_seq_readin(ref, ha[0], ha[1] - flank, ha[1] + flank))
                    if '^' not in jun[1]:
                        ref_b = seqio.ref_seq_readin(ref, hb[0], hb[1] - flank, hb[1] + flank)
                    else:
                        ref_b = _rc(seqio.ref_seq_readin(ref, hb[0], hb[2] - flank, hb[2] + flank))
                    k = yield from _window(ref_a + ref_b)
                    if not k == "Error":
                        alt_seq = ref_a[-flank:] + ref_b[:flank]
                        k = yield from _window(alt_seq)
                        if not k == "Error":
                            where = [ha[0], ha[2]] if '^' not in jun[0] else [ha[0], ha[1]]
                            reads = seqio.simple_del_chop_pacbio_read_simple_short(bam_in, where, flank)
                            if len(reads) > 0:
                                res = yield Score("s2", ref_a, alt_seq, reads, k)
                                _collect(res, reads, scores)
    return scores
