"""The reference's module surface (vapor_vali/Simple_function.pyx, "SF") on the HIP path.

Same function names, positional arguments, return shapes and sentinels ('Error', [0, 0], 'NA');
everything that fills or reduces a recurrence plot goes through libvapor_hip.so.  Functions of
the reference that no CLI path reaches (SURVEY.md components #7, #8) are not provided.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib as L
from . import drivers, finish, pipeline
from .finish import gt_estimate_log_likelihood, result_organize_ins  # noqa: F401  (SF:1219, SF:2054)
from .seqio import (bam_in_decide, chop_pacbio_read_by_pos, chromos_readin, cigar2alignstart_by_pos,  # noqa: F401
                    complementary, flank_length_calculate, minimize_pacbio_read_list, ref_seq_readin, reverse,
                    simple_chop_pacbio_read_simple_short, simple_del_chop_pacbio_read_simple_short)

invert_base = {'A': 'T', 'T': 'A', 'C': 'G', 'G': 'C', 'N': 'N', 'a': 't', 't': 'a', 'c': 'g', 'g': 'c', 'n': 'n'}
default_flank_length = drivers.default_flank_length
default_read_length = 4000
default_max_sv_test = drivers.default_max_sv_test

_figure_fn = None


def set_figure_function(fn) -> None:
    """Renderer for make_event_figure_1 requests (None = figures off)."""
    global _figure_fn
    _figure_fn = fn


# ---------------------------------------------------------------------------
# recurrence plots
# ---------------------------------------------------------------------------

def dotdata(kmerlen, seq1, seq2):
    """SF:545-549: [(pos_in_seq2, pos_in_seq1), ...] in the reference's order."""
    eng = pipeline.get_engine()
    ss = eng.seqset([seq1, seq2])
    try:
        st, hits = eng.dotplots(ss, eng.make_pairs([(0, 1, 0, int(kmerlen), 0)]))
    finally:
        ss.close()
    pipeline._raise_for_status(st[0])
    return [(int(j), int(i)) for j, i in hits[0]]


def window_size_refine(seq2, region_QC_Cff=0.4):
    """SF:2030-2046."""
    r = pipeline.refine_windows(pipeline.get_engine(), [seq2], region_QC_Cff)[0]
    if isinstance(r, BaseException):
        raise r
    return r


def qual_check_repetitive_region(dotdata_qual_check):
    """SF:1154-1171 on an explicit dot list."""
    from . import repeat_qc
    n = len(dotdata_qual_check)
    nd = sum(1 for x in dotdata_qual_check if x[0] == x[1])
    low = [x for x in dotdata_qual_check if x[0] > x[1]]
    return repeat_qc.qual_check_from_counts(n, nd, len(low), lambda: ([x[0] for x in low], [x[1] for x in low]))


def clean_dotdata_diagnal_and_anti_diagnal(ref_dotdata):
    """SF:432-448."""
    if ref_dotdata == []:
        return [[], []]
    arr = np.asarray(ref_dotdata, dtype=np.int32).reshape(-1, 2)
    _st, fl = pipeline.get_engine().clean_hits([arr], flags=[L.PF_C1])
    return [ref_dotdata[t] for t in range(len(ref_dotdata)) if fl[0][t] & L.HF_C1_KEPT]


def _one_score(kind, ref_seq, alt_seq, x, window_size):
    return pipeline.scorer_outputs(pipeline.get_engine(), kind, ref_seq, alt_seq, x, window_size)


def calcu_vapor_single_read_score_abs_dis_m1b(ref_seq, alt_seq, x, window_size):
    """SF:182-203."""
    return _one_score("s1", ref_seq, alt_seq, x, window_size)


def calcu_vapor_single_read_score_within_10Perc_m1b(ref_seq, alt_seq, x, window_size):
    """SF:277-294."""
    return _one_score("s2", ref_seq, alt_seq, x, window_size)


def calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal(ref_seq, alt_seq, x, window_size):
    """SF:241-257."""
    return _one_score("s3", ref_seq, alt_seq, x, window_size)


def log_likelihood_calcu(k, l, m, g, err=0.05):
    """SF:2071-2077."""
    out = -k * np.log(m)
    for _ in range(l):
        out += np.log((m - g) * err + g * (1 - err))
    for _ in range(k - l):
        out += np.log((m - g) * (1 - err) + g * err)
    return out


# ---------------------------------------------------------------------------
# locus drivers, one locus per call (the reference's signatures)
# ---------------------------------------------------------------------------

def _sync(gen):
    return pipeline.run_sync(gen, figure_fn=_figure_fn)


def vapor_simple_del_Vapor(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_simple_del(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_simple_inv_Vapor(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_simple_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_simple_tandup_Vapor(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_simple_tandup(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_simple_ins_Vapor(num_reads_cff, plt_li, bam_in, ref, ins_pos, ins_seq, out_figure_name, POLARITY):
    return _sync(drivers.vapor_simple_ins(num_reads_cff, plt_li, bam_in, ref, ins_pos, ins_seq, out_figure_name, POLARITY))


def vapor_simple_disdup_Vapor(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_simple_disdup(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_dup_inv_VapoR(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_dup_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_del_inv_Vapor(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_long_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_long_del_inv(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


def vapor_CANNOT_CLASSIFY_VapoR(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name):
    return _sync(drivers.vapor_cannot_classify(num_reads_cff, plt_li, bam_in, ref, sv_info, out_figure_name))


from .drivers import block_around_check, block_subsplot, bp_to_chr_hash, letter_split, list_unify  # noqa: E402,F401


# ---------------------------------------------------------------------------
# output writers and small helpers (SURVEY.md components #5, #6)
# ---------------------------------------------------------------------------

def path_mkdir(path):
    """SF:1138-1140."""
    if not os.path.isdir(path):
        os.makedirs(path, exist_ok=True)


def path_modify(path):
    """SF:1142-1145."""
    return path if path[-1] == '/' else path + '/'


def write_output_initiate(out_name):
    """SF:2079-2082."""
    with open(out_name, 'w') as fo:
        print('\t'.join(['#CHR', 'POS', 'END', 'SVTYPE', 'SVID', 'VaPoR_QS', 'VaPoR_GS', 'VaPoR_GT', 'VaPoR_GQ',
                         'VaPoR_Rec']), file=fo)


def format_output_row(out_list) -> str:
    """The line write_output_main appends (SF:2084-2088)."""
    if 'NA' not in out_list:
        return '\t'.join([str(i) for i in out_list[:-1] + gt_estimate_log_likelihood(out_list) + [out_list[-1]]])
    return '\t'.join([str(i) for i in out_list[:-1] + ['NA', 'NA', 'NA']])


def write_output_main(out_name, out_list):
    with open(out_name, 'a') as fo:
        print(format_output_row(out_list), file=fo)


def _info_field(pin, test, take, default):
    """One accessor for the INFO column (pin[7], ';'-separated) behind the seven reference functions below.  Each of them
    walks the entries and lets every entry that passes its test overwrite the answer - so the LAST such entry decides, a
    malformed earlier one still raises where the reference would (every hit is converted), and `default` stands when none
    passes.  `test`: a substring that must occur anywhere in the entry (str) or a prefix it must start with (('prefix',))."""
    if isinstance(test, tuple):
        hits = [x for x in pin[7].split(';') if x[:len(test[0])] == test[0]]
    else:
        hits = [x for x in pin[7].split(';') if test in x]
    vals = [take(x) for x in hits]
    return vals[-1] if vals else default


def _after_equals(x):
    return x.split('=')[1]


def svtype_extract(pin):
    """SF:1424-1431: SVTYPE from INFO, else the ALT column without its angle brackets."""
    return _info_field(pin, 'SVTYPE', _after_equals, '') or pin[4].replace('<', '').replace('>', '')


def chr_start_end_extract(pin):
    """SF:365-370: [chrom, pos] and an entry for EVERY END= of INFO (the reference appends each)."""
    return [pin[0], int(pin[1])] + [int(_after_equals(x)) for x in pin[7].split(';') if x[:4] == 'END=']


def sv_len_extract(pin):
    """SF:1433-1440: the text behind SVLEN (a str), the int 0 when there is none or it is empty."""
    return _info_field(pin, 'SVLEN', _after_equals, '') or 0


def sv_seq_extract(pin):
    """SF:1442-1447."""
    return _info_field(pin, ('SEQ=',), _after_equals, '')


def sv_insert_point_define(pin):
    """SF:1449-1456: insert_point=chrom:pos as a two-element list of str; [0, 0] without one."""
    return _info_field(pin, 'insert_point=', lambda x: _after_equals(x).split(':'), [0, 0])


def INS_length_detect(pin):
    """SF:833-838."""
    return _info_field(pin, 'SVLEN=', lambda x: int(_after_equals(x)), 0)


def polarity_detect(pin):
    """SF:1147-1152: the last comma-separated piece of the MEIINFO entry."""
    return _info_field(pin, 'MEIINFO=', lambda x: x.split(',')[-1], '+')


def vcf_rec_hash_modify(vcf_rec_hash):
    """SF:1935-1940: invert {record index: key} to {key: [record indices]}."""
    out = {}
    for k1, v in vcf_rec_hash.items():
        out.setdefault(v, []).append(k1)
    return out


def vcf_vapor_modify(vcf_input, vcf_rec_hash_new, header_offset_compat=False):
    """SF:1972-2028 (the second definition, which shadows SF:1942): rewrite <vcf>.vapor as the
    input VCF with ;VaPor_GS=..;VaPor_GT=..;VaPor_GQ=..;VaPor_REC=.. appended to INFO of every
    scored record.

    The reference numbers records two ways - vcf_list_readin counts every line of the file
    (vapor_vali/vapor:131-134), this function only the non-header lines (SF:1982-1987) - so with
    H header lines it annotates record r+H instead of r, or dies with KeyError.  Both agree, and
    this function matches the reference byte for byte, on header-less input.  Here record indices
    are file line numbers throughout, so headers are fine; header_offset_compat=True reproduces
    the reference's shifted lookup."""
    vapor_input = vcf_input + '.vapor'
    info = {}
    meta, header = [], []
    rec = -1
    line_no = -1
    with open(vcf_input) as fin:
        for line in fin:
            line_no += 1
            pin = line.strip().split()
            if not pin[0][0] == '#':
                rec += 1
                info[rec if header_offset_compat else line_no] = pin
            elif not pin[0] == '#CHROM':
                meta.append(pin)
            else:
                header = pin
    keep = []
    with open(vapor_input) as fin:
        for line in fin:
            pin = line.strip().split()
            if pin[0] in vcf_rec_hash_new:
                for y in vcf_rec_hash_new[pin[0]]:
                    gs = round(float(pin[2]), 2) if not pin[2] == 'NA' else pin[2]
                    gq = round(float(pin[4]), 2) if not pin[4] == 'NA' else pin[4]
                    info[y][7] += (';VaPor_GS=' + str(gs) + ';VaPor_GT=' + str(pin[3]) + ';VaPor_GQ=' + str(gq)
                                   + ';VaPor_REC=' + str(pin[5]))
                    keep.append(y)
    with open(vapor_input, 'w') as fo:
        prev = ''
        for line in meta:
            joined = ' '.join(line)
            cur = joined.split('=')[0]
            if prev == '##INFO' and not cur == '##INFO':
                print('##INFO=<ID=VaPoR_GS,Number=1,Type=Float,Description="VaPoR Score, representing the percentage of transverse long reads that support the prediction">', file=fo)
                print('##INFO=<ID=VaPoR_GT,Number=1,Type=String,Description="Genotype with the highest likelihood as estimated by VaPoR">', file=fo)
                print('##INFO=<ID=VaPoR_GQ,Number=1,Type=Float,Description="Genotype quality score - likelihood of the second most likely genotype on a -log10 normalized scale"', file=fo)
                print('##INFO=<ID=VaPoR_REC,Number=.,Type=Float,Description="Similarity scores assigned to each of the reads traversings the predicted SV">', file=fo)
            print(joined, file=fo)
            prev = cur
        print('\t'.join(header), file=fo)
        kept = set(keep)                  # (the reference asks its list per record: 15 s of scans on 60 000 records)
        for k1 in sorted(info.keys()):
            if k1 in kept:
                print('\t'.join([str(i) for i in info[k1]]), file=fo)
