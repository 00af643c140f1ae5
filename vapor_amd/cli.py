"""`vapor bed | vcf | svelter | ins` - the reference's command line (vapor_vali/vapor:287-496) on the HIP path.

Same sub-commands, same flags, same output files and rows.  What differs is the schedule: the
reference scores one locus at a time; here every locus becomes a driver generator
(vapor_amd.drivers) and `pipeline.run_batch` sends the pending dot plots of a whole chunk of loci
to the GPU together.  With more than one rank (torchrun), loci are sharded across the GPUs of
the node and the per-locus rows are all-gathered (vapor_amd.dist).

Extra flags (not in the reference): --no-figures (skip the recurrence-plot PNGs),
--chunk (loci per device batch).
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Optional

from . import dist as vdist
from . import drivers, pipeline
from . import simple_function as SF
from .finish import result_organize_ins, row_tail


# ------------------------------------------------------------------------------------------
# input parsers (SURVEY.md component #10)
# ------------------------------------------------------------------------------------------

def bed_info_readin(bed_input, out_path):
    """vapor_vali/vapor:22-50: rows are chr start end SVID TYPE [INS sequence]."""
    out_path = SF.path_modify(out_path)
    SF.path_mkdir(out_path)
    out = []
    with open(bed_input) as fin:
        for line in fin:
            pin = line.strip().split()
            t = pin[4]
            if 'DUP' in t or 'duplication' in t:
                out.append([pin[0]] + [int(i) for i in pin[1:3]] + [pin[3]] + ['a/a', 'a/aa'])
            elif 'DEL' in t or 'deletion' in t:
                out.append([pin[0]] + [int(i) for i in pin[1:3]] + [pin[3]] + ['a/a', '/a'])
            elif 'INV' in t or 'inversion' in t:
                out.append([pin[0]] + [int(i) for i in pin[1:3]] + [pin[3]] + ['a/a', 'a/a^'])
            elif 'INS' in t or 'ALU' in t or 'HERVK' in t or 'LINE1' in t or 'SVA' in t or 'insertion' in t:
                if len(pin) > 5:
                    out.append([pin[0], int(pin[1]), int(pin[2]), pin[3], pin[5], 'INS'])
                elif '_' in t:
                    v = t.split('_')[1]
                    out.append([pin[0], int(pin[1]), int(pin[2]), pin[3], int(v) if v.isdigit() else v, 'INS'])
    return out


def block_reorganize(block_hash):
    """vapor_vali/vapor:83-97: blocks of one chromosome ordered by start, duplicates dropped."""
    if len(block_hash) == 1:
        for k1 in block_hash:
            start = [i[1] for i in block_hash[k1]]
            order = [start.index(i) for i in sorted(start)]
            out = []
            for b in [block_hash[k1][i] for i in order]:
                if b not in out:
                    out.append(b)
            return out
    return 'error'


def del_inv_interprete(pin):
    """vapor_vali/vapor:99-111: del=chr:s-e / inv=chr:s-e INFO entries."""
    out = {}
    for x in pin[7].split(';'):
        for tag, name in (('del=', 'del'), ('DEL=', 'del'), ('inv=', 'inv'), ('INV=', 'inv')):
            if tag in x:
                v = x.split('=')[1]
                blk = [v.split(':')[0]] + [int(i) for i in v.split(':')[1].split('-')]
                out.setdefault(blk[0], []).append(blk + [name])
                break
    return block_reorganize(out)


def dup_inv_interprete(pin):
    """vapor_vali/vapor:113-125."""
    seg = [pin[0], int(pin[1])]
    ins = []
    for x in pin[7].split(';'):
        if 'END=' in x:
            seg.append(int(x.split('=')[1]))
        if 'insert_point' in x or 'INSERT_POINT' in x:
            ins = x.split('=')[1].split(':')
    if len(ins) > 1:
        return seg + [ins[0], int(ins[1])]
    return 'error'


def vcf_list_readin(file_in):
    """vapor_vali/vapor:127-202: records bucketed by type in first-seen order, plus
    {file line index: key} for the INFO rewrite."""
    out = {}
    rec_hash = {}
    rec = -1
    # `x not in out[T]` of the reference scans the bucket's list (every record against every earlier one of its type: minutes
    # on a call set of 10^5); the same test from a set of the entries as tuples beside each list
    seen = {}

    def key(x):
        return tuple(key(i) if isinstance(i, list) else i for i in x)

    def new(bucket, probe, item=None):
        """`probe not in out[bucket]`, and if so the item (default: the probe) appended."""
        ks = seen.setdefault(bucket, set())
        if key(probe) in ks:
            return False
        item = probe if item is None else item
        out[bucket].append(item)
        ks.add(key(item))
        return True
    with open(file_in) as fin:
        for line in fin:
            rec += 1
            pin = line.strip().split()
            if pin[0][0] == '#':
                continue
            pin[7] = pin[7].replace('MERGE_TYPE=', 'SVTYPE=')
            t = SF.svtype_extract(pin)
            pos = SF.chr_start_end_extract(pin)

            if t in ['del', 'DEL', 'deletion']:
                out.setdefault('DEL', [])
                if new('DEL', pos):
                    rec_hash[rec] = ':'.join([str(i) for i in pos] + ['DEL'])
            elif t in ['inv', 'INV', 'inversion']:
                out.setdefault('INV', [])
                if new('INV', pos):
                    rec_hash[rec] = ':'.join([str(i) for i in pos] + ['INV'])
            elif t in ['ins', 'INS', 'insertion', 'LINE1', 'SVA', 'ALU', 'HERVK']:
                sv_len = int(SF.sv_len_extract(pin))
                seq = SF.sv_seq_extract(pin)
                if sv_len > 0:
                    out.setdefault('INS', [])
                    if new('INS', pos, pos[:2] + [sv_len, seq]):     # (the probe has three fields, the entries four: never a duplicate)
                        rec_hash[rec] = ':'.join([str(i) for i in pos[:2] + [sv_len]] + ['INS'])
            elif t in ['disdup', 'DISDUP', 'dis-dup']:
                ip = SF.sv_insert_point_define(pin)
                out.setdefault('DISDUP', [])
                if new('DISDUP', pos, pos + ip):
                    rec_hash[rec] = ':'.join([str(i) for i in pos + ip] + ['DISDUP'])
            elif t in ['DEL_INV', 'del_inv']:
                out.setdefault('DEL_INV', [])
                info = del_inv_interprete(pin)
                if not info == 'error' and new('DEL_INV', info):
                    rec_hash[rec] = ':'.join(['_'.join([str(i) for i in j]) for j in info] + ['DEL_INV'])
            elif t in ['DUP_INV', 'dup_inv']:
                out.setdefault('DUP_INV', [])
                info = dup_inv_interprete(pin)
                if not info == 'error' and new('DUP_INV', info):
                    rec_hash[rec] = ':'.join([str(i) for i in info + ['DUP_INV']])
            elif t in ['tandup', 'TANDUP', 'DUP']:
                out.setdefault('TANDUP', [])
                if new('TANDUP', pos):
                    rec_hash[rec] = ':'.join([str(i) for i in pos] + ['TANDUP'])
            elif t in ['CNV', 'CSV', 'CPX']:
                continue
            else:
                if 'Other=' in pin[7]:
                    info = [i for i in pin[7].split(';') if i[:6] == 'Other=']
                elif 'OTHER=' in pin[7]:
                    info = [i for i in pin[7].split(';') if i[:6] == 'OTHER=']
                else:
                    continue
                o = info[0].split('=')[1].split('_')
                item = ['_'.join(i.split('/')) for i in o[:2]] + o[2].split(':')
                out.setdefault('Other', [])
                if new('Other', item):
                    rec_hash[rec] = ':'.join([str(i) for i in item + ['CANNOT_CLASSIFY']])
    return [out, rec_hash]


# ------------------------------------------------------------------------------------------
# jobs
# ------------------------------------------------------------------------------------------

class Job:
    """One output row: how to score it (a driver generator factory, or fixed scores) and how to
    write it.  `cost`: what the locus is expected to take (microseconds, `job_cost`), for the shares of the ranks."""
    __slots__ = ("key", "make", "fixed", "row_prefix", "label", "cost", "spec", "ctx")

    def __init__(self, key, make=None, fixed=None, row_prefix=None, label=None, cost=None, spec=None, ctx=None):
        self.key, self.make, self.fixed, self.row_prefix, self.label = key, make, fixed, row_prefix, label
        # the four simple types also say WHAT they are - (type, chrom, start, end, ins_seq) and (num_reads_cff, bam, ref) - so
        # that a chunk of them can take the array route (vapor_amd.fastpath); `make` stays the driver's own route
        self.spec, self.ctx = spec, ctx
        self.cost = cost if cost is not None else (COST_FIXED_US if make is None else COST_HOST_US)


# What a locus costs, for the ranks' shares (greedy longest-processing-time on the estimated cost, not round-robin
# by index - spans run from 50 bp to 21 kb windows, vapor_vali/vapor:334-367).  Three terms, priced on one MI355X box with its
# host (tools/fit_job_cost.py, profiles/r04_job_cost_fit.json): the per-locus interpreter work; what scales with the bases
# handled (reads extracted, trimmed, staged, uploaded and packed; windows read and uploaded); and the device term SURVEY 8e
# names, n_reads x Lr x (La_ref + La_alt) nominal cells, at the rate the kernels go through them.
COST_FIXED_US = 0.5            # a row without device work (an SV below 50 bp in vcf mode)
COST_HOST_US = 40.0            # generator protocol, extraction call, tables, row (fit: 39-48)
COST_PER_KBASE_US = 0.35       # per 1 000 bases of reads and windows handled (fit: 0.27-0.64)
COST_PER_GCELL_US = 0.25       # per 1e9 nominal read-bp x window-bp cells: the kernels' own rate (cfg2: 8.0e11 cells per 0.178 ms
                               # pass); too small beside the host terms for the fit to see
COST_XMEANS_US = 500.0         # a tandem duplication's alt window always meets the X-means branch of the repeat check
                               # (sklearn / scipy on the host workers: 290-1 360 us per locus measured, by no simple rule of the span)
_READS_KEPT = 20               # minimize_pacbio_read_list keeps at most 20 reads (SF:1091-1102)


def job_cost(svtype: str, span: int, extra: int = 0) -> float:
    """Expected cost of one locus in microseconds from its type and span alone (windows as the drivers cut them, SURVEY.md
    3.2): `span` = end - start (INS: the inserted length; complex types: the whole region), `extra` = the duplicated block of
    DISDUP / DUP_INV.  An estimate for balancing shares - nothing depends on its accuracy but the ranks' idle time."""
    span = max(int(span), 0)
    f = min(500, span) if span > 0 else 500
    short = span < drivers.default_max_sv_test
    if svtype == 'DEL':
        lr, la = 2 * f, ((span + 2 * f) + 2 * f if short else 4 * f)
    elif svtype == 'INV':
        lr, la = (span + 2 * f, 2 * (span + 2 * f)) if short else (2 * f, 4 * f)
    elif svtype == 'TANDUP':
        lr, la = (2 * span + 2 * f, (span + 2 * f) + (2 * span + 2 * f)) if short else (2 * f, 4 * f)
    elif svtype == 'INS':
        lr, la = span + 2 * f, (2 * f + (span if span < 5000 else 0)) + (span + 2 * f)
    else:                                   # DISDUP, DUP_INV, DEL_INV, Other: the whole region when it is short
        lr, la = (span + extra + 2 * f, (span + 2 * f) + (span + extra + 2 * f)) if short else (2 * f, 4 * f)
    bases = _READS_KEPT * lr + la
    cells = _READS_KEPT * lr * la
    xmeans = COST_XMEANS_US if (svtype == 'TANDUP' and short) else 0.0
    return COST_HOST_US + COST_PER_KBASE_US * bases / 1e3 + COST_PER_GCELL_US * cells / 1e9 + xmeans


def bed_jobs(bed_info, num_reads_cff, bam_in, ref, out_path, sample_name) -> List[Job]:
    """The loop of vapor_vali/vapor:334-367."""
    jobs = []
    plt_li = 0
    ctx = (num_reads_cff, bam_in, ref)
    for x in bed_info:
        tag = x[-1]
        if tag in ['a/', '/a', '/', 'DEL']:
            key = ':'.join([str(i) for i in x[:-3]] + ['DEL'])
            fn, name = drivers.vapor_simple_del, 'DEL'
        elif tag in ['a/a^', 'a^/a', 'a^/a^', 'INV']:
            key = ':'.join([str(i) for i in x[:-3]] + ['INV'])
            fn, name = drivers.vapor_simple_inv, 'INV'
        elif tag in ['INS']:
            key = ':'.join([str(i) for i in x[:-3] + ['INS']])
            plt_li += 1
            ins_pos = '_'.join([str(i) for i in x[:2]])
            ins_seq = ''.join(['X' for _ in range(x[4])]) if type(x[4]) == type(4) else x[4]
            fig = out_path + sample_name + '.INS.' + key.replace(':', '__') + '.png'
            jobs.append(Job(key, (lambda p=plt_li, a=ins_pos, s=ins_seq, f=fig:
                                  drivers.vapor_simple_ins(num_reads_cff, p, bam_in, ref, a, s, f, '+')),
                            row_prefix=x[3], label=x, cost=job_cost('INS', len(ins_seq)),
                            spec=('INS', x[0], x[1], None, ins_seq), ctx=ctx))
            continue
        elif tag in ['a/aa', 'aa/a', 'aa/aa', 'DUP', 'TANDUP']:
            key = ':'.join([str(i) for i in x[:-3]] + ['TANDUP'])
            fn, name = drivers.vapor_simple_tandup, 'TANDUP'
        else:
            print(x)
            continue
        plt_li += 1
        fig = out_path + sample_name + '.' + name + '.' + key.replace(':', '__') + '.png'
        jobs.append(Job(key, (lambda p=plt_li, f=fn, info=x[:-3], g=fig: f(num_reads_cff, p, bam_in, ref, info, g)),
                        row_prefix=x[3], label=x, cost=job_cost(name, x[2] - x[1]), spec=(name, x[0], x[1], x[2], None), ctx=ctx))
    return jobs


def vcf_jobs(vcf_list, num_reads_cff, bam_in, ref, out_path, sample_name) -> List[Job]:
    """The loop of vapor_vali/vapor:387-465 (TANDUP is bucketed but never scored there either)."""
    jobs = []
    plt_li = 0
    ctx = (num_reads_cff, bam_in, ref)
    for x in list(vcf_list.keys()):
        if x not in ('DEL', 'INV', 'INS', 'DISDUP', 'DEL_INV', 'DUP_INV', 'Other'):
            print(x)
            continue
        for y in vcf_list[x]:
            if 'NA' in y:
                continue
            print(y)
            plt_li += 1
            if x in ('DEL', 'INV'):
                if y[2] - y[1] < 50:        # both branches label the row DEL (vapor_vali/vapor:394, 407)
                    jobs.append(Job(':'.join([str(i) for i in y] + ['DEL']), fixed=[]))
                    continue
                key = ':'.join([str(i) for i in y] + [x])
                fn = drivers.vapor_simple_del if x == 'DEL' else drivers.vapor_simple_inv
                fig = out_path + sample_name + '.' + x + '.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, f=fn, info=y, g=fig: f(num_reads_cff, p, bam_in, ref, info, g)),
                                cost=job_cost(x, y[2] - y[1]), spec=(x, y[0], y[1], y[2], None), ctx=ctx))
            elif x == 'INS':
                key = ':'.join([str(i) for i in y[:3] + ['INS']])
                ins_pos = '_'.join([str(i) for i in y[:2]])
                ins_seq = y[-1] if len(y) == 4 else ''.join(['X' for _ in range(y[2])])
                fig = out_path + sample_name + '.INS.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, a=ins_pos, s=ins_seq, g=fig:
                                      drivers.vapor_simple_ins(num_reads_cff, p, bam_in, ref, a, s, g, '+')),
                                cost=job_cost('INS', len(ins_seq)), spec=('INS', y[0], y[1], None, ins_seq), ctx=ctx))
            elif x == 'DISDUP':
                key = ':'.join([str(i) for i in y + ['DISDUP']])
                fig = out_path + sample_name + '.DISDUP.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, info=y, g=fig:
                                      drivers.vapor_simple_disdup(num_reads_cff, p, bam_in, ref, info, g)),
                                cost=_dup_cost('DISDUP', y)))
            elif x == 'DEL_INV':
                key = ':'.join(['_'.join([str(i) for i in j]) for j in y] + ['DEL_INV'])
                fig = out_path + sample_name + '.DEL_INV.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, info=y, g=fig:
                                      drivers.vapor_del_inv(num_reads_cff, p, bam_in, ref, info, g)),
                                cost=job_cost('DEL_INV', _num(y[-1][2]) - _num(y[0][1]))))
            elif x == 'DUP_INV':
                key = ':'.join([str(i) for i in y + ['DUP_INV']])
                fig = out_path + sample_name + '.DUP_INV.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, info=y, g=fig:
                                      drivers.vapor_dup_inv(num_reads_cff, p, bam_in, ref, info, g)),
                                cost=_dup_cost('DUP_INV', y)))
            elif x == 'Other':
                key = ':'.join([str(i) for i in y + ['CANNOT_CLASSIFY']])
                fig = out_path + sample_name + '.CANNOT_CLASSIFY.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, info=y, g=fig:
                                      drivers.vapor_cannot_classify(num_reads_cff, p, bam_in, ref, info, g)),
                                cost=_other_cost(y)))
    return jobs


def _num(v) -> int:
    try:
        return int(v)
    except (TypeError, ValueError):
        return 0


def _dup_cost(svtype, y) -> float:
    """DISDUP / DUP_INV record [chrom, s, e, ins_chrom, ins_pos]: the region the drivers cut when block and insert point
    share a contig, the block itself otherwise."""
    s0, e0 = _num(y[1]), _num(y[2])
    if len(y) > 4 and y[0] == y[3]:
        bp = sorted([s0, e0, _num(y[4])])
        return job_cost(svtype, bp[-1] - bp[0], e0 - s0)
    return job_cost(svtype, e0 - s0)


def _other_cost(info) -> float:
    """`Other=` / SVelter record [ref structure, alt structure, chrom, bp, bp, ...]: the span of its numeric fields, once per
    alt allele."""
    nums = [int(v) for v in info[2:] if str(v).isdigit()]
    span = (max(nums) - min(nums)) if len(nums) >= 2 else 0
    n_alt = max(1, len([a for a in str(info[1]).split('_') if a and a not in str(info[0]).split('_')]))
    return n_alt * job_cost('Other', span)


def svelter_readin(file_in):
    """vapor_vali/vapor:255-268: {ref structure: {alt structure: [[chrom, bp, bp, ...], ...]}}."""
    out = {}
    seen = {}                       # (the reference scans the list of a structure pair per record; a set of its entries beside it)
    with open(file_in) as fin:
        fin.readline()
        for line in fin:
            pin = line.strip().split()
            r = '_'.join(pin[4].split('/'))
            a = '_'.join(pin[5].split('/'))
            lst = out.setdefault(r, {}).setdefault(a, [])
            ks = seen.setdefault((r, a), set())
            item = pin[3].split(':')
            if tuple(item) not in ks:
                ks.add(tuple(item))
                lst.append(item)
    return out


def svelter_jobs(sv_hash, num_reads_cff, bam_in, ref, out_path, sample_name) -> List[Job]:
    """The loop of vapor_vali/vapor:481-492."""
    jobs = []
    plt_li = 0
    for k1 in list(sv_hash.keys()):
        for k2 in list(sv_hash[k1].keys()):
            for k3 in sv_hash[k1][k2]:
                plt_li += 1
                key = '.' + '_'.join(k3)
                fig = out_path + sample_name + key.replace(':', '__') + '.png'
                info = [k1, k2] + k3
                print(info)
                jobs.append(Job(key, (lambda p=plt_li, i=info, g=fig:
                                      drivers.vapor_cannot_classify(num_reads_cff, p, bam_in, ref, i, g)),
                                cost=_other_cost(info)))
    return jobs


def output_row(head: list, scores) -> tuple:
    """The line write_output_main (SF:2084-2088) appends for one locus - `head` fields, then what result_organize_ins
    (SF:1219-1231) and gt_estimate_log_likelihood (SF:2054-2069) make of the scores - and the five values behind it
    (finish.row_tail: one rounding per score instead of round -> str -> split -> float).  The reference's test for an
    unscored locus is `'NA' in out_list`, over ALL fields: a head field that reads NA turns GT, GQ and Rec into NA as well."""
    tail = row_tail(scores)
    if tail[0] != 'NA' and 'NA' in head:
        tail = [tail[0], tail[1], 'NA', 'NA', 'NA']
    return '\t'.join([str(i) for i in head + tail]), tail


def output_rows(heads: list, scores_list: list) -> tuple:
    """output_row's line for every locus of the table, and the five values of finish.row_tail behind each (as computed: the
    NA rule of the writer changes the line only - vapor_vali/vapor:357 prints the result before the writer looks at it);
    the tails through finish.row_tails, one call of the library's host helper for the whole table."""
    from .finish import row_tails
    tails = row_tails(scores_list)
    lines = []
    for head, tail in zip(heads, tails):
        if tail[0] != 'NA' and 'NA' in head:
            tail = [tail[0], tail[1], 'NA', 'NA', 'NA']
        lines.append('\t'.join([str(i) for i in head + tail]))
    return lines, tails


def score_jobs(jobs: List[Job], chunk: int, figure_fn=None) -> List[object]:
    """Score every job (sharded over ranks, batched on each GPU); returns per job the list of read
    scores, in job order, identical on every rank."""
    import gc
    import time
    t0 = time.perf_counter()
    # The cyclic collector looks at every container alive each time its oldest generation is due, and a run keeps its jobs,
    # generators and read lists alive until the table is written: on an 8 000-locus run a quarter of the time went into
    # three or four such passes of ~50 ms, and the share grows with the run.  The loop makes no reference cycles outside its error
    # paths (reference counting frees the rest), so the thresholds are raised for its duration: young objects are still
    # collected every 200 000 net allocations, the older generations practically never.
    gc_was = gc.get_threshold()
    gc.set_threshold(max(gc_was[0], 200000), max(gc_was[1], 50), max(gc_was[2], 1000))
    try:
        return _score_jobs(jobs, chunk, figure_fn, t0)
    finally:
        gc.set_threshold(*gc_was)


def _chunk_threads_ok() -> bool:
    """Chunks are scored on several threads only with a read backend whose handles are per thread - the rule of
    pipeline._prefetch_threads: the Python BGZF reader (VAPOR_BAM_NATIVE=0) shares one file object and block cache between
    seek() and read(), the samtools hybrid and VAPOR_MEMORY_CHOP=records write one module-level result array."""
    from . import seqio
    be = seqio.get_backend()
    if not getattr(be, "chunk_threads_ok", False) or os.environ.get("VAPOR_BAM_NATIVE", "1") == "0":
        return False
    return os.environ.get("VAPOR_MEMORY_CHOP", "") != "records"


last_timing: dict = {}          # of the most recent score_jobs: seconds scoring this rank's share, seconds in the gather


def _score_jobs(jobs, chunk, figure_fn, t0):
    import time
    # shares by estimated cost (greedy longest-processing-time), the same list on every rank
    costs = [float(j.cost) for j in jobs]
    mine = vdist.my_share(len(jobs), costs)
    local: dict = {}

    def one_chunk(a, engine=None):
        part = mine[a:a + chunk]
        todo = [t for t in part if jobs[t].make is not None]
        done = {}
        if figure_fn is None and os.environ.get("VAPOR_FAST_PATH", "1") != "0":
            # the simple types of the chunk in array form (vapor_amd.fastpath); what leaves the drivers' straight route comes
            # back unanswered and goes the generators' way below - as everything does when figures are drawn (they need the
            # best read as text)
            from . import fastpath, seqio
            by_ctx = {}
            for t in todo:
                j = jobs[t]
                if j.spec is not None and j.ctx is not None:
                    by_ctx.setdefault(j.ctx, []).append(t)
            for ctx, ts in by_ctx.items():
                eng = engine or pipeline.get_engine()
                if len(ts) >= 8 and fastpath.capable(seqio.get_backend(), ctx[1], eng):
                    got = fastpath.run(eng, [jobs[t].spec for t in ts], ctx[1], ctx[2], ctx[0])
                    for t, r in zip(ts, got):
                        if r is not fastpath.FALLBACK:
                            done[t] = r
        rest = [t for t in todo if t not in done]
        res = pipeline.run_batch([jobs[t].make() for t in rest], engine=engine, figure_fn=figure_fn) if rest else []
        for t, r in zip(rest, res):
            done[t] = r
        return part, todo, [done[t] for t in todo]

    in_flight = max(1, int(os.environ.get("VAPOR_CHUNKS_IN_FLIGHT", "3")))
    if in_flight >= 2 and not _chunk_threads_ok():
        in_flight = 1
    if in_flight >= 2 and 256 <= len(mine) <= chunk:
        # a share of one chunk: in halves (thirds from 1 536 loci on), so that every thread has one - the native half of a
        # chunk's work (read selection, upload, planning, kernels) runs beside the others' Python
        chunk = -(-len(mine) // (min(in_flight, 3) if len(mine) >= 1536 else 2))
    starts = list(range(0, len(mine), max(chunk, 1)))
    if len(starts) >= 2 and in_flight >= 2:
        # Two chunks in flight (the reference's loop over loci, vapor_vali/vapor:334-367, has no such stage): each on a thread
        # with a library context of its own (one host thread per context), so that the host preparation of one chunk - allele
        # strings, read extraction, tables - runs while the other waits for its uploads and kernels; on the device the two
        # contexts' streams overlap as well.  Results are taken in chunk order.
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(in_flight, len(starts))) as pool:
            import threading
            slot = {}
            shared = set()
            lock = threading.Lock()

            def work(a):
                with lock:
                    k = slot.setdefault(threading.get_ident(), len(slot))
                eng = pipeline.engine_slot(k)
                if k not in shared and hasattr(eng, "set_param"):
                    # several chunks at once: a chunk's read extraction on the device keeps to five eighths of the CUs, so that
                    # the other chunks' packing / join / clean kernels do not wait behind its inflating wavefronts
                    shared.add(k)
                    try:
                        eng.set_param("bam_cu_share", int(os.environ.get("VAPOR_BAM_CU_EIGHTHS", "5")))
                    except Exception:       # noqa: BLE001 - an engine without the parameter (tests' stand-ins)
                        pass
                return one_chunk(a, eng)
            try:
                done = list(pool.map(work, starts))
            finally:
                for k in shared:                       # (a later run of one chunk has the device to itself)
                    try:
                        pipeline.engine_slot(k).set_param("bam_cu_share", 0)
                    except Exception:       # noqa: BLE001
                        pass
    else:
        done = [one_chunk(a) for a in starts]
    for part, todo, res in done:
        for t, r in zip(todo, res):
            local[t] = r
        for t in part:
            if jobs[t].make is None:
                local[t] = jobs[t].fixed
    t1 = time.perf_counter()
    allres = vdist.gather_results(local, len(jobs), costs)
    last_timing.update(score_s=t1 - t0, gather_s=time.perf_counter() - t1, loci=len(mine), cost=sum(costs[t] for t in mine))
    if os.environ.get("VAPOR_TIMING") and vdist.rank() == 0:
        dt = time.perf_counter() - t0
        print("vapor_amd.cli: scored %d loci on %d rank(s) in %.3f s -> %.1f loci/s" % (len(jobs), vdist.world(), dt, len(jobs) / dt),
              file=sys.stderr)
    for r in allres:
        if isinstance(r, BaseException):
            raise r
    return allres


# ------------------------------------------------------------------------------------------
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="vapor", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--sv-input', required=True, help='input file of SV calls')
    p.add_argument('--reference', required=True, help='reference sequences')
    p.add_argument('--pacbio-input', required=True, help='input pacbio sequences in bam format')
    p.add_argument('--output-path', required=True, help='path of output VaPoR figures')
    p.add_argument('--output-file', required=True, help='name of output file')
    p.add_argument('--PB-supp', required=False, help='minimum number of evaluable PacBio reads')
    p.add_argument('--no-figures', action='store_true', help='do not render recurrence-plot PNGs')
    p.add_argument('--chunk', type=int, default=2048, help='loci per device batch')
    return p


def main(argv: Optional[List[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 1:
        from . import prep
        prep.print_read_me()
        return 0
    mode = argv[0]
    if len(argv) == 1:
        from . import prep
        {'bed': prep.readme_bed, 'vcf': prep.readme_vcf, 'ins': prep.readme_melt}.get(mode, prep.print_read_me)()
        return 0
    args = build_parser().parse_args(argv[1:])
    num_reads_cff = int(args.PB_supp) if args.PB_supp else 3
    figure_fn = None
    if not args.no_figures:
        from . import figures
        figure_fn = figures.make_event_figure_1
        figures.warm()                  # (the drawing processes start while the input is parsed and the first batch scored)
    vdist.init_from_env()
    out_path = SF.path_modify(args.output_path)
    SF.path_mkdir(out_path)
    sample_name = '.'.join(args.sv_input.split('/')[-1].split('.')[:-1])
    bam_in, ref = args.pacbio_input, args.reference
    if mode == 'bed':
        bed_info = bed_info_readin(args.sv_input, out_path)
        jobs = bed_jobs(bed_info, num_reads_cff, bam_in, ref, out_path, sample_name)
        scores = score_jobs(jobs, args.chunk, figure_fn)
        if vdist.rank() == 0:
            SF.write_output_initiate(args.output_file)
            with open(args.output_file, 'a') as fo:
                # (result_organize_ins + write_output_main of vapor_vali/vapor:356-357 in one go: finish.row_tails)
                lines, tails = output_rows([j.key.split(':') + [j.row_prefix] for j in jobs], scores)
                fo.write(''.join([l + '\n' for l in lines]))
                for j, tail in zip(jobs, tails):
                    print([j.key, tail[0], tail[1], tail[4]])
    elif mode == 'vcf':
        vcf_list, rec_hash = vcf_list_readin(args.sv_input)
        rec_new = SF.vcf_rec_hash_modify(rec_hash)
        jobs = vcf_jobs(vcf_list, num_reads_cff, bam_in, ref, out_path, sample_name)
        scores = score_jobs(jobs, args.chunk, figure_fn)
        if vdist.rank() == 0:
            SF.write_output_initiate(args.sv_input + '.vapor')
            with open(args.sv_input + '.vapor', 'a') as fo:
                fo.write(''.join([l + '\n' for l in output_rows([[j.key] for j in jobs], scores)[0]]))
            SF.vcf_vapor_modify(args.sv_input, rec_new)
    elif mode == 'svelter':
        jobs = svelter_jobs(svelter_readin(args.sv_input), num_reads_cff, bam_in, ref, out_path, sample_name)
        scores = score_jobs(jobs, args.chunk, figure_fn)
        if vdist.rank() == 0:
            with open(args.output_file, 'a') as fo:      # appended, never initialised (vapor_vali/vapor:492)
                fo.write(''.join([l + '\n' for l in output_rows([[j.key] for j in jobs], scores)[0]]))
    elif mode == 'ins':
        from . import melt
        melt.run(args.sv_input, out_path, sample_name.split('.')[0], bam_in, ref, num_reads_cff, args.chunk, figure_fn)
    else:
        raise SystemExit("vapor: unknown mode %r (bed | vcf | svelter | ins)" % mode)
    vdist.finalize()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
