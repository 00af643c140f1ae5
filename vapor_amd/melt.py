"""`vapor ins`: MELT insertion calls with their assembled sequences (melt_info_readin,
vapor_vali/vapor:52-81).  <prefix>.vcf holds the sites, <prefix>.fa the inserted sequences keyed
'chrom_pos'; results go to <prefix>.vapor.  (The reference reads args.sv_input_prefix, which
its argparse never defines; here the prefix is --sv-input.)"""
from __future__ import annotations

from . import drivers, seqio
from . import simple_function as SF


def run(prefix, out_path, sample_name, bam_in, ref, num_reads_cff, chunk, figure_fn) -> None:
    from .cli import Job, job_cost, output_rows, score_jobs
    from . import dist as vdist
    jobs = []
    plt_li = 0
    with open(prefix + '.vcf') as fin:
        for line in fin:
            pin = line.strip().split()
            if pin[0][0] == '#':
                continue
            key = '_'.join(pin[:2])
            lines = list(seqio.get_backend().faidx_lines(prefix + '.fa', key))
            ins_seq = ''.join(l.strip() for l in lines[1:])
            if ins_seq == '':
                ins_seq = ''.join(['X' for _ in range(SF.INS_length_detect(pin))])
            if not ins_seq == '' and 'INS' in pin[3]:
                pol = SF.polarity_detect(pin)
                ins_seq = ins_seq.replace('N', 'X')
                plt_li += 1
                fig = out_path + sample_name + '.INS.' + key.replace(':', '__') + '.png'
                jobs.append(Job(key, (lambda p=plt_li, a=key, s=ins_seq, g=fig, q=pol:
                                      drivers.vapor_simple_ins(num_reads_cff, p, bam_in, ref, a, s, g, q)),
                                cost=job_cost('INS', len(ins_seq))))
    scores = score_jobs(jobs, chunk, figure_fn)
    if vdist.rank() == 0:
        SF.write_output_initiate(prefix + '.vapor')
        with open(prefix + '.vapor', 'a') as fo:
            # (format_output_row(result_organize_ins([key, scores])) per record, SF:1219-1231 / 2084-2088: the table's rows in one go)
            fo.write(''.join([l + '\n' for l in output_rows([[j.key] for j in jobs], scores)[0]]))
