"""End-to-end rate of the driver pipeline (`vapor bed` without the files): synthetic world ->
driver generators -> pipeline.run_batch -> rows.  GPU box.  usage: bench_pipeline.py [n_loci]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vapor_amd import synth, seqio, cli, pipeline
from vapor_amd.finish import result_organize_ins
from vapor_amd import simple_function as SF

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
t0 = time.perf_counter()
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), span_range=(100, 4000),
                     read_len=9500, n_reads=20)
print("world %.1fs" % (time.perf_counter() - t0), flush=True)
seqio.set_backend(seqio.MemorySamtools(w))
import tempfile
tmp = tempfile.mkdtemp()
bed = os.path.join(tmp, "in.bed"); open(bed, "w").write(synth.bed_text(w))
bed_info = cli.bed_info_readin(bed, tmp)
pipeline.get_engine()

def run():
    jobs = cli.bed_jobs(bed_info, 3, "x.bam", "ref.fa", tmp + "/", "s")
    scores = cli.score_jobs(jobs, 2048, None)
    rows = []
    for j, sc in zip(jobs, scores):
        res = result_organize_ins([j.key, sc])
        rows.append(SF.format_output_row(res[0].split(':') + [j.row_prefix] + res[1:]))
    return rows

run()
t0 = time.perf_counter(); rows = run(); dt = time.perf_counter() - t0
print("%d loci in %.3f s -> %.1f loci/s" % (len(rows), dt, len(rows) / dt))
cProfile.run("run()", "/tmp/pipe.prof")
pstats.Stats("/tmp/pipe.prof").sort_stats("tottime").print_stats(28)
