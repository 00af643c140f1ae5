"""Read extraction alone (FASTA window + BAM region through the in-process readers) on a synthetic world written to
files: native reader at several thread counts against the Python statement.  usage: bam_chop_timing.py [n_loci]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vapor_amd import _lib, seqio, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
for c in w.reads:
    w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
tmp = tempfile.mkdtemp()
fa, bam = synth.write_world_files(w, tmp, block_size=0xFF00)
be = seqio.InProcessBam()
seqio.set_backend(be)


def run(fn):
    t0 = time.perf_counter()
    for l in w.loci:
        f = min(500, l.end - l.start) if l.svtype != "INS" else 500
        fn(bam, l.chrom, l.start - f, l.end + f, f)
    return (time.perf_counter() - t0) / n * 1e3


run(be.chop)
lib = _lib.load()
for nt in (1, 2, 4, 8):
    lib.vapor_bam_set_threads(be._open(bam)._free[-1]["native"], nt)      # (the handle the run above opened and will take again)
    print("native, %d inflate thread(s): %.3f ms per locus" % (nt, run(be.chop)), flush=True)
print("python statement:            %.3f ms per locus" % run(be.chop_python))
t0 = time.perf_counter()
for l in w.loci:
    f = min(500, l.end - l.start) if l.svtype != "INS" else 500
    seqio.ref_seq_readin(fa, l.chrom, l.start - f, l.end + f)
print("reference window (.fai):     %.3f ms per locus" % ((time.perf_counter() - t0) / n * 1e3))
