"""Host-side cost of the driver pipeline WITHOUT a device (development container): an engine whose plans answer with constant
statistics and scores, so that what is timed is everything the Python side does per locus - BED parsing, the driver
generators, read extraction from the in-memory world, the assembly of sequence sets / pair tables / read tables, result rows.
Not a product path and not a measurement of the product: a profiler's harness.  usage: python tools/prof_host.py [n_loci] [--svtypes A,B] [--read-len L] [--reads R] [--files] [--profile]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import _lib as L
from vapor_amd import cli, pipeline, seqio, synth
from vapor_amd import simple_function as SF
from vapor_amd.finish import result_organize_ins


class _SS:
    def __init__(self, seqs, derived=None):
        self.n = len(seqs) + len(derived or ())
        self.lens = np.concatenate([np.fromiter(map(len, seqs), dtype=np.int32, count=len(seqs)),
                                    np.asarray([sum(g[2] for g in sg) for sg, _u in (derived or ())], dtype=np.int32)])
        self.n_invalid = np.zeros(self.n, dtype=np.int32)
        self.n_exc = np.zeros(self.n, dtype=np.int32)

    def close(self):
        pass


class _Plan:
    def __init__(self, ss, pairs):
        self.n = len(pairs)

    def run(self):
        st = np.zeros((max(self.n, 1), 16), dtype=np.int64)
        st[:, 0] = 1000; st[:, 7] = 1000
        return st[:self.n]

    def set_reads(self, reads, n_loci):
        self.read_scores = np.full(max(len(reads), 1), 0.5)

    def run_loci(self, device_out=0, want_host=True, want_scores=False):
        return None

    def fetch_hits(self, idx, want_flags=True):
        idx = list(idx)
        return np.zeros((0, 2), np.int32), None, np.zeros(len(idx) + 1, dtype=np.int64)

    def close(self):
        pass


class NullEngine:
    def seqset(self, seqs, upper=None, derived=None):
        return _SS(seqs, derived)

    def seqset_raw(self, addr, lens, derived=None, keepalive=None):
        ss = _SS.__new__(_SS)
        nd = (len(derived[0]) - 1) if derived is not None else 0
        ss.n = len(addr) + nd
        dl = np.zeros(nd, dtype=np.int32)
        if nd:
            np.add.at(dl, np.repeat(np.arange(nd), np.diff(derived[0])), derived[1]["len"][:int(derived[0][-1])])
        ss.lens = np.concatenate([np.asarray(lens, dtype=np.int32), dl])
        ss.n_invalid = np.zeros(ss.n, dtype=np.int32)
        ss.n_exc = np.zeros(ss.n, dtype=np.int32)
        return ss

    def plan(self, ss, pairs):
        return _Plan(ss, pairs)


args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else 2000
svtypes = tuple(args[args.index("--svtypes") + 1].split(",")) if "--svtypes" in args else ("DEL", "DEL", "INV", "INS")
t0 = time.perf_counter()
read_len = int(args[args.index("--read-len") + 1]) if "--read-len" in args else 9500
n_reads = int(args[args.index("--reads") + 1]) if "--reads" in args else 20
w = synth.make_world(seed=11, n_loci=n, svtypes=svtypes, span_range=(100, 4000), read_len=read_len, n_reads=n_reads)
print("world of %d loci in %.1fs" % (n, time.perf_counter() - t0), flush=True)
tmp = tempfile.mkdtemp()
bed = os.path.join(tmp, "in.bed")
open(bed, "w").write(synth.bed_text(w))
bam_path, fa_path = "x.bam", "ref.fa"
if "--files" in args:                       # FASTA/BAM files through the in-process readers instead of the in-memory world
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa_path, bam_path = synth.write_world_files(w, tmp, block_size=0xFF00)
    seqio.set_backend(seqio.InProcessBam())
else:
    seqio.set_backend(seqio.MemorySamtools(w))
bed_info = cli.bed_info_readin(bed, tmp)
pipeline._engine = NullEngine()


def run():
    jobs = cli.bed_jobs(bed_info, 3, bam_path, fa_path, tmp + "/", "s")
    scores = cli.score_jobs(jobs, 2048, None)
    return cli.output_rows([j.key.split(':') + [j.row_prefix] for j in jobs], scores)[0]          # (as cli.main writes them)


run()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); rows = run(); best = min(best, time.perf_counter() - t0)
print("%d loci in %.3f s -> %.1f loci/s of host work alone (best of 3; %.1f us per locus)" % (len(rows), best, len(rows) / best, best / len(rows) * 1e6))
if "--profile" in args:
    cProfile.run("run()", "/tmp/host.prof")
    pstats.Stats("/tmp/host.prof").sort_stats("tottime").print_stats(28)
