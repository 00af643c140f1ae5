"""Where the files path's extraction spends its time (host only: no GPU needed).  Writes a seeded world's FASTA/BAM, then times
InProcessBam.chop_many over all loci by prefetch-thread count, and splits one single-threaded sweep into the native reader's calls
(vapor_bam_chop: index chunks -> pread -> inflate -> CIGAR walk -> kept bases) and the Python around them.
  python tools/files_extract_probe.py [n_loci]"""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import seqio, synth, bamio, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
for c in w.reads:
    w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
tmp = tempfile.mkdtemp()
t0 = time.perf_counter()
fa, bam = synth.write_world_files(w, tmp, block_size=0xFF00)
print("files of %d loci in %.1f s (%.1f MB BAM), %d usable cores" % (n, time.perf_counter() - t0, os.path.getsize(bam) / 1e6, len(os.sched_getaffinity(0))), flush=True)
be = seqio.InProcessSamtools() if hasattr(seqio, "InProcessSamtools") else None
backend = seqio.InProcessBam() if be is None else be
loci = [(sv.chrom, sv.start, sv.end) for sv in w.svs] if hasattr(w, "svs") else None
if loci is None:
    rows = [l.split("\t") for l in synth.bed_text(w).strip().splitlines()]
    loci = [(r[0], int(r[1]), int(r[2])) for r in rows]
chroms = [l[0] for l in loci]; st = [l[1] for l in loci]; en = [l[2] for l in loci]; fl = [500] * len(loci)
for thr in ("1", "2", "4", "8"):
    os.environ["VAPOR_PREFETCH_THREADS"] = thr
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); out = backend.chop_many(bam, chroms, st, en, fl); best = min(best, time.perf_counter() - t0)
    print("chop_many, %s prefetch thread(s): %.3f s -> %.0f loci/s (%d reads kept)" % (thr, best, len(loci) / best, int(out[0][-1])), flush=True)
# one thread: native calls against the Python around them
os.environ["VAPOR_PREFETCH_THREADS"] = "1"
lib = _lib.load()
spent = [0.0, 0]
orig = bamio.BamReader._chop_with if hasattr(bamio, "BamReader") else None
cls = next(c for c in vars(bamio).values() if isinstance(c, type) and hasattr(c, "_chop_with"))
orig = cls._chop_with
real = lib.vapor_bam_chop
class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); spent[0] += time.perf_counter() - t0; spent[1] += 1; return r
def patched(self, lib_, *a, **k):
    class L:                                   # the library with one entry timed
        def __getattr__(s, name): return Timed() if name == "vapor_bam_chop" else getattr(lib_, name)
    return orig(self, L(), *a, **k)
cls._chop_with = patched
t0 = time.perf_counter(); backend.chop_many(bam, chroms, st, en, fl); tot = time.perf_counter() - t0
cls._chop_with = orig
print("one thread: %.3f s for %d loci = %.3f ms a locus; inside vapor_bam_chop %.3f s (%d calls, %.3f ms each), Python around it %.3f ms a locus" % (
    tot, len(loci), tot / len(loci) * 1e3, spent[0], spent[1], spent[0] / max(spent[1], 1) * 1e3, (tot - spent[0]) / len(loci) * 1e3), flush=True)
pr = cProfile.Profile(); pr.runcall(backend.chop_many, bam, chroms, st, en, fl)
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
