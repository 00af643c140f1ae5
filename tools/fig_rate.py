"""Rate of a `vapor bed` run WITH its recurrence-plot PNGs (the reference's default; SF:1072-1089) on an in-memory world.
usage: fig_rate.py [n_loci]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vapor_amd import cli, figures, pipeline, seqio, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
tmp = tempfile.mkdtemp()
bed = os.path.join(tmp, "in.bed")
open(bed, "w").write(synth.bed_text(w))
seqio.set_backend(seqio.MemorySamtools(w))
bed_info = cli.bed_info_readin(bed, tmp)
pipeline.get_engine()
_spent = [0.0]
_orig = figures.figure_specs


def _timed(reqs, engine=None):
    t0 = time.perf_counter()
    try:
        return _orig(reqs, engine)
    finally:
        _spent[0] += time.perf_counter() - t0


figures.figure_specs = _timed
for tag, fn in (("without figures", None), ("with figures", figures.make_event_figure_1), ("with figures", figures.make_event_figure_1)):
    jobs = cli.bed_jobs(bed_info, 3, "x.bam", "ref.fa", tmp + "/", "s")
    t0 = time.perf_counter()
    cli.score_jobs(jobs, 2048, fn)
    if hasattr(figures, "wait"):
        figures.wait()
    dt = time.perf_counter() - t0
    pngs = len([f for f in os.listdir(tmp) if f.endswith(".png")])
    print("%-16s %d loci in %.2f s -> %.1f loci/s (%d PNGs in %s; %.2f s in figure_specs on the main thread)" % (tag, n, dt, n / dt, pngs, tmp, _spent[0]), flush=True)
    _spent[0] = 0.0
