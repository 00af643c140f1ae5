import os, sys, tempfile, time, cProfile, pstats
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from vapor_amd import cli, pipeline, seqio, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
for c in w.reads:
    w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
tmp = tempfile.mkdtemp()
fa, bam = synth.write_world_files(w, tmp, block_size=0xFF00)
bed = os.path.join(tmp, "in.bed"); open(bed, "w").write(synth.bed_text(w))
def run(tag):
    return cli.main(["bed", "--sv-input", bed, "--reference", fa, "--pacbio-input", bam, "--output-path", tmp + "/f", "--output-file", tmp + "/o%s.vapor" % tag] + ([] if os.environ.get("VAPOR_PROF_FIGURES") else ["--no-figures"]) + (["--chunk", os.environ["VAPOR_PROF_CHUNK"]] if os.environ.get("VAPOR_PROF_CHUNK") else []))
if len(sys.argv) > 2 and sys.argv[2] == "--grid":
    import contextlib, io
    for chunk, fl in ((2048, 2), (1024, 2), (1024, 3), (512, 2), (512, 3), (512, 4), (256, 4)):
        os.environ["VAPOR_PROF_CHUNK"] = str(chunk)
        os.environ["VAPOR_CHUNKS_IN_FLIGHT"] = str(fl)
        best = 1e9
        for _ in range(4):
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter(); run("g"); best = min(best, time.perf_counter() - t0)
        print("chunk %4d, %d in flight: %.3f s -> %.0f loci/s" % (chunk, fl, best, n / best), file=sys.stderr, flush=True)
    sys.exit(0)
if os.environ.get("VAPOR_PROF_SWITCH"):
    sys.setswitchinterval(float(os.environ["VAPOR_PROF_SWITCH"]))
if os.environ.get("VAPOR_PROF_FIRST"):                   # where the first (cold) run of a process spends its time
    import contextlib, io
    pr = cProfile.Profile()
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); pr.runcall(run, "a"); dt = time.perf_counter() - t0
    print("first run: %d loci in %.3f s" % (n, dt), file=sys.stderr)
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(16)
    st.sort_stats("cumulative").print_stats(40)
    sys.exit(0)
run("a")
t0 = time.perf_counter(); run("b"); dt = time.perf_counter() - t0
print("in-process files run: %d loci in %.3f s -> %.1f loci/s" % (n, dt, n / dt))
cProfile.run('run("c")', "/tmp/pf.prof")
pstats.Stats("/tmp/pf.prof").sort_stats("tottime").print_stats(18)
if len(sys.argv) > 2 and sys.argv[2] == "--sweep":
    for thr, inf in ((8, 1), (8, 2), (12, 1), (12, 2), (16, 1), (12, 4)):
        os.environ["VAPOR_PREFETCH_THREADS"] = str(thr)
        os.environ["VAPOR_BAM_INFLATE_THREADS"] = str(inf)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run("s"); best = min(best, time.perf_counter() - t0)
        print("prefetch threads %2d x %d inflate: %.3f s -> %.0f loci/s" % (thr, inf, best, n / best), file=sys.stderr)
if len(sys.argv) > 2 and sys.argv[2] == "--soak":
    import resource
    for i in range(12):
        t0 = time.perf_counter(); run("k"); dt = time.perf_counter() - t0
        print("soak run %2d: %.3f s, %d open descriptors, max RSS %.0f MB" % (i, dt, len(os.listdir("/proc/self/fd")),
              resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), file=sys.stderr)
