"""The files path (FASTA + BAM through the product CLI, in one warm process) with the read extraction on the device
(vapor_bam_chop_device) against the host's (vapor_bam_chop on the prefetch threads): the two output tables byte for byte, and
loci/s of each.
  python tools/files_ab.py [n_loci] [block_size] [chunk] [--qual] [--repeat R]      (--qual: seeded qualities instead of 0xFF)"""
import contextlib, hashlib, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vapor_amd import cli, synth

argv = sys.argv[1:]
rep = 1
if "--repeat" in argv:
    k = argv.index("--repeat")
    rep = int(argv[k + 1])
    del argv[k:k + 2]
args = [a for a in argv if not a.startswith("--")]
n = int(args[0]) if args else 2000
block = int(args[1], 0) if len(args) > 1 else 0xFF00
chunk = args[2] if len(args) > 2 else None
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
for c in w.reads:
    w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
tmp = tempfile.mkdtemp()
fa, bam = synth.write_world_files(w, tmp, block_size=block, qual_seed=(7 if '--qual' in sys.argv else None))
bed = os.path.join(tmp, "in.bed")
open(bed, "w").write(synth.bed_text(w) * rep)               # (--repeat R: the same loci R times over - a long run from a short file)
n *= rep
print("files of %d loci: %.1f MB BAM, blocks of %d, %d usable cores" % (n, os.path.getsize(bam) / 1e6, block, len(os.sched_getaffinity(0))), flush=True)


def run(tag):
    out = tmp + "/o%s.vapor" % tag
    args = ["bed", "--sv-input", bed, "--reference", fa, "--pacbio-input", bam, "--output-path", tmp + "/f", "--output-file", out, "--no-figures"]
    if chunk:
        args += ["--chunk", chunk]
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        rc = cli.main(args)
        dt = time.perf_counter() - t0
    assert rc in (0, None), rc
    return dt, hashlib.sha256(open(out, "rb").read()).hexdigest()[:16], sum(1 for _ in open(out))


res = {}
for name, dev in (("device", "1"), ("host", "0"), ("device", "1"), ("host", "0")):
    os.environ["VAPOR_BAM_DEVICE"] = dev
    run(name)                                            # (warm: engines, pools, page cache)
    best, sha, rows = 1e9, None, 0
    for _ in range(3):
        dt, sha, rows = run(name)
        best = min(best, dt)
    res.setdefault(name, []).append((best, sha, rows))
    print("%-6s extraction: %d loci in %.3f s -> %.0f loci/s; table %s (%d rows)" % (name, n, best, n / best, sha, rows), flush=True)
if "--child" in sys.argv:
    # the same run as a child process (its start, imports, contexts and first launches included), device and host extraction
    import subprocess
    for name, dev in (("device", "1"), ("host", "0")):
        out = tmp + "/child_%s.vapor" % name
        cmd = [sys.executable, "-m", "vapor_amd.cli", "bed", "--sv-input", bed, "--reference", fa, "--pacbio-input", bam, "--output-path", tmp + "/fc",
               "--output-file", out, "--no-figures"]
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            subprocess.run(cmd, env=dict(os.environ, VAPOR_BAM_DEVICE=dev, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            best = min(best, time.perf_counter() - t0)
        sha = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
        print("%-6s extraction as a child process: %d loci in %.3f s -> %.0f loci/s; table %s" % (name, n, best, n / best, sha), flush=True)
if "--profile" in sys.argv:
    import cProfile, pstats
    os.environ["VAPOR_BAM_DEVICE"] = "1"
    pr = cProfile.Profile()
    pr.runcall(run, "p")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
shas = {v[1] for vs in res.values() for v in vs}
print("tables equal: %s" % (len(shas) == 1), flush=True)
