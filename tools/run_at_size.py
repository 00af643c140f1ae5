"""BASELINE.json configs[3] and configs[4] at their stated sizes on ONE GPU, through the product CLI (GPU box):

  cfg4  `vapor vcf` on a seeded synthetic VCF of 10 000 records: DEL / INV / INS and the complex types DUP_INV, DISDUP,
        DEL_INV and Other= (vapor_vali/vapor:368-466), 15 kb reads, 40 reads per locus
  cfg5  `vapor bed` on a seeded synthetic BED of 50 000 loci: DEL / DUP / INV / INS, 30 kb reads, 60 reads per locus
        (vapor_vali/vapor:334-367; the drivers keep the 20 reads with the smallest miss_bp, SF:1091-1102)

The world is built once for `--base` distinct loci (every locus a contig of its own with its reads) and tiled under alias
contig names up to the stated number of loci: a tile is the same work again - window and read extraction, allele strings,
window refinement, dot plots, scores, rows - without the minutes of read synthesis a world of 3 million 30 kb reads takes.
Figures off (the reference's PNG per locus is measured by tools/fig_rate.py); VAPOR_QC_SEED fixes the X-means of the repeat
check so that the rows can be compared.  Writes loci/s, peak host RSS and a sample of the output rows to --out;
tests/at_size_check.py replays the base world through the CPU twin and compares every sampled row.

usage: python tools/run_at_size.py cfg4|cfg5 [--loci N] [--base B] [--out file.json]"""
import hashlib
import json
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import cli, seqio, synth

# the complex records of a VCF that reach a scorer in the reference (the defects it dies of - str > int for a DISDUP with
# spanning reads, SF:1801; blocks >= 100 bp apart, SF:1585 - are pinned by tests/golden/locus_complex.json.gz, not run here)
CX = [dict(type="DUP_INV", a=200, gap=1400), dict(type="DUP_INV", a=260, gap=1900), dict(type="DUP_INV", a=180, gap=-1300),
      dict(type="DUP_INV", a=320, gap=10400), dict(type="DUP_INV", a=280, xchrom=1000),
      dict(type="DISDUP", a=300, gap=11000), dict(type="DISDUP", a=260, xchrom=900), dict(type="DISDUP", a=260, xchrom=15000),
      dict(type="DEL_INV", a=300, b=420), dict(type="DEL_INV", a=350, b=300, order="inv,del"), dict(type="DEL_INV", a=700, b=9600),
      dict(type="OTHER", a=420, b=520, other=("ab/ab", "b/b^")), dict(type="OTHER", a=380, b=460, other=("ab/ab", "a/ab")),
      dict(type="OTHER", a=300, b=2400, other=("ab/ab", "aba/ab")), dict(type="OTHER", a=400, b=500, other=("ab/ab", "ba^/ab"))]

SPEC = {
    "cfg4": dict(mode="vcf", loci=10000, base=300, n_reads=40, read_len=15000, seed=404),
    "cfg5": dict(mode="bed", loci=50000, base=250, n_reads=60, read_len=30000, seed=505),
}


def base_world(cfg, base):
    sp = SPEC[cfg]
    if cfg == "cfg5":
        w = synth.make_world(seed=sp["seed"], n_loci=base, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), span_range=(50, 11000),
                             read_len=sp["read_len"], n_reads=sp["n_reads"])
        return w, None
    n_cx = base // 3                                         # a third complex records
    w = synth.make_world(seed=sp["seed"], n_loci=base - n_cx, svtypes=("DEL", "INV", "INS", "DEL"), span_range=(60, 11000),
                         read_len=sp["read_len"], n_reads=sp["n_reads"])
    rng = np.random.default_rng(sp["seed"] + 1)
    specs = []
    for t in range(n_cx):
        s = dict(CX[t % len(CX)])
        s["a"] = int(s["a"] * rng.uniform(0.9, 1.3))         # sizes vary from record to record
        specs.append(s)
    cx = synth.make_complex_world(sp["seed"] + 2, specs, n_reads=sp["n_reads"])
    return w, cx


def tile(w, n_total, text_of):
    """Alias contigs `<name>_t<k>` sharing the base strings and record lists; returns the tiled world and its input text."""
    base_names = list(w.contigs)
    per = len(w.loci)
    big = synth.SynthWorld()
    lines = []
    k = 0
    while len(big.loci) < n_total:
        sub = synth.SynthWorld()
        for c in base_names:
            big.contigs["%s.t%d" % (c, k)] = w.contigs[c]
            big.reads["%s.t%d" % (c, k)] = w.reads.get(c, [])
        for l in w.loci[:n_total - len(big.loci)]:
            extra = dict(l.extra) if l.extra else None
            if extra and "insert_chrom" in extra:
                extra["insert_chrom"] = "%s.t%d" % (extra["insert_chrom"], k)
            sub.loci.append(synth.Locus("%s.t%d" % (l.chrom, k), l.svtype, l.start, l.end, "%s.t%d" % (l.svid, k), l.ins_seq, extra))
        big.loci += sub.loci
        lines.append(text_of(sub))
        k += 1
    return big, "".join(lines), per


def main():
    cfg = sys.argv[1]
    sp = SPEC[cfg]
    arg = lambda name, d: type(d)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
    n_total, base = arg("--loci", sp["loci"]), arg("--base", sp["base"])
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "r3_%s_at_size.json" % cfg))
    os.environ.setdefault("VAPOR_QC_SEED", "7")
    os.environ["VAPOR_TIMING"] = "1"
    t0 = time.perf_counter()
    w, cx = base_world(cfg, base)
    t_world = time.perf_counter() - t0
    tmp = tempfile.mkdtemp(prefix="vapor_at_size_")
    if cfg == "cfg5":
        big, text, per = tile(w, n_total, synth.bed_text)
        src = os.path.join(tmp, "in.bed")
    else:
        n_cx_total = n_total // 3
        b1, t1, _ = tile(w, n_total - n_cx_total, lambda s: synth.vcf_text(s, header=False))
        b2, t2, _ = tile(cx, n_cx_total, lambda s: synth.complex_vcf_text(s, header=False))
        big = b1
        big.contigs.update(b2.contigs); big.reads.update(b2.reads); big.loci += b2.loci
        text = t1 + t2
        src = os.path.join(tmp, "in.vcf")
    open(src, "w").write(text)
    n_records = text.count("\n")
    seqio.set_backend(seqio.MemorySamtools(big))
    result = os.path.join(tmp, "out.vapor") if cfg == "cfg5" else src + ".vapor"
    argv = [sp["mode"], "--sv-input", src, "--reference", "ref.fa", "--pacbio-input", "x.bam", "--output-path", os.path.join(tmp, "figs"),
            "--output-file", result, "--no-figures"]
    print("%s: %d records (%d distinct loci tiled; world in %.1f s), running `vapor %s`" % (cfg, n_records, base, t_world, sp["mode"]), flush=True)
    devnull = open(os.devnull, "w")
    real_stdout = sys.stdout
    sys.stdout = devnull                                   # (the CLI prints every result list, as the reference does)
    t0 = time.perf_counter()
    try:
        if "--profile" in sys.argv:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            rc = pr.runcall(cli.main, argv)
            sys.stdout = real_stdout
            pstats.Stats(pr).sort_stats("tottime").print_stats(22)
            pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
        else:
            rc = cli.main(argv)
    finally:
        sys.stdout = real_stdout
    dt = time.perf_counter() - t0
    rows = open(result).read().splitlines()
    body = rows[1:] if rows and rows[0].startswith("#CHR") else rows
    rss_mb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
    rng = np.random.default_rng(1)
    pick = list(range(len(body))) if "--all-rows" in sys.argv else sorted(rng.choice(len(body), size=min(400, len(body)), replace=False).tolist())
    scored = sum(1 for r in body if "\tNA\t" not in r and not r.endswith("\tNA"))
    rec = {"config": cfg, "mode": sp["mode"], "records": n_records, "rows": len(body), "rows_with_scores": scored, "base_loci": base,
           "n_reads_per_locus": sp["n_reads"], "read_len": sp["read_len"], "seed": sp["seed"], "seconds": round(dt, 3),
           "loci_per_s": round(n_records / dt, 1), "world_seconds": round(t_world, 1), "peak_host_rss_mb": round(rss_mb, 1),
           "rows_sha256": hashlib.sha256("\n".join(body).encode()).hexdigest(), "qc_seed": os.environ["VAPOR_QC_SEED"],
           "sample": [[t, body[t]] for t in pick], "rc": rc,
           "note": "one process, one GPU, in-memory world, figures off; tiles repeat the base loci under alias contig names"}
    json.dump(rec, open(out, "w"))
    print(json.dumps({k: v for k, v in rec.items() if k != "sample"}), flush=True)


if __name__ == "__main__":
    main()
