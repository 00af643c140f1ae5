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

usage: python tools/run_at_size.py cfg4|cfg5 [--loci N] [--base B] [--spans simulate] [--distinct] [--out file.json]"""
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
from vapor_amd import cli, pipeline, seqio, synth

from vapor_amd.workload import AT_SIZE as SPEC, at_size_base_world as base_world, at_size_tile as tile  # noqa: E402


def main():
    cfg = sys.argv[1]
    sp = SPEC[cfg]
    arg = lambda name, d: type(d)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
    n_total, base = arg("--loci", sp["loci"]), arg("--base", sp["base"])
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "r3_%s_at_size.json" % cfg))
    os.environ.setdefault("VAPOR_QC_SEED", "7")
    os.environ["VAPOR_TIMING"] = "1"
    t0 = time.perf_counter()
    from vapor_amd.workload import at_size_input
    distinct = "--distinct" in sys.argv
    span_dist = arg("--spans", "") or None                 # "simulate": spans drawn from the reference's simulated truth sets
    big, text, n_records = at_size_input(cfg, n_total, base, distinct=distinct, cap=arg("--lru", 4096), span_dist=span_dist)
    spans = [l.end - l.start for l in big.loci if l.svtype != "INS"]
    if "--only" in sys.argv:
        # (tests/at_size_check.py: the sampled records of a distinct world alone, in their order)
        only = json.load(open(sys.argv[sys.argv.index("--only") + 1]))
        lines = text.splitlines(True)
        text = "".join(lines[t] for t in only)
        n_records = len(only)
    t_world = time.perf_counter() - t0
    tmp = tempfile.mkdtemp(prefix="vapor_at_size_")
    src = os.path.join(tmp, "in.bed" if cfg == "cfg5" else "in.vcf")
    open(src, "w").write(text)
    seqio.set_backend(seqio.MemorySamtools(big))
    result = os.path.join(tmp, "out.vapor") if cfg == "cfg5" else src + ".vapor"
    argv = [sp["mode"], "--sv-input", src, "--reference", "ref.fa", "--pacbio-input", "x.bam", "--output-path", os.path.join(tmp, "figs"),
            "--output-file", result, "--no-figures", "--chunk", str(arg("--chunk", 2048))]
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
    body = [r for r in rows if r and not r.startswith("#")]         # (the table's header line; the ##INFO lines of the rewritten VCF)
    rss_mb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
    rng = np.random.default_rng(1)
    n_sample = arg("--sample", 2000 if distinct else 400)
    pick = list(range(len(body))) if "--all-rows" in sys.argv else sorted(rng.choice(len(body), size=min(n_sample, len(body)), replace=False).tolist())
    scored = sum(1 for r in body if "\tNA\t" not in r and not r.endswith("\tNA"))
    rec = {"config": cfg, "mode": sp["mode"], "records": n_records, "rows": len(body), "rows_with_scores": scored, "base_loci": base,
           "n_reads_per_locus": sp["n_reads"], "read_len": sp["read_len"], "seed": sp["seed"], "seconds": round(dt, 3),
           "loci_per_s": round(n_records / dt, 1), "world_seconds": round(t_world, 1), "peak_host_rss_mb": round(rss_mb, 1),
           "rows_sha256": hashlib.sha256("\n".join(body).encode()).hexdigest(), "qc_seed": os.environ["VAPOR_QC_SEED"],
           "sample": [[t, body[t]] for t in pick], "rc": rc,
           # windows that met the reference's unseeded X-means: `one_cluster` outcomes are the reference's
           # answer under any seed; `sizes_decide` counts the windows where the cluster sizes could change the window size at all
           "xmeans_windows": dict(pipeline.qc_counts),
           "distinct": distinct, "distinct_loci": (n_records if distinct else base),
           "span_dist": span_dist or "uniform",
           "spans": {"median": float(np.median(spans)) if spans else None, "max": int(max(spans)) if spans else None,
                     "frac_ge_10kb": round(float(np.mean(np.asarray(spans) >= 10000)), 4) if spans else None},
           "note": ("one process, one GPU, figures off; every tile of the base world carries its own substitutions in contigs, reads and "
                    "insertion payloads and is made when a chunk reaches it (synth.DistinctTilesWorld): as many distinct loci as "
                    "records; the time includes making them" if distinct else
                    "one process, one GPU, in-memory world, figures off; tiles repeat the base loci under alias contig names")}
    if distinct:
        parts = getattr(big, "parts", [big])
        rec["tile_generation"] = {"contigs_made": sum(p_.contigs.made for p_ in parts), "read_lists_made": sum(p_.reads.made for p_ in parts),
                                  "seconds_inside_the_run": round(sum(p_.contigs.seconds + p_.reads.seconds for p_ in parts), 2)}
    json.dump(rec, open(out, "w"))
    print(json.dumps({k: v for k, v in rec.items() if k != "sample"}), flush=True)


if __name__ == "__main__":
    main()
