"""Prices cli.job_cost's three terms on the GPU box: worlds of ONE type and ONE span each go through
the product path (cli.bed_jobs -> score_jobs, in-memory world, one process, figures off), the time per locus is measured
(best of three) and fitted by non-negative least squares (on relative errors) to

    host_us + per_kbase_us * (20 * Lr + La) / 1000 + per_gcell_us * 20 * Lr * La / 1e9 + xmeans_us * [tandem duplication]

with Lr / La the windows cli.job_cost derives from type and span.  Writes profiles-style JSON to --out (default
gpurun_out/r04_job_cost_fit.json): the measurements, the fitted coefficients, the constants cli.py carries, and how evenly
LPT shares of a mixed world come out under either.

usage: python tools/fit_job_cost.py [--loci 240] [--out file.json]"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import cli, dist, seqio, synth


def windows(svtype, span):
    """(Lr, La_ref + La_alt) exactly as cli.job_cost cuts them (kept in step by tests/test_dist_gloo.py)."""
    saved = (cli.COST_HOST_US, cli.COST_PER_KBASE_US, cli.COST_PER_GCELL_US, cli.COST_XMEANS_US)
    try:
        cli.COST_HOST_US, cli.COST_PER_KBASE_US, cli.COST_PER_GCELL_US, cli.COST_XMEANS_US = 0.0, 1.0, 0.0, 0.0
        bases = cli.job_cost(svtype, span) * 1e3            # 20 * Lr + La
        cli.COST_PER_KBASE_US, cli.COST_PER_GCELL_US = 0.0, 1.0
        cells = cli.job_cost(svtype, span) * 1e9            # 20 * Lr * La
    finally:
        cli.COST_HOST_US, cli.COST_PER_KBASE_US, cli.COST_PER_GCELL_US, cli.COST_XMEANS_US = saved
    return bases, cells


def rate(svtype, span, n_loci, read_len):
    w = synth.make_world(seed=900 + span, n_loci=n_loci, svtypes=(svtype,), span_range=(span, span), read_len=read_len, n_reads=20,
                         ins_len_range=(span, span))
    tmp = tempfile.mkdtemp(prefix="vapor_fit_")
    bed = os.path.join(tmp, "in.bed")
    open(bed, "w").write(synth.bed_text(w))
    seqio.set_backend(seqio.MemorySamtools(w))
    try:
        info = cli.bed_info_readin(bed, tmp)
        best = 1e9
        scored = 0
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(4):
                t0 = time.perf_counter()
                jobs = cli.bed_jobs(info, 3, "x.bam", "ref.fa", tmp + "/", "s")
                scores = cli.score_jobs(jobs, 2048, None)
                best = min(best, time.perf_counter() - t0)
                scored = sum(1 for s in scores if s)
    finally:
        seqio.set_backend(None)
    return best / n_loci * 1e6, scored


def main():
    arg = lambda name, d: type(d)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else d
    n_loci = arg("--loci", 240)
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "r04_job_cost_fit.json"))
    cases = [("DEL", 100), ("DEL", 1000), ("DEL", 8000), ("INV", 200), ("INV", 2000), ("INV", 6000), ("INV", 9500),
             ("TANDUP", 300), ("TANDUP", 2500), ("TANDUP", 7000), ("INS", 300), ("INS", 2000)]
    os.environ.setdefault("VAPOR_QC_SEED", "7")
    rows = []
    for t, span in cases:
        lr_need = {"DEL": 1500, "INV": span + 1500, "TANDUP": 2 * span + 1500, "INS": span + 1500}[t]
        us, scored = rate(t, span, n_loci, max(2500, lr_need))
        b, c = windows(t, span)
        rows.append({"type": t, "span": span, "us_per_locus": round(us, 2), "bases": b, "cells": c, "loci_with_scores": scored})
        print(rows[-1], flush=True)
    # four non-negative terms: the three of every locus, and what the X-means of a tandem duplication's alt window adds
    from scipy.optimize import nnls
    A = np.array([[1.0, r["bases"] / 1e3, r["cells"] / 1e9, 1.0 if r["type"] == "TANDUP" else 0.0] for r in rows])
    y = np.array([r["us_per_locus"] for r in rows])
    coef, _res = nnls(A / y[:, None], np.ones(len(y)))       # (relative errors: a 30 us locus counts like a 700 us one)
    pred = A @ coef
    carried = np.array([cli.COST_HOST_US, cli.COST_PER_KBASE_US, cli.COST_PER_GCELL_US, cli.COST_XMEANS_US])
    # how evenly the shares of a mixed world come out: true time = measured, estimate = fitted / carried / unit costs
    rng = np.random.default_rng(3)
    pick = rng.integers(0, len(rows), size=4000)
    true = y[pick]

    def spread(est, nw):
        parts = dist.partition(list(est), nw)
        loads = [float(true[p].sum()) for p in parts]
        return round(max(loads) / (sum(loads) / nw), 4)
    rec = {"measurements": rows,
           "fit": {"host_us": round(float(coef[0]), 2), "per_kbase_us": round(float(coef[1]), 3), "per_gcell_us": round(float(coef[2]), 3),
                   "xmeans_us": round(float(coef[3]), 1),
                   "max_rel_error": round(float(np.max(np.abs(pred - y) / y)), 3)},
           "carried_by_cli": {"host_us": cli.COST_HOST_US, "per_kbase_us": cli.COST_PER_KBASE_US, "per_gcell_us": cli.COST_PER_GCELL_US, "xmeans_us": cli.COST_XMEANS_US,
                              "max_rel_error": round(float(np.max(np.abs(A @ carried - y) / y)), 3)},
           "share_spread_max_over_mean": {str(nw): {"fitted": spread(pred[pick], nw), "carried": spread((A @ carried)[pick], nw),
                                                    "by_count": spread(np.ones(len(pick)), nw)} for nw in (2, 4, 8)},
           "note": "one process, one GPU, in-memory worlds of %d loci x 20 reads, figures off, best of 4; the spread is the most loaded "
                   "rank's measured time over the mean when 4 000 loci drawn from these cases are shared out by LPT on each estimate" % n_loci}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "measurements"}))


if __name__ == "__main__":
    main()
