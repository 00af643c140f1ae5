#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE implementation on seeded inputs.

TEST INFRASTRUCTURE - runs only in the development container, where
/root/reference is mounted.  It imports the authoritative source
(/root/reference/vapor_vali/Simple_function.pyx, plain Python despite the suffix;
SURVEY.md §0.2, §8c) by path, replaces its two samtools pipes with an in-memory
world built by this repo's own generator (vapor_amd.synth), and writes inputs
together with the reference's outputs to tests/golden/*.json.gz.

Nothing from the reference is copied: fixtures hold data only (sequences made
here, numbers the reference returned).  Re-run:  python oracle/gen_golden.py
"""
from __future__ import annotations

import gzip
import hashlib
import importlib.machinery
import importlib.util
import io
import json
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vapor_amd import synth  # noqa: E402
from vapor_amd.seqio import MemorySamtools  # noqa: E402

REF_SF = "/root/reference/vapor_vali/Simple_function.pyx"
REF_CLI = "/root/reference/vapor_vali/vapor"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    ld = importlib.machinery.SourceFileLoader("vapor_ref_sf", REF_SF)
    spec = importlib.util.spec_from_loader("vapor_ref_sf", ld)
    m = importlib.util.module_from_spec(spec)
    ld.exec_module(m)
    return m


class _ShimPath:
    def __getattr__(self, k):
        return getattr(os.path, k)

    @staticmethod
    def isfile(p):
        return str(p).endswith(".bam") or os.path.isfile(p)


class ShimOS:
    """Stands in for the `os` name inside the reference module: popen answers the two
    samtools commands from memory, everything else is the real os."""

    def __init__(self, world):
        self.be = MemorySamtools(world)
        self.path = _ShimPath()
        self.calls = 0

    def __getattr__(self, k):
        return getattr(os, k)

    def popen(self, cmd):
        self.calls += 1
        f = cmd.split()
        assert f[0] == "samtools", cmd
        if f[1] == "faidx":
            lines = self.be.faidx_lines(f[2], f[3])
        elif f[1] == "view":
            lines = self.be.view_lines(f[2], f[3])
        else:
            raise AssertionError(cmd)
        return io.StringIO("".join(l + "\n" for l in lines))

    def system(self, cmd):
        if cmd.startswith("mkdir"):
            os.makedirs(cmd.split()[-1], exist_ok=True)
            return 0
        raise AssertionError(cmd)


def jsonable(x):
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (list, tuple)):
        return [jsonable(i) for i in x]
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, float) and math.isnan(x):
        return "nan"
    return x


def dump(name, obj):
    p = os.path.join(OUT, name)
    raw = json.dumps(jsonable(obj), separators=(",", ":")).encode()
    with gzip.GzipFile(p, "wb", mtime=0) as f:
        f.write(raw)
    print("wrote %s: %d cases, %.1f KB" % (name, len(obj["cases"]) if "cases" in obj else -1,
                                           os.path.getsize(p) / 1024))


def hits_digest(hits):
    a = np.asarray(hits, dtype=np.int32).reshape(-1, 2)
    return hashlib.sha256(a.tobytes()).hexdigest()


def call(fn, *a):
    try:
        return {"ok": fn(*a)}
    except Exception as e:  # noqa: BLE001 - the reference's failure mode is part of the vector
        return {"error": type(e).__name__}


# ---------------------------------------------------------------------------
def gen_kmerhits(m):
    rng = np.random.default_rng(101)
    cases = []

    def add(name, k, s1, s2, full=None):
        r = call(m.dotdata, k, s1, s2)
        c = {"name": name, "k": k, "seq1": s1, "seq2": s2}
        if "error" in r:
            c["error"] = r["error"]
        else:
            h = r["ok"]
            c["n_hits"] = len(h)
            c["sha256"] = hits_digest(h)
            if full or (full is None and len(h) <= 1500):
                c["hits"] = [list(t) for t in h]
        cases.append(c)

    kat = "ACGTACGTACGTTTGACCA"
    add("kat_self", 10, kat, kat)
    add("empty_both", 10, "", "")
    add("short_seq1", 10, "ACGTACG", synth.random_dna(rng, 50))
    add("short_seq2", 10, synth.random_dna(rng, 50), "ACGTACG")
    add("len_eq_k", 10, "ACGTTGCAAC", "ACGTTGCAAC")
    for k in (10, 20, 30, 40):
        a = synth.random_dna(rng, 1500)
        r, _ = synth.mutate(rng, a[200:1300])
        add("noisy_k%d" % k, k, r, a)
        r2, _ = synth.mutate(rng, a[200:1300], 0.002, 0.01, 0.005)
        add("clean_k%d" % k, k, r2, a)
        add("revstrand_k%d" % k, k, synth.revcomp(r2), a)
        add("self_k%d" % k, k, a[:600], a[:600])
    # palindromes (read k-mer equal to its own reverse complement -> duplicate tuples)
    pal = "ACGTACGTAC" + "GTACGTACGT"
    add("palindrome_k10", 10, pal * 3, pal * 2)
    add("palindrome_k20", 20, pal * 3, pal * 2)
    at = "AT" * 40
    add("at_repeat_k10", 10, at, at)
    add("homopolymer", 10, "A" * 60, "A" * 45 + "T" * 45)
    # case sensitivity, N, IUPAC
    a = synth.random_dna(rng, 400)
    low = a[:100] + a[100:220].lower() + a[220:]
    add("lower_seq2", 10, a, low)
    add("lower_seq1", 10, low, a)
    add("lower_both", 10, low, low)
    mixed = a[:150] + "".join(c.lower() if i % 3 == 0 else c for i, c in enumerate(a[150:200])) + a[200:]
    add("mixed_case_both", 10, mixed, mixed)
    nn = a[:120] + "N" * 25 + a[145:300] + "n" * 12 + a[312:]
    add("N_runs_both", 10, nn, nn)
    add("N_runs_seq2_only", 10, a, nn)
    add("allN", 10, "N" * 30, "N" * 40)
    iu = a[:50] + "R" + a[51:90] + "y" + a[91:130] + "SWKM" + a[134:170] + "bdhv" + a[174:]
    add("iupac_both", 10, iu, iu)
    add("iupac_vs_N", 10, iu, iu.replace("R", "N").replace("y", "n"))
    add("iupac_k20", 20, iu, iu)
    add("X_in_seq2", 10, a, a[:200] + "X" * 15 + a[215:])
    add("X_in_seq1", 10, a[:200] + "X" * 15 + a[215:], a)
    add("X_in_seq1_too_short", 10, "ACGXT", a)
    add("star_in_seq1", 10, a[:30] + "*" + a[31:], a)
    add("U_in_seq2", 10, a, a[:100] + "U" + a[101:])
    add("lower_x_seq2", 10, a, a[:100] + "x" + a[101:])
    # tandem repeats -> off-diagonal structure
    unit = synth.random_dna(rng, 37)
    rep = synth.random_dna(rng, 150) + unit * 12 + synth.random_dna(rng, 150)
    rr, _ = synth.mutate(rng, rep)
    add("tandem_repeat", 10, rr, rep)
    add("tandem_repeat_self_k20", 20, rep, rep)
    # benchmark-like shapes, digest only
    big = synth.random_dna(rng, 20000)
    br, _ = synth.mutate(rng, big[3000:13000])
    add("10k_x_20k", 10, br[:10000], big, full=False)
    add("10k_x_20k_k20", 20, br[:10000], big, full=False)
    big2 = big[:8000] + big[8000:14000].lower() + big[14000:]
    add("10k_x_20k_softmasked", 10, br[:10000], big2, full=False)
    dump("kmerhits.json.gz", {"source": "dotdata() SF:545-549/951-983", "cases": cases})


# ---------------------------------------------------------------------------
def gen_cleaners(m):
    """C1 / C2 / R4 on explicit hit lists, including tie and threshold edges."""
    rng = np.random.default_rng(202)
    cases = []

    def diag(j0, i0, n, step=1):
        return [(j0 + t * step, i0 + t * step) for t in range(n)]

    def anti(j0, i0, n):
        return [(j0 + t, i0 - t) for t in range(n)]

    def add(name, hits):
        hits = sorted(hits)
        c = {"name": name, "hits": [list(h) for h in hits]}
        c["c1"] = call(m.clean_dotdata_diagnal_and_anti_diagnal, list(hits))
        d = call(m.clean_dotdata_diagnal_m1b, list(hits))
        c["c2_diag"] = {"ok": d["ok"][0]} if "ok" in d else d
        if "ok" in d:
            kept = d["ok"][0]
            left = [h for h in hits if list(h) not in kept]
            a = call(m.clean_dotdata_anti_diagnal_m1b, left)
            c["c2_anti_on_left"] = {"ok": a["ok"][0]} if "ok" in a else a
            uni = kept + (a["ok"][0] if "ok" in a else [])
            if uni:
                c["count10"] = m.eu_dis_dots_within_10perc(uni)
        if "ok" in c["c1"] and len(c["c1"]["ok"]) > 0 and c["c1"]["ok"] != [[], []]:
            kept1 = c["c1"]["ok"]
            c["meanabs"] = m.eu_dis_abs_calcu(kept1)
            r4 = call(m.dis_to_diagnal_most_abundant_defined, [list(h) for h in kept1])
            c["r4"] = r4
            if "ok" in r4:
                c["dir"] = call(m.eu_dis_dir_calcu, [[h[0] + r4["ok"], h[1]] for h in kept1])
        cases.append(c)

    add("one_diag_11", diag(5, 5, 11))
    add("one_diag_10", diag(5, 5, 10))
    add("diag_51", diag(0, 3, 51))
    add("diag_50", diag(0, 3, 50))
    add("two_diags_tie", diag(0, 0, 30) + diag(0, 200, 30))
    add("two_diags_gap9", diag(0, 0, 30) + diag(100, 109, 30))
    add("two_diags_gap10", diag(0, 0, 30) + diag(100, 110, 30))
    add("diag_plus_anti", diag(0, 0, 60) + anti(100, 300, 40))
    add("anti_only_12", anti(10, 200, 12))
    add("anti_only_60", anti(10, 200, 60))
    add("noise_only", [(int(a), int(b)) for a, b in rng.integers(0, 3000, size=(40, 2))])
    add("diag_with_noise", diag(0, 0, 200) + [(int(a), int(b)) for a, b in rng.integers(0, 400, size=(60, 2))])
    add("dup_tuples", diag(0, 0, 8) + diag(0, 0, 8))
    add("shifted_diag_del", diag(0, 0, 150) + diag(150, 450, 150))
    add("r4_kat", [(100 + t, 103 + t) for t in range(50)] + [(100 + t, 600 + t) for t in range(20)])
    add("zero_j", [(0, 0), (0, 5)] + diag(1, 1, 20))
    add("chain_gaps", [(0, 9 * t) for t in range(30)])
    add("chain_gaps10", [(0, 10 * t) for t in range(30)])
    for t in range(12):
        n = int(rng.integers(30, 400))
        base = diag(0, int(rng.integers(0, 50)), n)
        pts = list(base)
        for _ in range(int(rng.integers(0, 4))):
            pts += diag(int(rng.integers(0, 300)), int(rng.integers(0, 900)), int(rng.integers(5, 70)))
        for _ in range(int(rng.integers(0, 3))):
            pts += anti(int(rng.integers(0, 300)), int(rng.integers(300, 900)), int(rng.integers(5, 70)))
        pts += [(int(a), int(b)) for a, b in rng.integers(0, 900, size=(int(rng.integers(0, 80)), 2))]
        add("random_mix_%d" % t, pts)
    dump("cleaners.json.gz", {"source": "SF:404-448,551-591,705-733", "cases": cases})


# ---------------------------------------------------------------------------
def gen_scorers(m):
    rng = np.random.default_rng(303)
    cases = []

    def add(name, ref, alt, read, miss, k):
        x = [read, miss, name]
        c = {"name": name, "ref": ref, "alt": alt, "read": read, "miss": miss, "k": k}
        c["s1"] = call(m.calcu_vapor_single_read_score_abs_dis_m1b, ref, alt, x, k)
        c["s2"] = call(m.calcu_vapor_single_read_score_within_10Perc_m1b, ref, alt, x, k)
        c["s3"] = call(m.calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal, ref, alt, x, k)
        cases.append(c)

    for t in range(40):
        svt = ["DEL", "TANDUP", "INV", "INS"][t % 4]
        span = int(rng.integers(60, 1800))
        f = min(500, span)
        g = synth.random_dna(rng, 2 * f + 3 * span + 400)
        s, e = f + 100, f + 100 + span
        refw = g[s - f:e + f]
        if svt == "DEL":
            altw = refw[:f] + refw[-f:]
        elif svt == "TANDUP":
            altw = refw[:f] + refw[f:-f] * 2 + refw[-f:]
        elif svt == "INV":
            altw = refw[:f] + synth.revcomp(refw[f:-f]) + refw[-f:]
        else:
            ins = synth.random_dna(rng, span)
            refw = g[s - f:s + f + span]
            altw = g[s - f:s] + ins + g[s:s + f]
        hap = altw if (t // 4) % 2 == 0 else refw
        wl = {"DEL": 2 * f, "TANDUP": 2 * span + 2 * f, "INV": span + 2 * f, "INS": span + 2 * f}[svt]
        src = hap + synth.random_dna(rng, 200)
        err = (0.01, 0.08, 0.04) if t % 5 else (0.001, 0.004, 0.002)
        read, _ = synth.mutate(rng, src, *err)
        miss = 0 if t % 7 else int(rng.integers(1, 40))
        read = read[:max(wl - miss, 20)]
        k = 10 if t % 9 else 20
        add("%s_%d" % (svt, t), refw, altw, read, miss, k)
    # gate/edge cases
    g = synth.random_dna(rng, 1200)
    add("unrelated_read", g, g[:500] + g[-500:], synth.random_dna(rng, 900), 0, 10)
    add("perfect_diag_ref", g, g[:500] + g[-500:], g[:1000], 0, 10)
    add("perfect_diag_alt", g, g[:500] + g[-500:], (g[:500] + g[-500:])[:990], 0, 10)
    add("tiny_read", g, g[:500] + g[-500:], g[:25], 0, 10)
    add("read_shorter_than_k", g, g[:500] + g[-500:], "ACGT", 0, 10)
    add("only_ref_span", g, g[:300], g[:1100], 0, 10)
    add("only_alt_span", g[:300], g, g[:1100], 0, 10)
    junk = synth.random_dna(rng, 700)
    add("ref_span_alt_short_cover", g, g[:300] + junk, g[:1100], 0, 10)
    add("alt_span_ref_short_cover", g[:300] + junk, g, g[:1100], 0, 10)
    add("neither_span", g[:300] + junk, g[:280] + junk[::-1], g[:1100], 0, 10)
    low = g[:400] + g[400:800].lower() + g[800:]
    rd, _ = synth.mutate(rng, g[:1100], 0.002, 0.01, 0.005)
    add("softmasked_ref", low, low[:500] + low[-500:], rd, 0, 10)
    add("softmasked_ref_read_alt", low, low[:500] + low[-500:], (g[:500] + g[-500:])[:950], 0, 10)
    add("ref_with_N", g[:600] + "N" * 40 + g[640:], g[:500] + g[-500:], rd, 0, 10)
    add("read_with_N", g, g[:500] + g[-500:], rd[:300] + "N" * 15 + rd[315:], 0, 10)
    add("alt_all_X_ins", g[:1000], g[:500] + "X" * 300 + g[500:1000], rd[:1000], 0, 10)
    add("miss_bp_large", g, g[:500] + g[-500:], rd[200:], 200, 10)
    add("k30", g, g[:500] + g[-500:], rd, 0, 30)
    add("k40", g, g[:500] + g[-500:], rd, 0, 40)
    add("read_X", g, g[:500] + g[-500:], rd[:100] + "X" + rd[101:], 0, 10)
    dump("scorers.json.gz", {"source": "SF:182-203,241-257,277-294", "cases": cases})


# ---------------------------------------------------------------------------
def gen_window(m):
    rng = np.random.default_rng(404)
    cases = []
    trace = []
    orig_qc = m.qual_check_repetitive_region
    orig_xm = m.X_means_cluster_reformat
    xm_calls = [0]

    def qc(hits):
        n = len(hits)
        d = sum(1 for x in hits if x[0] == x[1])
        lo = sum(1 for x in hits if x[0] > x[1])
        trace.append([n, d, lo])
        return orig_qc(hits)

    def xm(other):
        xm_calls[0] += 1
        return orig_xm(other)

    m.qual_check_repetitive_region = qc
    m.X_means_cluster_reformat = xm

    XM_SEEDS = (7, 8, 1234)

    def add(name, seq):
        del trace[:]
        xm_calls[0] = 0
        np.random.seed(XM_SEEDS[0])         # (sklearn's KMeans and scipy's kmeans draw from numpy's global generator, SF:860-881)
        r = call(m.window_size_refine, seq)
        c = {"name": name, "seq": seq, "qc_trace": [list(t) for t in trace], "xmeans_calls": xm_calls[0]}
        if "error" in r:
            c["error"] = r["error"]
        else:
            w, q = r["ok"]
            c["window_size"] = w
            c["qc"] = q
        if xm_calls[0] > 0:
            # the X-means branch is unseeded in the reference, but many of its outcomes do not depend on
            # the draws (BIC keeps one cluster).  The reference's answer under several seeds: equal answers = pinned in fact.
            by_seed = {str(XM_SEEDS[0]): jsonable(r)}
            for sd in XM_SEEDS[1:]:
                np.random.seed(sd)
                by_seed[str(sd)] = jsonable(call(m.window_size_refine, seq))
            c["xmeans_by_seed"] = by_seed
            vals = list(by_seed.values())
            c["xmeans_seed_independent"] = bool(all("ok" in v for v in vals) and all(v == vals[0] for v in vals))
            c["xmeans_reference_raises"] = sorted({v["error"] for v in vals if "error" in v})
        cases.append(c)

    for n in (60, 500, 1500, 4000):
        add("random_%d" % n, synth.random_dna(rng, n))
    add("short", "ACGTAC")
    add("empty", "")
    g = synth.random_dna(rng, 900)
    add("many_N", g[:300] + "N" * 101 + g[300:])
    add("N_100_ok", g[:300] + "N" * 60 + "n" * 40 + g[300:])
    add("with_X", g[:300] + "X" * 200 + g[300:])
    add("softmasked", g[:300] + g[300:600].lower() + g[600:])
    add("all_lower", g.lower())
    for unit_len, copies in ((37, 12), (120, 5), (300, 3), (15, 40), (500, 2)):
        unit = synth.random_dna(rng, unit_len)
        add("tandem_%dx%d" % (unit_len, copies),
            synth.random_dna(rng, 200) + unit * copies + synth.random_dna(rng, 200))
    unit = synth.random_dna(rng, 200)
    add("inverted_repeat", synth.random_dna(rng, 200) + unit + synth.random_dna(rng, 100) + synth.revcomp(unit) + synth.random_dna(rng, 200))
    add("homopolymer", "A" * 300)
    add("dinuc", "AC" * 200)
    # alt windows of tandem duplications as vapor_simple_tandup_Vapor builds them (SF:1755: ref[:f] + mid + mid + ref[-f:]):
    # the windows of the product that meet this branch
    for mid_len, flank in ((150, 150), (300, 300), (600, 500), (1000, 500), (2000, 500), (3500, 500), (5000, 500), (8000, 500)):
        gg = synth.random_dna(rng, mid_len + 2 * flank)
        mid = gg[flank:-flank]
        add("tandup_alt_%d" % mid_len, gg[:flank] + mid + mid + gg[-flank:])
    gg = synth.random_dna(rng, 1600)
    add("tandup_alt_triple_500", gg[:500] + gg[500:1000] * 3 + gg[-500:])
    m.qual_check_repetitive_region = orig_qc
    m.X_means_cluster_reformat = orig_xm
    dump("window.json.gz", {"source": "window_size_refine SF:2030-2046, qual_check SF:1154-1171",
                            "note": "cases with xmeans_calls>0 pass through the reference's unseeded X-means (SURVEY §8a-Q); they were run "
                                    "under numpy seeds %s: where all answers agree (xmeans_seed_independent) the answer is pinned, "
                                    "elsewhere only qc_trace is (xmeans_reference_raises: what the reference raised - scipy.std, "
                                    "SF:878, is gone from current SciPy, so every split into more than one cluster ends there)" % (XM_SEEDS,),
                            "cases": cases})


# ---------------------------------------------------------------------------
def gen_genotype(m):
    rng = np.random.default_rng(505)
    cases = []

    def add(scores):
        r = call(m.result_organize_ins, ["key", list(scores)])
        c = {"scores": list(scores), "organize": r}
        if "ok" in r and "NA" not in r["ok"]:
            c["gt"] = call(m.gt_estimate_log_likelihood, r["ok"])
        cases.append(c)

    add([])
    add([0.5, 0.4, -0.3, -1.0])
    add([0.5] * 10)
    add([0.004, -0.2, -0.3, -0.4, -0.5, -0.6])
    add([0.3, -0.2, -0.3, -0.4, -0.5, -0.6, -0.7, -0.8])
    add([0.0])
    add([-0.0, 1e-9])
    add([0.005, 0.015, 0.025, -0.005, 0.125, 0.675])
    for n in range(1, 21):
        for l in sorted({0, 1, n // 4, n // 2, (3 * n) // 4, n - 1, n}):
            if 0 <= l <= n:
                s = [float(rng.uniform(0.05, 0.9)) for _ in range(n - l)] + \
                    [float(-rng.uniform(0.0, 30.0)) for _ in range(l)]
                rng.shuffle(s)
                add(s)
    for _ in range(40):
        n = int(rng.integers(1, 25))
        add([float(x) for x in rng.normal(0, 1, size=n)])
    dump("genotype.json.gz", {"source": "result_organize_ins SF:1219-1231, gt_estimate_log_likelihood SF:2054-2077",
                              "cases": cases})


# ---------------------------------------------------------------------------
def world_to_json(w):
    return {"contigs": w.contigs,
            "reads": {c: [[r.qname, r.pos, r.cigar, r.seq, r.ref_span] for r in rs] for c, rs in w.reads.items()},
            "loci": [[l.chrom, l.svtype, l.start, l.end, l.svid, l.ins_seq, l.extra] for l in w.loci]}


def gen_io(m):
    rng = np.random.default_rng(606)
    cases = []
    kats = [("100S500M20I300M", 1000, 1200, 2200), ("50M300D500M", 1000, 1200, 2200),
            ("500M", 1000, 1200, 2200), ("10S20M5I30M5D400M", 100, 130, 400),
            ("100M", 1000, 1200, 2200), ("20M1000N50M", 100, 150, 300), ("30M5X40M", 100, 120, 160),
            ("10H20S300M", 5, 100, 200), ("*", 5, 100, 200), ("5I100M", 1, 1, 50)]
    for c in kats:
        cases.append({"fn": "cigar2alignstart_by_pos", "args": list(c), "out": call(m.cigar2alignstart_by_pos, *c)})
    for _ in range(30):
        seg = synth.random_dna(rng, int(rng.integers(50, 600)))
        _r, cg = synth.mutate(rng, seg)
        a0 = int(rng.integers(1, 500))
        st = a0 + int(rng.integers(-20, len(seg) + 30))
        args = [cg, a0, st, st + 100]
        cases.append({"fn": "cigar2alignstart_by_pos", "args": args, "out": call(m.cigar2alignstart_by_pos, *args)})
    # read extraction on a small world
    w = synth.make_world(61, 4, ("DEL", "TANDUP", "INV", "INS"), span_range=(150, 700), read_len=2600, n_reads=26)
    shim = ShimOS(w)
    m.os = shim
    for l in w.loci:
        f = min(500, l.end - l.start) if l.svtype != "INS" else min(500, len(l.ins_seq))
        for fn, info in (("simple_del_chop_pacbio_read_simple_short", [l.chrom, l.start, l.end]),
                         ("simple_chop_pacbio_read_simple_short", [l.chrom, l.start, l.end])):
            cases.append({"fn": fn, "args": ["x.bam", info, f], "out": call(getattr(m, fn), "x.bam", info, f)})
        cases.append({"fn": "ref_seq_readin", "args": ["ref.fa", l.chrom, l.start - f, l.end + f],
                      "out": call(m.ref_seq_readin, "ref.fa", l.chrom, l.start - f, l.end + f)})
        cases.append({"fn": "ref_seq_readin", "args": ["ref.fa", l.chrom, l.start, l.start + 75, "TRUE"],
                      "out": call(m.ref_seq_readin, "ref.fa", l.chrom, l.start, l.start + 75, "TRUE")})
    m.os = os
    many = [[synth.random_dna(rng, 10), int(rng.integers(0, 6)), "q%d" % i] for i in range(45)]
    cases.append({"fn": "minimize_pacbio_read_list", "args": [many], "out": call(m.minimize_pacbio_read_list, many)})
    cases.append({"fn": "minimize_pacbio_read_list", "args": [many[:7]], "out": call(m.minimize_pacbio_read_list, many[:7])})
    for b in ([["c", 100, 150]], [["c", 100, 700]], [["c", 100, 600]], [["c", 100, 190]]):
        cases.append({"fn": "flank_length_calculate", "args": b, "out": call(m.flank_length_calculate, *b)})
    dump("io.json.gz", {"source": "SF:309-354,1091-1102,1203-1217,1378-1401,794-802",
                        "world": world_to_json(w), "cases": cases})


# ---------------------------------------------------------------------------
def load_cli(m):
    """Pull the parsing helpers out of the reference CLI script without running its
    argparse tail: exec only the `def` blocks above the `if len(sys.argv)<2:` line."""
    src = open(REF_CLI).read()
    head = src.split("\nif len(sys.argv)<2:")[0]
    ns = {"__name__": "vapor_ref_cli"}
    ns.update({k: getattr(m, k) for k in dir(m) if not k.startswith("__")})
    exec(compile(head, REF_CLI, "exec"), ns)
    return ns


def run_bed(m, cli, world, tmp, num_reads_cff=3):
    """What `vapor bed` does (vapor_vali/vapor:322-367) with figures disabled."""
    bed = os.path.join(tmp, "in.bed")
    open(bed, "w").write(synth.bed_text(world))
    out_name = os.path.join(tmp, "out.vapor")
    out_path = os.path.join(tmp, "figs") + "/"
    os.makedirs(out_path, exist_ok=True)
    bed_info = cli["bed_info_readin"](bed, out_path)
    m.write_output_initiate(out_name)
    per_locus = []
    for x in bed_info:
        if x[-1] in ["a/", "/a", "/", "DEL"]:
            key = ":".join([str(i) for i in x[:-3]] + ["DEL"])
            sc = call(m.vapor_simple_del_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", x[:-3], out_path + "f.png")
        elif x[-1] in ["a/a^", "a^/a", "a^/a^", "INV"]:
            key = ":".join([str(i) for i in x[:-3]] + ["INV"])
            sc = call(m.vapor_simple_inv_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", x[:-3], out_path + "f.png")
        elif x[-1] in ["INS"]:
            key = ":".join([str(i) for i in x[:-3] + ["INS"]])
            ins_pos = "_".join([str(i) for i in x[:2]])
            ins_seq = "X" * x[4] if isinstance(x[4], int) else x[4]
            sc = call(m.vapor_simple_ins_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", ins_pos, ins_seq, out_path + "f.png", "+")
        elif x[-1] in ["a/aa", "aa/a", "aa/aa", "DUP", "TANDUP"]:
            key = ":".join([str(i) for i in x[:-3]] + ["TANDUP"])
            sc = call(m.vapor_simple_tandup_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", x[:-3], out_path + "f.png")
        else:
            continue
        rec = {"bed_row": x, "key": key, "scores": sc}
        if getattr(m, "_xm_counter", None) is not None:
            rec["xmeans_calls"] = m._xm_counter[0]          # X-means calls of this locus (window_size_refine's repeat check)
            m._xm_counter[0] = 0
        if "ok" in sc:
            res = m.result_organize_ins([key, sc["ok"]])
            m.write_output_main(out_name, res[0].split(":") + [x[3]] + res[1:])
            rec["organize"] = res
        per_locus.append(rec)
    return per_locus, open(out_name).read()


def gen_locus(m):
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None   # rendering is out of scope
    out_cases = []
    tmp = tempfile.mkdtemp(prefix="vapor_golden_")
    specs = [
        ("bed_small_mix", dict(seed=71, n_loci=8, svtypes=("DEL", "TANDUP", "INV", "INS"), span_range=(120, 1500), read_len=4600, n_reads=9)),
        ("bed_del_short_span", dict(seed=72, n_loci=4, svtypes=("DEL",), span_range=(50, 400), read_len=1800, n_reads=8)),
        ("bed_few_reads", dict(seed=73, n_loci=3, svtypes=("DEL", "INV", "TANDUP"), span_range=(200, 600), read_len=2400, n_reads=3)),
        ("bed_hom_alt", dict(seed=74, n_loci=4, svtypes=("DEL", "TANDUP", "INV", "INS"), span_range=(300, 900), read_len=3200, n_reads=8, alt_fraction=1.0)),
        ("bed_hom_ref", dict(seed=75, n_loci=4, svtypes=("DEL", "TANDUP", "INV", "INS"), span_range=(300, 900), read_len=3200, n_reads=8, alt_fraction=0.0)),
        ("bed_many_reads", dict(seed=76, n_loci=2, svtypes=("DEL", "INV"), span_range=(300, 700), read_len=2000, n_reads=30)),
    ]
    # a tandem-duplication-only world whose alt windows all meet the X-means branch of the repeat check
    specs.append(("bed_tandup_band", dict(seed=78, n_loci=6, svtypes=("TANDUP",), span_range=(150, 3000), read_len=7200, n_reads=8)))
    calls, xm_orig = _count_xmeans(m)
    m._xm_counter = calls
    for name, kw in specs:
        w = synth.make_world(**kw)
        m.os = ShimOS(w)
        calls[0] = 0
        np.random.seed(7)                 # (the reference's X-means draws from numpy's global generator, SF:860-881)
        per_locus, text = run_bed(m, cli, w, tmp)
        case = {"name": name, "world": world_to_json(w), "bed": synth.bed_text(w), "per_locus": per_locus, "vapor_text": text}
        if any(p.get("xmeans_calls") for p in per_locus):
            # the same run under other seeds: equal tables = the rows do not depend on the draws (pinned end to end)
            same = True
            for sd in (8, 1234):
                np.random.seed(sd)
                calls[0] = 0
                pl2, text2 = run_bed(m, cli, w, tmp)
                same = same and text2 == text and jsonable([p["scores"] for p in pl2]) == jsonable([p["scores"] for p in per_locus])
            case["xmeans_seed_independent"] = bool(same)
        m.os = os
        out_cases.append(case)
        print("  %s: %s" % (name, [len(p["scores"].get("ok", [])) if "ok" in p["scores"] else p["scores"] for p in per_locus]))
    # long-span fallbacks (>= 10 kb): junction windows only, 1 kb reads
    w = synth.make_world(seed=77, n_loci=3, svtypes=("DEL", "INV", "TANDUP"), span_range=(10050, 10400), read_len=1500, n_reads=8)
    m.os = ShimOS(w)
    calls[0] = 0
    np.random.seed(7)
    per_locus, text = run_bed(m, cli, w, tmp)
    m.os = os
    m._xm_counter = None
    m.X_means_cluster_reformat = xm_orig
    out_cases.append({"name": "bed_long_span", "world": world_to_json(w), "bed": synth.bed_text(w),
                      "per_locus": per_locus, "vapor_text": text})
    print("  bed_long_span: %s" % [p["scores"] for p in per_locus])
    dump("locus_bed.json.gz", {"source": "vapor bed loop vapor_vali/vapor:322-367 + drivers SF:1701-1933 (figures off)",
                               "cases": out_cases})


def gen_long(m):
    """Spans of 20 kb and more (the junction-window branches, SF:1728-1745, 1769-1785, 1918-1933, at the spans the
    truth sets' tail holds - up to 99 kb): a fixture of its own, so that the files of the earlier rounds stay byte for byte."""
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None
    tmp = tempfile.mkdtemp(prefix="vapor_golden_")
    calls, xm_orig = _count_xmeans(m)
    m._xm_counter = calls
    out_cases = []
    for name, kw in (
            ("bed_spans_20kb_up", dict(seed=91, n_loci=6, svtypes=("DEL", "INV", "TANDUP", "DEL", "INV", "DEL"),
                                       spans=(25013, 42077, 20480, 99001, 21950, 60003), read_len=1500, n_reads=8)),
            ("bed_spans_20kb_up_hom", dict(seed=92, n_loci=3, svtypes=("DEL", "INV", "TANDUP"), spans=(33333, 20001, 27500), read_len=1800,
                                           n_reads=7, alt_fraction=1.0))):
        w = synth.make_world(**kw)
        m.os = ShimOS(w)
        calls[0] = 0
        np.random.seed(7)
        per_locus, text = run_bed(m, cli, w, tmp)
        m.os = os
        assert not any(p.get("xmeans_calls") for p in per_locus), "a junction window met the X-means branch: not pinned"
        out_cases.append({"name": name, "world": world_to_json(w), "bed": synth.bed_text(w), "per_locus": per_locus, "vapor_text": text})
        print("  %s: %s" % (name, [len(p["scores"].get("ok", [])) if "ok" in p["scores"] else p["scores"] for p in per_locus]))
    m._xm_counter = None
    m.X_means_cluster_reformat = xm_orig
    dump("locus_long.json.gz", {"source": "vapor bed loop vapor_vali/vapor:322-367 + the drivers' junction-window branches for spans >= "
                                          "default_max_sv_test (SF:1728-1745, 1769-1785, 1918-1933), figures off", "cases": out_cases})


def run_vcf(m, cli, world, tmp, header, num_reads_cff=3):
    """What `vapor vcf` does (vapor_vali/vapor:374-466) with figures disabled."""
    vcf = os.path.join(tmp, "in_%d.vcf" % int(header))
    open(vcf, "w").write(synth.vcf_text(world, header=header))
    out_path = os.path.join(tmp, "figs") + "/"
    vcf_list, rec_hash = cli["vcf_list_readin"](vcf)
    rec_new = m.vcf_rec_hash_modify(rec_hash)
    m.write_output_initiate(vcf + ".vapor")
    rows = []
    for x in list(vcf_list.keys()):
        for y in vcf_list[x]:
            y0 = jsonable(y)
            if x in ("DEL", "INV"):
                if y[2] - y[1] < 50:
                    key, sc = ":".join([str(i) for i in y] + ["DEL"]), {"ok": []}
                else:
                    key = ":".join([str(i) for i in y] + [x])
                    fn = m.vapor_simple_del_Vapor if x == "DEL" else m.vapor_simple_inv_Vapor
                    sc = call(fn, num_reads_cff, 1, "x.bam", "ref.fa", y, out_path + "f.png")
            elif x == "INS":
                key = ":".join([str(i) for i in y[:3] + ["INS"]])
                ins_pos = "_".join([str(i) for i in y[:2]])
                ins_seq = y[-1] if len(y) == 4 else "X" * y[2]
                sc = call(m.vapor_simple_ins_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", ins_pos, ins_seq, out_path + "f.png", "+")
            elif x == "DISDUP":
                key = ":".join([str(i) for i in y + ["DISDUP"]])
                sc = call(m.vapor_simple_disdup_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", y, out_path + "f.png")
            elif x == "DEL_INV":
                key = ":".join(["_".join([str(i) for i in j]) for j in y] + ["DEL_INV"])
                sc = call(m.vapor_del_inv_Vapor, num_reads_cff, 1, "x.bam", "ref.fa", y, out_path + "f.png")
            elif x == "DUP_INV":
                key = ":".join([str(i) for i in y + ["DUP_INV"]])
                sc = call(m.vapor_dup_inv_VapoR, num_reads_cff, 1, "x.bam", "ref.fa", y, out_path + "f.png")
            else:
                continue
            rec = {"type": x, "item": y0, "key": key, "scores": sc}
            if "ok" in sc:
                m.write_output_main(vcf + ".vapor", m.result_organize_ins([key, sc["ok"]]))
            rows.append(rec)
    table = open(vcf + ".vapor").read()
    final = call(m.vcf_vapor_modify, vcf, rec_new)
    return {"vcf": synth.vcf_text(world, header=header), "per_record": rows, "table": table,
            "final": open(vcf + ".vapor").read() if "ok" in final else None, "final_status": final if "error" in final else "ok"}


def gen_vcf(m):
    # the complex types (DISDUP, DUP_INV, DEL_INV, Other=) have worlds of their own, built to stay out of the
    # reference's unseeded X-means branch: gen_complex
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None
    tmp = tempfile.mkdtemp(prefix="vapor_golden_vcf_")
    os.makedirs(os.path.join(tmp, "figs"), exist_ok=True)
    cases = []
    specs = [
        ("vcf_simple", dict(seed=81, n_loci=8, svtypes=("DEL", "INV", "INS", "TANDUP"), span_range=(120, 1200), read_len=4200, n_reads=8)),
        ("vcf_tiny_span", dict(seed=83, n_loci=4, svtypes=("DEL", "INV"), span_range=(20, 60), read_len=1600, n_reads=6)),
    ]
    for name, kw in specs:
        w = synth.make_world(**kw)
        for header in (False, True):
            m.os = ShimOS(w)
            r = run_vcf(m, cli, w, tmp, header)
            m.os = os
            r.update({"name": name + ("_hdr" if header else "_nohdr"), "world": world_to_json(w) if not header else None,
                      "world_of": name + "_nohdr", "header": header})
            cases.append(r)
            print("  %s: %s final=%s" % (r["name"], [(p["type"], len(p["scores"].get("ok", [])) if "ok" in p["scores"] else p["scores"]["error"]) for p in r["per_record"]], r["final_status"]))
    dump("locus_vcf.json.gz", {"source": "vapor vcf loop vapor_vali/vapor:374-466 + vcf_vapor_modify SF:1972-2028 (figures off)",
                               "cases": cases})


def gen_other(m):
    """vapor_CANNOT_CLASSIFY_VapoR (SF:1490-1555) on letter-structure calls, incl. the junction fallback."""
    rng = np.random.default_rng(909)
    m.make_event_figure_1 = lambda *a, **k: None
    tmp = tempfile.mkdtemp(prefix="vapor_golden_other_")
    cases = []
    specs = [("del_b", "ab_ab", "a_ab", 700, 500, 8, 0.6), ("inv_b", "ab_ab", "ab^_ab", 600, 900, 8, 0.6),
             ("dup_b", "ab_ab", "abb_ab", 500, 400, 8, 0.6), ("swap", "ab_ab", "ba_ab", 450, 650, 8, 1.0),
             ("both_alt", "ab_ab", "a_b^a", 500, 500, 9, 1.0), ("few_reads", "ab_ab", "a_ab", 500, 500, 3, 0.6),
             ("long_span", "ab_ab", "a_ab", 6000, 5000, 8, 0.6), ("same_as_ref", "ab_ab", "ab_ab", 500, 500, 6, 0.5)]
    w = synth.SynthWorld()
    for t, (name, ref_s, alt_s, la, lb, n_reads, alt_frac) in enumerate(specs):
        chrom = "o%d" % (t + 1)
        f = 500
        left = f + 400
        contig = synth.random_dna(rng, left + la + lb + 9000)
        b1, b2, b3 = left, left + la, left + la + lb
        w.contigs[chrom] = contig
        blocks = {"a": contig[b1:b2], "b": contig[b2:b3]}
        alts = [x for x in alt_s.split("_") if x not in ref_s.split("_")] or ["ab"]
        recs = []
        for ri in range(n_reads):
            from_alt = rng.random() < alt_frac
            if from_alt:
                al = alts[ri % len(alts)]
                mid = ""
                for ch in al:
                    if ch == "^":
                        continue
                mid = "".join(synth.revcomp(blocks[x[0]]) if "^" in x else blocks[x] for x in _letters(al))
                hap = contig[:b1] + mid + contig[b3:]
            else:
                hap = contig
            a0 = b1 - f - 1 - int(rng.integers(1, 250))
            seg = hap[a0:a0 + la + lb + 2 * f + 2600]
            rd, cg = synth.mutate(rng, seg)
            recs.append(synth.SamRecord("%s_%d" % (name, ri), chrom, a0 + 1, cg, rd, len(seg)))
        w.reads[chrom] = recs
        cases.append({"name": name, "sv_info": [ref_s, alt_s, chrom, str(b1), str(b2), str(b3)]})
    ref_path = os.path.join(tmp, "ref.fa")
    with open(ref_path + ".fai", "w") as fo:
        for k, v in w.contigs.items():
            fo.write("%s\t%d\t0\t60\t61\n" % (k, len(v)))
    m.os = ShimOS(w)
    for c in cases:
        c["scores"] = call(m.vapor_CANNOT_CLASSIFY_VapoR, 3, 1, "x.bam", ref_path, list(c["sv_info"]), os.path.join(tmp, "f.png"))
        print("  %s: %s" % (c["name"], c["scores"] if "error" in c["scores"] else [round(v, 3) for v in c["scores"]["ok"]]))
    m.os = os
    dump("locus_other.json.gz", {"source": "vapor_CANNOT_CLASSIFY_VapoR SF:1490-1555 (figures off)",
                                 "world": world_to_json(w), "cases": cases})


def _letters(al):
    out = []
    for ch in al:
        if ch == "^":
            out[-1] += ch
        else:
            out.append(ch)
    return out


def gen_config1(m):
    """BASELINE.json configs[0]: the loci of the reference's own vapor_test/vapor_test.bed (coordinates and
    types are fixture data) on a stand-in genome, through the reference's `vapor bed` loop."""
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None
    rows = [l.split() for l in open("/root/reference/vapor_test/vapor_test.bed") if l.strip()]
    w = synth.make_world_from_bed(rows, seed=10)
    tmp = tempfile.mkdtemp(prefix="vapor_golden_cfg1_")
    m.os = ShimOS(w)
    per_locus, text = run_bed(m, cli, w, tmp)
    m.os = os
    print("  config1: %s" % [len(p["scores"].get("ok", [])) if "ok" in p["scores"] else p["scores"] for p in per_locus])
    dump("config1_bed.json.gz", {"source": "vapor_test/vapor_test.bed rows (chr start end TYPE) + vapor bed loop vapor_vali/vapor:322-367",
                                 "bed_rows": rows, "seed": 10, "bed": synth.bed_text(w),
                                 "cases": [{"name": "vapor_test_bed", "per_locus": per_locus, "vapor_text": text}]})


def gen_melt(m):
    """`vapor ins` (MELT calls): melt_info_readin, vapor_vali/vapor:52-81, driven directly because the
    script's own `ins` branch reads an argparse attribute that does not exist (vapor_vali/vapor:310)."""
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None
    tmp = tempfile.mkdtemp(prefix="vapor_golden_melt_")
    w = synth.make_world(seed=91, n_loci=5, svtypes=("INS",), span_range=(100, 200), read_len=2600, n_reads=8,
                         ins_len_range=(150, 700))
    prefix = os.path.join(tmp, "S1.melt.sites")
    lines = ["##fileformat=VCFv4.2"]
    fa = {}
    for t, l in enumerate(w.loci):
        key = "%s_%d" % (l.chrom, l.start)
        pol = "+" if t % 2 == 0 else "-"
        seq = l.ins_seq if pol == "+" else synth.revcomp(l.ins_seq)
        if t == 3:
            seq = ""                      # no assembled sequence: the reader falls back to X * SVLEN
        if t == 1:
            seq = seq[:40] + "N" * 5 + seq[45:]
        fa[key] = seq
        lines.append("\t".join([l.chrom, str(l.start), ".", "<INS:ME:ALU>", "N", ".", "PASS",
                                 "SVTYPE=ALU;SVLEN=%d;MEIINFO=AluYa5,3,281,%s" % (len(l.ins_seq), pol)]))
    open(prefix + ".vcf", "w").write("\n".join(lines) + "\n")
    w.contigs.update(fa)                  # the in-memory samtools serves `faidx <prefix>.fa key` from here
    shim = ShimOS(w)
    m.os = shim
    cli["os"] = shim
    cli["num_reads_cff"] = 3
    r = call(cli["melt_info_readin"], prefix, os.path.join(tmp, "figs") + "/", "S1", "x.bam", "ref.fa")
    m.os = os
    text = open(prefix + ".vapor").read() if os.path.exists(prefix + ".vapor") else None
    print("  melt:", r if "error" in r else "ok", (text or "").count("\n"), "lines")
    for k in fa:
        w.contigs.pop(k)
    dump("melt_ins.json.gz", {"source": "melt_info_readin vapor_vali/vapor:52-81 (figures off)", "world": world_to_json(w),
                              "fasta": fa, "vcf": "\n".join(lines) + "\n", "status": r if "error" in r else "ok",
                              "cases": [{"name": "melt", "vapor_text": text}]})

# ---------------------------------------------------------------------------
COMPLEX_SPECS = [
    # DUP_INV on one contig, copy right of the block (a b a^), left of it (b^ a b), inside it (a a^)
    dict(type="DUP_INV", a=200, gap=1400), dict(type="DUP_INV", a=260, gap=1900), dict(type="DUP_INV", a=180, gap=-1300),
    dict(type="DUP_INV", a=520, inside=120, alt_fraction=0.0),
    # far copy (>= 10 kb): the junction-only branch (within_10Perc_m1b); copy on another contig: abs_dis_m1b
    dict(type="DUP_INV", a=320, gap=10400), dict(type="DUP_INV", a=280, xchrom=1000), dict(type="DUP_INV", a=240, gap=1500, n_reads=3),
    # DISDUP from a VCF reaches a scorer only where the whole-region branch is not taken (SF:1801 compares str with
    # int there and raises): far copy, copy on another contig (near and far coordinates), too few spanning reads
    dict(type="DISDUP", a=300, gap=11000), dict(type="DISDUP", a=260, xchrom=900), dict(type="DISDUP", a=260, xchrom=15000),
    dict(type="DISDUP", a=220, gap=1500),
    # DEL_INV: adjacent blocks in both orders, >= 10 kb (vapor_long_del_inv), blocks >= 100 bp apart (TypeError)
    dict(type="DEL_INV", a=300, b=420), dict(type="DEL_INV", a=350, b=300, order="inv,del"),
    dict(type="DEL_INV", a=700, b=9600), dict(type="DEL_INV", a=300, b=400, apart=150), dict(type="DEL_INV", a=240, b=260, n_reads=3),
    # Other=: letter structures through vapor_CANNOT_CLASSIFY_VapoR
    dict(type="OTHER", a=420, b=520, other=("ab/ab", "b/b^")), dict(type="OTHER", a=380, b=460, other=("ab/ab", "a/ab")),
    dict(type="OTHER", a=300, b=2400, other=("ab/ab", "aba/ab")), dict(type="OTHER", a=400, b=500, other=("ab/ab", "ba^/ab")),
]


def run_vcf_complex(m, cli, world, tmp, ref_path, num_reads_cff=3):
    """The complex-type branches of `vapor vcf` (vapor_vali/vapor:425-463) with figures disabled."""
    vcf = os.path.join(tmp, "in.vcf")
    text = synth.complex_vcf_text(world)
    open(vcf, "w").write(text)
    out_path = os.path.join(tmp, "figs") + "/"
    vcf_list, rec_hash = cli["vcf_list_readin"](vcf)
    rec_new = m.vcf_rec_hash_modify(rec_hash)
    m.write_output_initiate(vcf + ".vapor")
    rows = []
    for x in list(vcf_list.keys()):
        for y in vcf_list[x]:
            y0 = jsonable(y)
            if x == "DISDUP":
                key = ":".join([str(i) for i in y + ["DISDUP"]])
                sc = call(m.vapor_simple_disdup_Vapor, num_reads_cff, 1, "x.bam", ref_path, y, out_path + "f.png")
            elif x == "DEL_INV":
                key = ":".join(["_".join([str(i) for i in j]) for j in y] + ["DEL_INV"])
                sc = call(m.vapor_del_inv_Vapor, num_reads_cff, 1, "x.bam", ref_path, y, out_path + "f.png")
            elif x == "DUP_INV":
                key = ":".join([str(i) for i in y + ["DUP_INV"]])
                sc = call(m.vapor_dup_inv_VapoR, num_reads_cff, 1, "x.bam", ref_path, y, out_path + "f.png")
            elif x == "Other":
                key = ":".join([str(i) for i in y + ["CANNOT_CLASSIFY"]])
                sc = call(m.vapor_CANNOT_CLASSIFY_VapoR, num_reads_cff, 1, "x.bam", ref_path, y, out_path + "f.png")
            else:
                raise AssertionError(x)
            rec = {"type": x, "item": y0, "key": key, "scores": sc}
            if "ok" in sc:
                m.write_output_main(vcf + ".vapor", m.result_organize_ins([key, sc["ok"]]))
            rows.append(rec)
    table = open(vcf + ".vapor").read()
    final = call(m.vcf_vapor_modify, vcf, rec_new)
    return {"vcf": text, "per_record": rows, "table": table,
            "final": open(vcf + ".vapor").read() if "ok" in final else None, "final_status": final if "error" in final else "ok"}


def _count_xmeans(m):
    """Counts entries into the reference's unseeded clustering (SF:1165-1167); the complex worlds must never get there."""
    calls = [0]
    orig = m.X_means_cluster_reformat

    def counted(*a, **k):
        calls[0] += 1
        return orig(*a, **k)
    m.X_means_cluster_reformat = counted
    return calls, orig


def gen_complex(m):
    """BASELINE.json configs[3]'s complex types (DISDUP, DUP_INV, DEL_INV, Other=) through the reference's VCF
    loop, on worlds whose windows stay out of the X-means band, so that the vectors are reproducible."""
    cli = load_cli(m)
    m.make_event_figure_1 = lambda *a, **k: None
    tmp = tempfile.mkdtemp(prefix="vapor_golden_cx_")
    os.makedirs(os.path.join(tmp, "figs"), exist_ok=True)
    calls, orig = _count_xmeans(m)
    cases = []
    for name, seed, specs in (("vcf_cx_a", 181, COMPLEX_SPECS), ("vcf_cx_b", 182, COMPLEX_SPECS[:3] + COMPLEX_SPECS[11:13] + COMPLEX_SPECS[16:18])):
        w = synth.make_complex_world(seed, specs)
        ref_path = os.path.join(tmp, name + ".fa")
        with open(ref_path + ".fai", "w") as fo:
            for k, v in w.contigs.items():
                fo.write("%s\t%d\t0\t60\t61\n" % (k, len(v)))
        m.os = ShimOS(w)
        r = run_vcf_complex(m, cli, w, tmp, ref_path)
        m.os = os
        r.update({"name": name, "world": world_to_json(w)})
        cases.append(r)
        print("  %s: %s final=%s" % (name, [(p["type"], len(p["scores"]["ok"]) if "ok" in p["scores"] else p["scores"]["error"])
                                            for p in r["per_record"]], r["final_status"]))
    # driver-level: DISDUP with integer coordinates (the whole-region branch, directed_dis scorer), which the VCF
    # parser never produces (it hands the insert point over as a string)
    drv = []
    w = synth.make_complex_world(183, [dict(type="DISDUP", a=220, gap=1500), dict(type="DISDUP", a=260, gap=-1700),
                                       dict(type="DISDUP", a=300, inside=150, alt_fraction=0.0),
                                       dict(type="DISDUP", a=200, gap=1400, n_reads=3)])
    m.os = ShimOS(w)
    for l in w.loci:
        info = [l.chrom, l.start, l.end, l.chrom, int(l.extra["insert_point"])]
        sc = call(m.vapor_simple_disdup_Vapor, 3, 1, "x.bam", "ref.fa", list(info), os.path.join(tmp, "f.png"))
        drv.append({"sv_info": info, "scores": sc})
        print("  disdup driver %s: %s" % (info, sc if "error" in sc else [round(v, 3) for v in sc["ok"]]))
    m.os = os
    m.X_means_cluster_reformat = orig
    assert calls[0] == 0, "a complex world entered the unseeded X-means branch (%d calls): change its block sizes" % calls[0]
    dump("locus_complex.json.gz", {"source": "vapor vcf loop vapor_vali/vapor:425-463 + drivers SF:1490-1699, 1786-1854 (figures off); "
                                             "xmeans_calls == 0 asserted when generated",
                                   "cases": cases, "disdup_driver": {"world": world_to_json(w), "cases": drv}})


def gen_deep(m):
    """BASELINE.json configs[4]'s concordance leg at a size the reference finishes: loci of 60 reads, every read
    scored by the reference's scorers directly (the drivers keep 20 reads, SF:1091-1102), then the drivers' per-read
    rule, result_organize_ins and gt_estimate_log_likelihood."""
    rng = np.random.default_rng(505)
    cases = []
    for li, (t, n_reads, alt_frac) in enumerate((("DEL", 60, 0.5), ("TANDUP", 60, 0.5), ("INV", 60, 1.0), ("INS", 60, 0.0),
                                                 ("DEL", 64, 0.15), ("INV", 33, 0.5))):
        L = 2400
        ref = synth.random_dna(rng, L)
        span = int(rng.integers(250, 600))
        s = int(rng.integers(700, 900))
        if t == "DEL":
            alt = ref[:s] + ref[s + span:]
        elif t == "TANDUP":
            span = 160
            alt = ref[:s + span] + ref[s:s + span] + ref[s + span:]
        elif t == "INV":
            alt = ref[:s] + synth.revcomp(ref[s:s + span]) + ref[s + span:]
        else:
            alt = ref[:s] + synth.random_dna(rng, span) + ref[s:]
        reads = []
        for ri in range(n_reads):
            hap = alt if rng.random() < alt_frac else ref
            rd, _c = synth.mutate(rng, hap[:min(len(ref), len(alt))])
            reads.append([rd[:min(len(ref), len(alt)) - 5], 0, "d%d_%d" % (li, ri)])
        k = m.window_size_refine(ref)[0]
        scores = []
        per_read = []
        for x in reads:
            if t == "DEL":
                a = m.calcu_vapor_single_read_score_abs_dis_m1b(ref, alt, x, k)
                b = m.calcu_vapor_single_read_score_within_10Perc_m1b(ref, alt, x, k)
                per_read.append([a, b])
                if not 0 in a and not 0 in b:
                    scores.append(min([1 - float(a[1]) / float(a[0]), 1 - float(b[1]) / float(b[0])]))
                elif not 0 in a:
                    scores.append(1 - float(a[1]) / float(a[0]))
                elif not 0 in b:
                    scores.append(1 - float(b[1]) / float(b[0]))
            else:
                fn = m.calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal if t == "TANDUP" else m.calcu_vapor_single_read_score_abs_dis_m1b
                a = fn(ref, alt, x, k)
                per_read.append([a])
                if not 0 in a:
                    scores.append(1 - float(a[1]) / float(a[0]))
        org = m.result_organize_ins(["k%d" % li, scores])
        gt = m.gt_estimate_log_likelihood(org) if org[1] != "NA" else None
        cases.append({"name": "%s_%d" % (t, n_reads), "svtype": t, "ref": ref, "alt": alt, "k": k, "reads": reads,
                      "per_read": per_read, "scores": scores, "organize": org, "gt": gt})
        print("  deep %s: %d reads, %d scored, %s" % (t, n_reads, len(scores), gt))
    dump("deep_loci.json.gz", {"source": "scorers SF:182-294 on every read, drivers' per-read rule SF:1718-1726, 1909-1915, "
                                         "result_organize_ins SF:1219-1231, gt_estimate_log_likelihood SF:2054-2077",
                               "cases": cases})

class PltRecorder:
    """Stands in for matplotlib.pyplot inside the reference module: records what make_event_figure_1 /
    makeDotplot_subfigure (SF:1041-1089) would draw instead of drawing it."""

    def __init__(self):
        self.figs = []
        self.cur = None

    def figure(self, n=None):
        self.cur = {"figure": n, "subplots": [], "saved": None}
        self.figs.append(self.cur)
        return self.cur

    def subplot(self, pos):
        self.cur["subplots"].append({"pos": int(pos)})

    def plot(self, x, y, marker, color=None):
        sp = self.cur["subplots"][-1]
        xa, ya = np.asarray(x, dtype=np.int32), np.asarray(y, dtype=np.int32)
        sp.update({"n": int(len(xa)), "marker": marker, "color": color,
                   "xy_sha256": hashlib.sha256(np.stack([xa, ya], 1).tobytes()).hexdigest(),
                   "first": [int(xa[0]), int(ya[0])], "last": [int(xa[-1]), int(ya[-1])]})

    def xticks(self, ticks, labels):
        self.cur["subplots"][-1].update({"xticks": [float(t) for t in ticks], "xticklabels": list(labels)})

    def title(self, t):
        self.cur["subplots"][-1]["title"] = t

    def grid(self, flag):
        self.cur["subplots"][-1]["grid"] = bool(flag)

    def savefig(self, name):
        self.cur["saved"] = name

    def close(self, fig):
        pass


def gen_figures(m):
    """make_event_figure_1 (SF:1072-1089) as the drivers call it (vapor_vali/vapor:338 naming): the arguments of
    every call and what it plots - the four point sets, tick lists, titles and the (clamped) file name."""
    cli = load_cli(m)
    tmp = tempfile.mkdtemp(prefix="vapor_golden_fig_")
    # (no TANDUP: its alt window always has a 1/6 - 1/4 lower-triangle share and sends the reference into its unseeded
    # X-means, SF:1165, so its figure calls would not be reproducible)
    w = synth.make_world(seed=171, n_loci=8, svtypes=("DEL", "INV", "INS", "DEL"), span_range=(150, 900), read_len=3000, n_reads=7)
    xm_calls, xm_orig = _count_xmeans(m)
    rec = PltRecorder()
    calls = []
    orig = m.make_event_figure_1

    def wrapped(plt_li, scores, best, k, ref_seq, alt_seq, name):
        n0 = len(rec.figs)
        orig(plt_li, scores, best, k, ref_seq, alt_seq, name)
        calls.append({"plt_li": plt_li, "scores": [float(s) for s in scores], "best_read": jsonable(best), "k": k,
                      "ref_seq": ref_seq, "alt_seq": alt_seq, "name": name,
                      "drawn": rec.figs[n0] if len(rec.figs) > n0 else None})
    m.plt = rec
    m.make_event_figure_1 = wrapped
    m.os = ShimOS(w)
    per_locus, _text = run_bed(m, cli, w, tmp)
    m.os = os
    # direct calls: a file name longer than 150 characters (SF:1080-1081), no best read, an empty plot
    c0 = [c for c in calls if c["drawn"] is not None][0]
    long_name = tmp + "/" + "s." + "DEL." + "x" * 170 + ".png"
    for best, name in ((c0["best_read"], long_name), ("", tmp + "/none.png"), ([], tmp + "/none2.png"),
                       (["ACGT", 0, "tiny"], tmp + "/empty.png")):
        wrapped(99, c0["scores"], best, c0["k"], c0["ref_seq"], c0["alt_seq"], name)
    m.make_event_figure_1 = orig
    m.X_means_cluster_reformat = xm_orig
    assert xm_calls[0] == 0
    import matplotlib.pyplot as real_plt
    m.plt = real_plt
    for c in calls:
        c["name"] = c["name"].replace(tmp, "<tmp>")
        if c["drawn"] and c["drawn"]["saved"]:
            c["drawn"]["saved"] = c["drawn"]["saved"].replace(tmp, "<tmp>")
    print("  figures: %d calls, %d drawn" % (len(calls), sum(1 for c in calls if c["drawn"])))
    dump("figures.json.gz", {"source": "make_event_figure_1 / makeDotplot_subfigure SF:1041-1089 with matplotlib.pyplot replaced by a recorder",
                             "world": world_to_json(w), "bed": synth.bed_text(w), "cases": calls})


# ---------------------------------------------------------------------------
def parser_digest(obj) -> str:
    """sha256 of the canonical JSON of a parser's result (dict keys in insertion order: first-seen order is part of the result)."""
    return hashlib.sha256(json.dumps(jsonable(obj), separators=(",", ":")).encode()).hexdigest()


def parser_inputs():
    """The reference's own parser fixtures: vapor_test/vapor_test.{vcf,bed} and simulate/Structural_Variants_*."""
    import glob
    base = os.path.dirname(os.path.dirname(REF_SF))
    files = [os.path.join(base, "vapor_test", "vapor_test.vcf"), os.path.join(base, "vapor_test", "vapor_test.bed")]
    for d in ("Structural_Variants_het", "Structural_Variants_homo"):
        files += sorted(glob.glob(os.path.join(base, "simulate", d, "*.vcf"))) + sorted(glob.glob(os.path.join(base, "simulate", d, "*.bed")))
    return base, files


def bed_five_columns(text: str) -> str:
    """`chr start end TYPE` rows re-expressed as `chr start end SVID TYPE` (the 4-column form of the shipped files no longer
    parses: bed_info_readin reads pin[4], vapor_vali/vapor:31); 5-column rows stay as they are."""
    out = []
    for n, line in enumerate(text.splitlines()):
        f = line.split()
        out.append("\t".join(f[:3] + ["sv%d" % (n + 1), f[3]]) if len(f) == 4 else line)
    return "\n".join(out) + "\n"


def gen_parsers(m):
    """The reference's vcf_list_readin / bed_info_readin (vapor_vali/vapor:22-50, 127-202) on its own fixture files: per file
    the digest of what it returns (or the exception it raises), for the whole file and for its first 300 lines - the slices,
    and the two small vapor_test files whole, travel as fixture inputs so that the comparison also runs where the reference
    is not mounted."""
    cli = load_cli(m)
    base, files = parser_inputs()
    tmp = tempfile.mkdtemp(prefix="vapor_golden_parsers_")
    cases = []

    def run(kind, text):
        path = os.path.join(tmp, "in." + kind)
        open(path, "w").write(text)
        if kind == "vcf":
            r = call(cli["vcf_list_readin"], path)
        else:
            r = call(cli["bed_info_readin"], path, os.path.join(tmp, "figs") + "/")
        if "error" in r:
            return {"error": r["error"]}
        res = r["ok"]
        if kind == "vcf":
            return {"digest": parser_digest([res[0], sorted(res[1].items())]), "buckets": {k: len(v) for k, v in res[0].items()},
                    "records_keyed": len(res[1])}
        return {"digest": parser_digest(res), "rows": len(res)}

    for f in files:
        rel = os.path.relpath(f, base)
        kind = "vcf" if f.endswith(".vcf") else "bed"
        text = open(f).read()
        small = len(text) < 20000
        head = text if small else "".join(text.splitlines(True)[:300])
        c = {"file": rel, "kind": kind, "input_sha256": hashlib.sha256(text.encode()).hexdigest(), "lines": text.count("\n"),
             "whole": run(kind, text), "slice_lines": head.count("\n"), "slice": run(kind, head), "slice_text": head}
        if kind == "bed":
            c["whole_5col"] = run(kind, bed_five_columns(text))
            c["slice_5col"] = run(kind, bed_five_columns(head))
        cases.append(c)
        print("  %s: %s" % (rel, {k: v for k, v in c["whole"].items() if k != "digest"}))
    dump("parsers.json.gz", {"source": "vcf_list_readin / bed_info_readin, vapor_vali/vapor:22-50, 127-202, on vapor_test/ and simulate/Structural_Variants_*",
                             "note": "digest = sha256 of the canonical JSON of the returned structure (vcf: [buckets in first-seen order, sorted "
                                     "record-index map]); *_5col: 4-column BED rows re-expressed as chr start end SVID TYPE",
                             "cases": cases})


def main():
    os.makedirs(OUT, exist_ok=True)
    m = load_reference()
    which = sys.argv[1:] or ["kmerhits", "cleaners", "scorers", "window", "genotype", "io", "locus", "vcf", "other", "config1", "melt", "complex", "deep", "figures", "parsers"]
    for w in which:
        print("== " + w)
        globals()["gen_" + w](m)


if __name__ == "__main__":
    main()
