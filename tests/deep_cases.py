"""Loci of 33-64 reads scored read by read by the reference (tests/golden/deep_loci.json.gz, oracle/gen_golden.py
gen_deep): inputs for the device plan and the expected per-read scores / QS / GS / GT / GQ."""
import numpy as np

from conftest import load_golden
from vapor_amd import _lib as L

DEEP = load_golden("deep_loci.json.gz")["cases"]
FLAGS = {"DEL": L.PF_C1 | L.PF_C2, "TANDUP": L.PF_C1 | L.PF_DIR, "INV": L.PF_C1, "INS": L.PF_C1}
KIND = {"DEL": 0, "TANDUP": 3, "INV": 1, "INS": 1}


def build():
    """(seqs, pairs rows, READ_DTYPE table, n_loci) with locus l = case l; read r uses pairs 2r (ref), 2r+1 (alt)."""
    seqs, rows, reads = [], [], []
    for li, c in enumerate(DEEP):
        ri, ai = len(seqs), len(seqs) + 1
        seqs += [c["ref"], c["alt"]]
        for x in c["reads"]:
            seqs.append(x[0])
            q = len(seqs) - 1
            rows.append((q, ri, int(x[1]), int(c["k"]), FLAGS[c["svtype"]]))
            rows.append((q, ai, int(x[1]), int(c["k"]), FLAGS[c["svtype"]]))
            reads.append((KIND[c["svtype"]], li, len(c["ref"]), len(c["alt"])))
    t = np.zeros(len(reads), dtype=L.READ_DTYPE)
    n = len(reads)
    t["ref_a"] = t["ref_b"] = 2 * np.arange(n)
    t["alt_a"] = t["alt_b"] = 2 * np.arange(n) + 1
    t["kind"] = [r[0] for r in reads]
    t["locus"] = [r[1] for r in reads]
    t["len_ref"] = [r[2] for r in reads]
    t["len_alt"] = [r[3] for r in reads]
    return seqs, rows, t, len(DEEP)


def expected(c):
    """(scores, QS, GS, GT index, GQ) the reference produced."""
    org, gt = c["organize"], c["gt"]
    return ([float(v) for v in c["scores"]], float(org[1]), float(org[2]), ["0/0", "0/1", "1/1"].index(gt[0]), float(gt[1]))
