"""Records per pair and the selfplot golden rows (GPU box, ad hoc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden

eng = Engine(0)
w = wl.make_workload("cfg2", seed=1000, **wl.WORKLOADS["cfg2"])
plan = eng.plan(eng.seqset(w.seqs), w.pairs)
st = plan.run()
rc = plan.record_counts()
print("dots/pair mean %.0f  records/pair mean %.0f max %d p90 %.0f  dots/record %.2f" % (st[:, 0].mean(), rc.mean(), rc.max(), np.percentile(rc, 90), st[:, 0].sum() / rc.sum()))
cases = load_golden("window.json.gz")["cases"]
seqs, rows, exp = [], [], []
for c in cases:
    s = "".join(ch for ch in c["seq"] if ch != "X")
    for step, tr in enumerate(c["qc_trace"]):
        seqs.append(s); rows.append((len(seqs) - 1, len(seqs) - 1, 0, 10 + 10 * step, 0)); exp.append(tr)
ss = eng.seqset(seqs)
p2 = eng.plan(ss, eng.make_pairs(rows))
st = p2.run()
rc2 = p2.record_counts()
bad = 0
for t, (g, e) in enumerate(zip(st[:, [0, 7, 8]].tolist(), exp)):
    if g != e:
        bad += 1
        if bad < 12:
            print("row", t, "k", rows[t][3], "len", len(seqs[t]), "got", g, "exp", e, "records", rc2[t], "status", st[t, 15])
print("bad rows", bad, "of", len(exp))
