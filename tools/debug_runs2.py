import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd.engine import Engine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
eng = Engine(0)
cases = load_golden("window.json.gz")["cases"]
seqs, rows, exp = [], [], []
for c in cases:
    s = "".join(ch for ch in c["seq"] if ch != "X")
    for step, tr in enumerate(c["qc_trace"]):
        seqs.append(s); rows.append((len(seqs) - 1, len(seqs) - 1, 0, 10 + 10 * step, 0)); exp.append(tr)
for t in (18, 27):
    ss = eng.seqset([seqs[t]])
    p = eng.plan(ss, eng.make_pairs([(0, 0, 0, rows[t][3], 0)]))
    for rep in range(3):
        st = p.run()
        print("row", t, "rep", rep, "stats0", st[0, 0], "st14", st[0, 14], "status", st[0, 15], "records", p.record_counts()[0], p.timings()["retried_pairs"], flush=True)
