"""Does a clean workgroup beside a join workgroup pay?  The 'small' shape (2 kb reads x 4 kb windows: a clean workgroup needs
~12 KB of LDS) with one and with two plans in flight.  Round 3 ran it on developer builds whose join tile was 24 576 or
18 432 positions (a -DVAPOR_DEV_TA2 switch in JoinBig, removed again: 0 or 14.6 KB of a CU's LDS left beside a join
workgroup) and found no difference - profiles/r03_coresident_probe.txt.  usage (GPU box): python tools/coresident_probe.py [shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "small"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
ss = eng.seqset(w.seqs)
plans = []
for _ in range(2):
    p = eng.plan(ss, w.pairs); p.set_reads(wl.read_table(w), w.n_loci); p.run_loci(want_host=False); plans.append(p)
alone = plans[0].timings()
def timed(ps, n=60):
    for i in range(2 * len(ps)): ps[i % len(ps)].run_loci_async()
    for p in ps: p.sync(want_host=False)
    t0 = time.perf_counter()
    for i in range(n): ps[i % len(ps)].run_loci_async()
    for p in ps: p.sync(want_host=False)
    return (time.perf_counter() - t0) / n * 1e3
one, two = timed(plans[:1]), timed(plans)
print("%s: %d pairs; alone join %.4f clean %.4f ms; per pass one plan %.4f ms, two plans %.4f ms" % (name, len(w.pairs), alone["join_ms"], alone["clean_ms"], one, two))
