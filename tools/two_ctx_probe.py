"""Do two library contexts on two Python threads overlap?  Uploads alone, then whole batches.  GPU box."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import workload as wl, _lib as L
from vapor_amd.engine import Engine

w = wl.make_workload("cfg2", seed=1000, **wl.WORKLOADS["cfg2"])
engs = [Engine(0), Engine(0), Engine(0)]
table = wl.read_table(w)


def upload(e):
    w.upload(e).close()


def batch(e):
    ss = w.upload(e)
    p = e.plan(ss, w.pairs)
    p.set_reads(table, w.n_loci)
    p.run_loci()
    p.close(); ss.close()


def plan_only(e, ss):
    p = e.plan(ss, w.pairs)
    p.set_reads(table, w.n_loci)
    p.run_loci()
    p.close()


for name, fn in (("upload", upload), ("batch", batch)):
    for e in engs:
        fn(e); fn(e)
    for nthr in (1, 2, 3):
        n_each = 12
        def work(e):
            for _ in range(n_each):
                fn(e)
        th = [threading.Thread(target=work, args=(engs[k],)) for k in range(nthr)]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t0
        print("%-7s %d thread(s): %.3f ms per call overall (%.3f ms per thread-call)" % (name, nthr, dt / (n_each * nthr) * 1e3, dt / n_each * 1e3), flush=True)
sets = [w.upload(e) for e in engs]
for nthr in (1, 2):
    n_each = 12
    def work(k):
        for _ in range(n_each):
            plan_only(engs[k], sets[k])
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print("plan+run %d thread(s): %.3f ms per call overall" % (nthr, dt / (n_each * nthr) * 1e3), flush=True)
