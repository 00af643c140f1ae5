"""Ad-hoc timing of the cfg2 batch under different pair flags / reads_per_task (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
ss = eng.seqset(w.seqs)
for rpt in (512, 256, 1024, 200):
    eng.set_param("join_tasks", rpt)
    for fl in (None, 0, 1, 2, 3, 5, 7):
        pairs = w.pairs.copy()
        if fl is not None:
            pairs["flags"] = fl
        plan = eng.plan(ss, pairs)
        for _ in range(3):
            plan.run()
        tj = tc = tt = 0.0
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            plan.run()
            tm = plan.timings()
            tj += tm["join_ms"]; tc += tm["clean_ms"]; tt += tm["total_ms"]
        wall = (time.perf_counter() - t0) / n * 1e3
        print("jt=%4d flags=%s join=%.3f clean=%.3f dev=%.3f wall=%.3f ms" % (rpt, fl, tj / n, tc / n, tt / n, wall), flush=True)
        plan.close()
        if rpt != 512 and fl is None:
            break
