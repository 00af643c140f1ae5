"""join time against the number of join tasks (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
ss = eng.seqset(w.seqs)
for jt in (256, 384, 512, 768, 1024, 1536, 2048, 4000):
    eng.set_param("join_tasks", jt)
    plan = eng.plan(ss, w.pairs)
    for _ in range(3):
        plan.run()
    tj, tc = [], []
    for _ in range(15):
        plan.run(); t = plan.timings(); tj.append(t["join_ms"]); tc.append(t["clean_ms"])
    print("join_tasks %5d  join %.4f ms  clean %.4f ms  launches %d" % (jt, np.median(tj), np.median(tc), t["join_launches"]), flush=True)
    plan.close()
