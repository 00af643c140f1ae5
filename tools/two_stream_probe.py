"""Would two plans on two streams (steps alternating) overlap one step's clean/finish with the next step's join?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
for S in (1, 2, 3):
    lanes = []
    for _ in range(S):
        st = torch.cuda.Stream()
        eng = Engine(0); eng.set_stream(st.cuda_stream)
        ss = eng.seqset(w.seqs); plan = eng.plan(ss, w.pairs); plan.set_reads(wl.read_table(w), w.n_loci)
        out = torch.empty((w.n_loci, 8), dtype=torch.float64, device="cuda")
        plan.run_loci(device_out=out.data_ptr())
        lanes.append((st, eng, ss, plan, out))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 60
        for i in range(K):
            st, eng, ss, plan, out = lanes[i % S]
            plan.run_loci_async(device_out=out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        tms = []
        for st, eng, ss, plan, out in lanes:
            plan.sync(want_host=False); tms.append(plan.timings())
        print("streams %d: %.4f ms/step   join %.4f clean %.4f (per-kernel event times)" % (S, dt, np.mean([t["join_ms"] for t in tms]), np.mean([t["clean_ms"] for t in tms])), flush=True)
    for st, eng, ss, plan, out in lanes:
        plan.close()
