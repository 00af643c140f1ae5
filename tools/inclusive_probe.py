"""bench.inclusive_rate (the very function of the bench line) in a process that differs from bench.py's in ONE thing at a time:
  plain      no torch in the process
  torch      torch imported, its HIP context initialised, one side stream made (what bench.py has done by then)
  resident   plain + the two resident plans of the timed region alive on the same engine
  after      plain + the timed region's passes run before (20 000 asynchronous steps), plans still alive
GPU box: python tools/inclusive_probe.py plain|torch|resident|after"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == "torch":
    import torch
    torch.cuda.init()
    _s = torch.cuda.Stream()
    _t = torch.zeros(16, device="cuda")
    torch.cuda.synchronize()
import bench
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
eng = Engine(0)
w = wl.make_workload("cfg2", seed=1000, **wl.WORKLOADS["cfg2"])
keep = []
if mode in ("resident", "after"):
    ss = w.upload(eng)
    for _ in range(2):
        p = eng.plan(ss, w.pairs)
        p.set_reads(wl.read_table(w), w.n_loci)
        p.run_loci()
        keep.append(p)
    if mode == "after":
        t0 = time.perf_counter()
        for _ in range(10000):
            for p in keep:
                p.run_loci_async()
        for p in keep:
            p.sync()
        print("after: 20000 steps in %.2f s" % (time.perf_counter() - t0), flush=True)
for rep in range(2):
    r = bench.inclusive_rate(eng, w, wl)
    print(mode, rep, "in flight, %d batches a thread: %.3f ms/batch -> %.0f loci/s (first 8: %.3f ms, second half: %.3f ms); one at a time %.3f ms %s" % (
        r["in_flight_batches_per_thread"], r["ms_per_batch_in_flight"], r["value"], r["first_8_batches"]["ms_per_batch"], r["second_half"]["ms_per_batch"],
        r["one_at_a_time"]["ms_per_batch"], {k: round(float(v), 3) for k, v in r["one_at_a_time"]["ms"].items()}), flush=True)
