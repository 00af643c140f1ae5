"""Upload of a cfg2 batch (Engine.seqset with derived alt windows) by staging-thread count.  GPU box: python tools/upload_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

w = wl.make_workload("cfg2", seed=1000, **wl.WORKLOADS["cfg2"])
eng = Engine(0)
nbytes = sum(map(len, w.seqs[:w.n_lit]))
for thr in (1, 2, 4, 6, 8, 12, 16):
    eng.set_param("stage_threads", thr)
    for _ in range(3):
        w.upload(eng).close()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); ss = w.upload(eng); ts.append(time.perf_counter() - t0); ss.close()
    print("stage_threads %2d: %.1f MB of bytes + %d derived: %.3f ms best, %.3f median" % (thr, nbytes / 1e6, len(w.derived), min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), flush=True)
eng.set_param("shared_join", 0)
eng.set_param("stage_threads", 8)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); ss = w.upload(eng); ts.append(time.perf_counter() - t0); ss.close()
print("without share groups (8 threads): %.3f ms best" % (min(ts) * 1e3))
ts = []
for _ in range(15):
    t0 = time.perf_counter(); ss = eng.seqset(w.seqs); ts.append(time.perf_counter() - t0); ss.close()
print("every sequence as bytes (8 threads): %.3f ms best" % (min(ts) * 1e3))
