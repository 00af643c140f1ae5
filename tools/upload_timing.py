"""What the upload of a cfg2 batch's sequences costs (Engine.seqset: pointers, staging, transfer, pack kernel) beside the
plain transfer of as many bytes.  GPU box: python tools/upload_timing.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
nbytes = sum(map(len, w.seqs))
for _ in range(3):
    eng.seqset(w.seqs).close()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    ss = eng.seqset(w.seqs)
    ts.append(time.perf_counter() - t0)
    ss.close()
print("%s: %d sequences, %.1f MB: seqset %.3f ms best, %.3f median" % (name, len(w.seqs), nbytes / 1e6, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
best = 1e9
for _ in range(10):
    t0 = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("plain pinned transfer of as many bytes: %.3f ms (%.1f GB/s)" % (best * 1e3, nbytes / best / 1e9))
blob = np.frombuffer("".join(w.seqs).encode(), dtype=np.uint8)
stage = np.empty_like(blob)
best = 1e9
for _ in range(10):
    t0 = time.perf_counter(); np.copyto(stage, blob); best = min(best, time.perf_counter() - t0)
print("one core copying as many bytes: %.3f ms (%.1f GB/s)" % (best * 1e3, nbytes / best / 1e9))
