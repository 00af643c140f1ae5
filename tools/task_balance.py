"""Per-workgroup ticks of join_kernel against what each task holds (GPU box; build: python tools/phase_timing.py --build).

  VAPOR_HIP_LIB=tools/libvapor_hip_phases.so python tools/task_balance.py [cfg2]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VAPOR_HIP_LIB", os.path.join(ROOT, "tools", "libvapor_hip_phases.so"))
import numpy as np
from vapor_amd import _lib as L
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w = wl.make_workload(name, seed=seed, **wl.WORKLOADS[name])
eng = Engine(0)
ss = w.upload(eng)          # (derived alt windows: the plan shares its joins, as bench.py runs it)
plan = eng.plan(ss, w.pairs)
lib = L.load()
for _ in range(3):
    st = plan.run()
lib.vapor_debug_block_ticks.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int32]
bt = np.zeros(4096)
lib.vapor_debug_block_ticks(L.ptr(bt, ctypes.c_double), 4096)
first = np.zeros(4096, dtype=np.int32); nr = np.zeros(4096, dtype=np.int32); order = np.zeros(len(w.pairs), dtype=np.int32)
lib.vapor_debug_plan_tasks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
nt = lib.vapor_debug_plan_tasks(plan._h, first.ctypes.data, nr.ctypes.data, 4096, order.ctypes.data, len(order))
dots = st[:, 0]
recs = plan.record_counts()
lens = np.array([len(s) for s in w.seqs])
rows = []
for t in range(nt):
    idx = order[first[t]:first[t] + nr[t]]
    # (an index past the caller's pairs is a shared join: read r against the hidden sequence of its locus' group - in these
    # workloads every read has its two pairs, so shared join r serves pairs 2r and 2r + 1)
    n0 = len(w.pairs)
    own = np.array([i for i in idx if i < n0], dtype=np.int64)
    shared = np.array([i - n0 for i in idx if i >= n0], dtype=np.int64)
    alleles = len(set(int(w.pairs["seq2"][i]) for i in own) | set(-1 - int(w.read_locus[r]) for r in shared))
    rows.append((bt[t], nr[t], alleles, int(dots[own].sum() + dots[2 * shared].sum()), int(recs[own].sum() + recs[2 * shared].sum()),
                 int(lens[w.pairs["seq1"][own]].sum() + lens[w.pairs["seq1"][2 * shared]].sum())))
rows = np.array(rows, dtype=np.float64)
print("tasks %d  ticks: min %.0f mean %.0f max %.0f  max/mean %.3f" % (nt, rows[:, 0].min(), rows[:, 0].mean(), rows[:, 0].max(), rows[:, 0].max() / rows[:, 0].mean()))
# least squares: ticks ~ a*alleles + b*read symbols + c*dots + d*records
A = np.stack([rows[:, 2], rows[:, 5], rows[:, 3], rows[:, 4]], axis=1)
coef, res, *_ = np.linalg.lstsq(A, rows[:, 0], rcond=None)
pred = A @ coef
print("fit ticks = %.1f*tables + %.5f*read_symbols + %.5f*dots + %.5f*records   (rms residual %.0f of mean %.0f)" % (*coef, np.sqrt(np.mean((pred - rows[:, 0]) ** 2)), rows[:, 0].mean()))
o = np.argsort(-rows[:, 0])
print("slowest / fastest tasks: ticks reads tables dots records read_symbols")
for t in list(o[:8]) + list(o[-5:]):
    print("  %8.0f %3d %2d %8d %7d %8d" % tuple(rows[t]))
print("dots per pair: min %d median %d mean %d max %d" % (dots.min(), np.median(dots), dots.mean(), dots.max()))
print("mean ticks by workgroup index mod 8 (XCD):", " ".join("%.0f" % rows[x::8, 0].mean() for x in range(8)))
print("mean ticks by number of tables:", {int(k): round(float(rows[rows[:, 2] == k, 0].mean())) for k in np.unique(rows[:, 2])})
print("ticks in workgroup order:")
for x in range(0, nt, 16):
    print("  " + " ".join("%5.0f" % (v / 100) for v in rows[x:x + 16, 0]))
# where every workgroup ran, and what its waves spent their time on
lib.vapor_debug_block_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
info = np.zeros((4096, 4)); phase = np.zeros((4096, 8))
# one clean run so that the per-workgroup phase sums are of a single launch
lib.vapor_debug_block_info(info.ctypes.data, phase.ctypes.data, 4096)
plan.run()
lib.vapor_debug_block_info(info.ctypes.data, phase.ctypes.data, 4096)
lib.vapor_debug_block_ticks(L.ptr(bt, ctypes.c_double), 4096)
hw = info[:nt, 0].astype(np.uint64); xcc = info[:nt, 1].astype(np.uint64) & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
t0 = info[:nt, 2] - info[:nt, 2].min(); t1 = info[:nt, 3] - info[:nt, 2].min()
key = [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)]
from collections import defaultdict
byc = defaultdict(list)
for t in range(nt):
    byc[key[t]].append(t)
print("distinct (xcc, se, sh, cu) slots used: %d for %d workgroups" % (len(byc), nt))
multi = {k: v for k, v in byc.items() if len(v) > 1}
print("slots that ran more than one workgroup: %d" % len(multi))
for k, v in list(multi.items())[:12]:
    print("  xcc %d se %d sh %d cu %2d: " % k + "  ".join("wg %3d [%6.0f..%6.0f]" % (t, t0[t], t1[t]) for t in v))
slow = bt[:nt] > 1.3 * np.median(bt[:nt])
print("slow workgroups (> 1.3 x median): %d; of them sharing a slot with another workgroup: %d" % (slow.sum(), sum(1 for t in range(nt) if slow[t] and len(byc[key[t]]) > 1)))
ph = phase[:nt]
names = ["build", "staging", "lookup", "scans", "fill", "verify", "tail"]
print("phase ticks per workgroup (sum over its 16 waves), fast vs slow workgroups:")
for x, nm in enumerate(names):
    print("  %-8s fast %9.0f   slow %9.0f   ratio %.2f" % (nm, ph[~slow, x].mean(), ph[slow, x].mean() if slow.any() else 0, (ph[slow, x].mean() / max(ph[~slow, x].mean(), 1)) if slow.any() else 0))
print("start offsets (ticks after the first workgroup): median %.0f  p90 %.0f  max %.0f" % (np.median(t0), np.percentile(t0, 90), t0.max()))
print("per xcc: workgroups, distinct cus, mean ticks")
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %3d wgs on %2d cus, mean %.0f, slow %d" % (x, m.sum(), len(set(k for k, mm in zip(key, m) if mm)), bt[:nt][m].mean(), slow[m].sum()))
cyc = ph.sum(axis=1) / 16.0
print("per xcc: shader cycles per wave / wall time of the workgroup = effective clock (GHz)")
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %.2f GHz (min %.2f max %.2f)  cycles/wave mean %.0f" % (x, (cyc[m] / (bt[:nt][m] * 10.0)).mean(), (cyc[m] / (bt[:nt][m] * 10.0)).min(), (cyc[m] / (bt[:nt][m] * 10.0)).max(), cyc[m].mean()))

# ---- the same in the steady state: many passes enqueued back to back (no host gap, clocks stay up), one plan -----
if "--steady" in sys.argv:
    import time
    plan.set_reads(wl.read_table(w), w.n_loci)
    plan.run_loci(want_host=False)
    for rep in range(2):
        t_0 = time.perf_counter()
        NP = 400
        for _ in range(NP):
            plan.run_loci_async()
        plan.sync(want_host=False)
        dt = time.perf_counter() - t_0
        print("steady: %d passes in %.3f s = %.4f ms per pass; timings %s" % (NP, dt, dt / NP * 1e3, {k: round(float(v), 4) for k, v in plan.timings().items() if k.endswith("_ms")}))
    lib.vapor_debug_block_info(info.ctypes.data, phase.ctypes.data, 4096)
    lib.vapor_debug_block_ticks(L.ptr(bt, ctypes.c_double), 4096)
    b = bt[:nt]
    print("steady, last pass: workgroup ticks min %.0f mean %.0f p90 %.0f max %.0f  max/mean %.3f" % (b.min(), b.mean(), np.percentile(b, 90), b.max(), b.max() / b.mean()))
    xcc = info[:nt, 1].astype(np.uint64) & 15
    print("  by xcc: " + " ".join("%.0f" % b[xcc == x].mean() for x in range(8)))
    t0 = info[:nt, 2] - info[:nt, 2].min(); t1 = info[:nt, 3] - info[:nt, 2].min()
    print("  start offsets median %.0f max %.0f; end median %.0f max %.0f" % (np.median(t0), t0.max(), np.median(t1), t1.max()))
