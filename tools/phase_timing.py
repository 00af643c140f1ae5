"""Per-phase tick breakdown of join_kernel / clean_kernel on the cfg2 batch (GPU box).

Build first (here or on the box):  python tools/phase_timing.py --build
Run on the box:                    VAPOR_HIP_LIB=tools/libvapor_hip_phases.so python tools/phase_timing.py
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libvapor_hip_phases.so")

if "--build" in sys.argv:
    from vapor_amd import build as B
    cmd = [B.hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + B.EXTRA_FLAGS + ["-DVAPOR_DEV_BUILD", "-DVAPOR_PHASE_TIMING",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vapor_amd", "csrc"),
           "-Wno-unused-function", "-o", SO] + B.SOURCES + ["-lz"]
    subprocess.check_call(cmd)
    print(SO)
    sys.exit(0)

os.environ.setdefault("VAPOR_HIP_LIB", SO)
import numpy as np
from vapor_amd import _lib as L
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

NAMES = {0: "join: table build (+wait prev probe)", 1: "join: strip staging", 2: "join: keys + bucket reads issued",
         3: "join: scans", 4: "join: queue fill", 5: "join: verify + store", 6: "join: tail wait", 7: "join: CANDIDATES (count, not ticks)",
         8: "clean: pass 0 (stage, bitmap i-j)", 9: "clean: body total", 10: "clean: flag write-back",
         16: "clean: axis1 bitmap", 17: "clean: axis1 starts+ranks", 18: "clean: axis1 sizes", 19: "clean: axis1 flags",
         20: "clean: axis2 bitmap", 21: "clean: axis2 starts+ranks", 22: "clean: axis2 sizes", 23: "clean: axis2 flags",
         32: "dir: level 1", 33: "dir: level 2 loops", 34: "dir: median", 35: "dir: final sums"}

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
ss = w.upload(eng)          # (derived alt windows: the plan shares its joins, as bench.py runs it)
plan = eng.plan(ss, w.pairs)
lib = L.load()
lib.vapor_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int32]
out = np.zeros(64)
for _ in range(3):
    plan.run()
lib.vapor_debug_phases(L.ptr(out, ctypes.c_double), 64)
N = 10
for _ in range(N):
    plan.run()
lib.vapor_debug_phases(L.ptr(out, ctypes.c_double), 64)
print("timings", plan.timings())
for grp, ks in (("join", range(0, 8)), ("clean", (8, 9, 10)), ("clean.axis", range(16, 24)), ("dir", range(32, 36))):
    tot = sum(out[k] for k in ks) or 1.0
    for k in ks:
        if out[k]:
            print("%-40s %14.0f ticks/run  %5.1f %% of %s" % (NAMES.get(k, str(k)), out[k] / N, 100 * out[k] / tot, grp))

lib.vapor_debug_block_ticks.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int32]
bt = np.zeros(4096)
lib.vapor_debug_block_ticks(L.ptr(bt, ctypes.c_double), 4096)
bt = bt[bt > 0]
if len(bt):
    print("join workgroups: %d  ticks min %.0f  mean %.0f  p90 %.0f  max %.0f   max/mean %.3f" % (len(bt), bt.min(), bt.mean(), np.percentile(bt, 90), bt.max(), bt.max() / bt.mean()))
