"""Wall time of every join workgroup (a -DVAPOR_BLOCK_TIMING build: two clock reads per workgroup, nothing else).
   python tools/block_times.py --build ; then on the GPU box: python tools/block_times.py [workload]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libvapor_hip_blocks.so")
if "--build" in sys.argv:
    from vapor_amd import build as B
    subprocess.check_call([B.hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + B.EXTRA_FLAGS + ["-DVAPOR_DEV_BUILD", "-DVAPOR_BLOCK_TIMING",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vapor_amd", "csrc"),
                           "-Wno-unused-function", "-o", SO] + B.SOURCES + ["-lz"])
    print(SO); sys.exit(0)
os.environ["VAPOR_HIP_LIB"] = SO
import numpy as np
from vapor_amd import _lib as L
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
eng = Engine(0)
plan = eng.plan(eng.seqset(w.seqs), w.pairs)
lib = L.load()
lib.vapor_debug_block_ticks.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int32]
for _ in range(5):
    st = plan.run()
print("timings", plan.timings())
bt = np.zeros(4096)
lib.vapor_debug_block_ticks(L.ptr(bt, ctypes.c_double), 4096)
bt = bt[bt > 0] / 100.0          # us
print("join workgroups %d: us min %.1f mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f  max/mean %.3f" %
      (len(bt), bt.min(), bt.mean(), np.median(bt), np.percentile(bt, 90), np.percentile(bt, 99), bt.max(), bt.max() / bt.mean()))
print("sorted tail:", np.sort(bt)[-12:].round(1).tolist())
print("by index (first 40):", bt[:40].round(0).astype(int).tolist())
