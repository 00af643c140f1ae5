"""Fresh-interpreter body of tests/test_host_cpu.py::test_three_threads_into_a_cold_repeat_check.

Round 4's hang (gpurun_out/r4_hang.log): three chunk threads of a run met their first tandem-duplication window at the same
moment and imported scikit-learn / SciPy against each other inside repeat_qc - and never came back.  The fix (8596378) makes
the first caller import everything under a lock (repeat_qc._warm).  Here three threads enter a COLD repeat_qc.cluster_sizes
together; an import hook counts how many threads are inside a first import of the clustering libraries at once and holds
each such import open for a moment, so that threads that CAN overlap there DO.  Prints one JSON line.
`nowarm` as argument removes the fix (the test's negative control: the hook then sees the threads overlap)."""
import builtins
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler

faulthandler.dump_traceback_later(int(os.environ.get("HANG_CASE_SECONDS", "100")), exit=True)
import numpy as np

from vapor_amd import repeat_qc

assert "sklearn" not in sys.modules and "scipy.cluster" not in sys.modules, "the repeat check's libraries must be cold here"
if len(sys.argv) > 1 and sys.argv[1] == "nowarm":
    repeat_qc._warm = lambda: None

WATCH = ("sklearn", "scipy.cluster", "scipy.spatial", "threadpoolctl")
_real_import = builtins.__import__
_guard = threading.Lock()
_inside = {}
peak = [0]


def _hook(name, globals=None, locals=None, fromlist=(), level=0):
    first = level == 0 and name.startswith(WATCH) and name not in sys.modules
    if not first:
        return _real_import(name, globals, locals, fromlist, level)
    me = threading.get_ident()
    with _guard:
        _inside[me] = _inside.get(me, 0) + 1
        peak[0] = max(peak[0], len(_inside))
    try:
        if _inside[me] == 1:
            time.sleep(0.05)                     # (a thread that could enter beside this one gets the time to)
        return _real_import(name, globals, locals, fromlist, level)
    finally:
        with _guard:
            _inside[me] -= 1
            if not _inside[me]:
                del _inside[me]


builtins.__import__ = _hook
rng = np.random.default_rng(1)
js = np.concatenate([rng.integers(500, 600, 80), rng.integers(1500, 1600, 80)]).tolist()
is_ = np.concatenate([rng.integers(100, 200, 80), rng.integers(900, 1000, 80)]).tolist()
out = [None] * 3
bar = threading.Barrier(3)


def work(k):
    bar.wait()
    out[k] = [float(x) for x in repeat_qc.cluster_sizes(js, is_)]


th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
builtins.__import__ = _real_import
print(json.dumps({"seconds": round(time.perf_counter() - t0, 2), "threads_importing_at_once": peak[0], "sizes": out}))
