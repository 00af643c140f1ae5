"""Worker processes for the host-side work that is neither the device's nor worth the interpreter lock of the main process:
drawing the recurrence-plot PNGs (matplotlib, ~65 ms a figure) and the X-means clustering of the repeat check (sklearn /
scipy, ~1 ms a window).  `python -m vapor_amd.host_worker` processes - fresh interpreters that never load the HIP library -
each fed by one thread of a thread pool: a task writes a pickled (module, function, arguments) to its thread's process
and reads the pickled result back.  Child processes of our own rather than a multiprocessing pool: that one would import the
caller's main module again in every worker.

VAPOR_HOST_PROCS (or, older, VAPOR_FIGURE_PROCS): number of processes; default the usable cores divided by the ranks on the
host, minus one, at most 16; below 2 the callers do the work themselves."""
from __future__ import annotations

import os
import pickle
import struct
import subprocess
import sys
import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Optional

_pool: Optional["Workers"] = None
_pool_lock = threading.Lock()


def n_workers() -> int:
    want = os.environ.get("VAPOR_HOST_PROCS", os.environ.get("VAPOR_FIGURE_PROCS"))
    if want is not None:
        return max(0, int(want))
    from . import pipeline
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(0, min(16, pipeline._usable_cores() // ranks_here - 1))


class WorkerLost(RuntimeError):
    """The worker process could not be started or ended before it answered (the caller then does the work itself)."""


class Workers:
    def __init__(self, n: int):
        self.n = n
        self.pool = ThreadPoolExecutor(max_workers=n)
        self.tls = threading.local()
        self.lock = threading.Lock()
        self.procs: list = []

    def _proc(self):
        p = getattr(self.tls, "p", None)
        if p is None or p.poll() is not None:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), MPLBACKEND="Agg",
                       OMP_NUM_THREADS="1")
            p = subprocess.Popen([sys.executable, "-m", "vapor_amd.host_worker"], stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, env=env)
            self.tls.p = p
            with self.lock:
                self.procs.append(p)
        return p

    def _task(self, payload: bytes):
        try:
            p = self._proc()
            p.stdin.write(struct.pack("<q", len(payload)))
            p.stdin.write(payload)
            p.stdin.flush()
            tag = p.stdout.read(1)
        except OSError as e:
            raise WorkerLost("host worker: %s" % e) from e
        if tag not in (b"\x00", b"\x01"):
            raise WorkerLost("host worker ended (exit code %s)" % p.poll())
        # a worker that dies after the tag leaves a short header or body, or an envelope that does not unpickle: that is a
        # lost worker as well (the caller falls back to doing the work itself), and the process is not used again
        try:
            hdr = p.stdout.read(8)
            if len(hdr) != 8:
                raise EOFError("short header")
            n = struct.unpack("<q", hdr)[0]
            if n < 0 or n > (1 << 34):
                raise EOFError("implausible length %d" % n)
            raw = p.stdout.read(n)
            if len(raw) != n:
                raise EOFError("short body")
            body = pickle.loads(raw)
        except (OSError, EOFError, pickle.UnpicklingError, AttributeError, ImportError, IndexError, struct.error) as e:
            self._drop(p)
            raise WorkerLost("host worker answered with a damaged envelope: %s" % e) from e
        if tag == b"\x00":
            return body
        raise body if isinstance(body, BaseException) else RuntimeError("host worker: %s" % (body,))

    def _drop(self, p) -> None:
        """Ends a process whose stream can no longer be trusted; the thread starts a new one on its next task."""
        self.tls.p = None
        try:
            p.kill()
        except OSError:
            pass

    def submit(self, module: str, function: str, *args) -> Future:
        return self.pool.submit(self._task, pickle.dumps((module, function, args), protocol=4))

    def close(self) -> None:
        self.pool.shutdown(wait=True)
        with self.lock:
            procs, self.procs = self.procs, []
        for p in procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        # (called from an atexit handler: a worker that does not end on a closed stdin must not hang the interpreter's exit)
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    pass


def get() -> Optional[Workers]:
    """The process-wide pool, started on first use; None when the work is to be done by the caller."""
    global _pool
    if _pool is None:
        with _pool_lock:                           # (chunks of a run are scored on two threads)
            if _pool is None:
                n = n_workers()
                if n < 2:
                    return None
                import atexit
                _pool = Workers(n)
                atexit.register(shutdown)
    return _pool


def shutdown() -> None:
    global _pool
    if _pool is not None:
        p, _pool = _pool, None
        p.close()
