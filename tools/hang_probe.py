"""Runs tools/run_at_size.py with the given arguments and dumps every thread's Python stack if it is still running after
HANG_PROBE_SECONDS (default 90) - to find where a run hangs.  GPU box."""
import faulthandler, os, runpy, sys
faulthandler.dump_traceback_later(int(os.environ.get("HANG_PROBE_SECONDS", "90")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "tools", "run_at_size.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
