"""Why does an RCCL communicator made EARLY slow this process's clean / finish kernels (VERDICT r03 item 3 iii)?  One rank, one GPU.
  base         no process group
  eager        init_process_group(nccl, device_id=...) (communicator made at once), THEN the library context and plans
  eager_after  the library context, plans and a warm pass first, THEN the eager process group
  lazy_first   init_process_group(nccl) without device_id, one all_reduce (communicator made there), THEN the library
  lazy_after   the library first, then the lazy process group and its first all_reduce   (what bench.py does)
  env_fg       no process group; HSA_FORCE_FINE_GRAIN_PCIE=1 in the environment before anything loads
Prints per-pass wall time, the kernels' HIP-event times, and the C-level values of a few environment variables afterwards."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == "env_fg":
    os.environ["HSA_FORCE_FINE_GRAIN_PCIE"] = "1"
import torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
import torch.distributed as dist
from vapor_amd import workload as wl
from vapor_amd.engine import Engine

def group(eager):
    if eager:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()

def library():
    eng = Engine(0)
    w = wl.make_workload("cfg2", seed=1000, **wl.WORKLOADS["cfg2"])
    p = eng.plan(w.upload(eng), w.pairs)
    p.set_reads(wl.read_table(w), w.n_loci)
    p.run_loci()
    return eng, w, p

if mode in ("eager", "lazy_first"):
    group(mode == "eager")
eng, w, p = library()
if mode in ("eager_after", "lazy_after"):
    group(mode == "eager_after")
for _ in range(200):
    p.run_loci_async()
p.sync()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    p.run_loci_async()
p.sync()
dt = time.perf_counter() - t0
tm = p.timings()
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
env = {k: (libc.getenv(k.encode()) or b"").decode() for k in ("HSA_FORCE_FINE_GRAIN_PCIE", "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_HOST_COHERENT", "HSA_ENABLE_SDMA", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG")}
print("%-11s one plan, %d passes: %.4f ms a pass; join %.4f clean %.4f finish %.4f ms; env now %s" % (
    mode, n, dt / n * 1e3, tm["join_ms"], tm["clean_ms"], tm["finish_ms"], {k: v for k, v in env.items() if v}), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
