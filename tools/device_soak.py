"""Free device memory and host RSS over repeated runs of the in-memory pipeline in one process (a leak of device blocks,
pinned buffers or plans would show as a slope).  GPU box: python tools/device_soak.py [runs] [loci]"""
import ctypes, os, resource, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vapor_amd import cli, pipeline, seqio, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "TANDUP", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
tmp = tempfile.mkdtemp()
bed = os.path.join(tmp, "in.bed"); open(bed, "w").write(synth.bed_text(w))
seqio.set_backend(seqio.MemorySamtools(w))
info = cli.bed_info_readin(bed, tmp)
pipeline.get_engine()
hip = ctypes.CDLL(None)            # (the HIP runtime the library was loaded with: its symbols are global)
free, total = ctypes.c_size_t(), ctypes.c_size_t()
for r in range(runs):
    t0 = time.perf_counter()
    jobs = cli.bed_jobs(info, 3, "x.bam", "ref.fa", tmp + "/", "s")
    scores = cli.score_jobs(jobs, 512, None)
    dt = time.perf_counter() - t0
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    print("run %2d: %.3f s, device memory in use %.1f MB, max RSS %.0f MB" % (r, dt, (total.value - free.value) / 1e6,
          resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
