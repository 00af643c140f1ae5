"""The read extraction on the device (vapor_bam_chop_device) against the host's (vapor_bam_chop) on a seeded world's BAM file:
kept reads, miss_bp, and the bases themselves (the bit planes of a mixed sequence set made from the device addresses against
those of a set uploaded from the host's strings), then both timed.
  python tools/bamdev_probe.py [n_loci] [block_size] [--qual]      (--qual: random qualities instead of 0xFF - literal-heavy blocks)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vapor_amd import seqio, synth, pipeline

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 300
block = int(args[1], 0) if len(args) > 1 else 0xFF00
w = synth.make_world(seed=11, n_loci=n, svtypes=("DEL", "DEL", "INV", "INS"), span_range=(100, 4000), read_len=9500, n_reads=20)
for c in w.reads:
    w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
tmp = tempfile.mkdtemp()
fa, bam = synth.write_world_files(w, tmp, block_size=block, qual_seed=(7 if '--qual' in sys.argv else None))
print("files of %d loci: %.1f MB BAM, blocks of %d" % (n, os.path.getsize(bam) / 1e6, block), flush=True)
rows = [l.split("\t") for l in synth.bed_text(w).strip().splitlines()]
chroms = [r[0] for r in rows]
st = np.asarray([max(int(r[1]) - 500, 1) for r in rows], dtype=np.int64)
en = np.asarray([int(r[2]) + 500 for r in rows], dtype=np.int64)
fl = np.full(len(rows), 500, dtype=np.int64)
be = seqio.InProcessBam()
eng = pipeline.get_engine()
b = be._open(bam)

# ---- host answer, region by region ------------------------------------------------------------------------------------------
host = []
for g in range(len(rows)):
    r = b.chop_native_raw(chroms[g], int(st[g]), int(en[g]), int(fl[g]))
    if r is None:
        host.append(([], []))
        continue
    whole, off, ln, miss = r
    order = np.arange(len(off))
    if len(order) > 20:
        order = np.argsort(miss, kind="stable")[:20]
    host.append(([whole[int(off[i]):int(off[i]) + int(ln[i])] for i in order], [int(miss[i]) for i in order]))

# ---- device answer ------------------------------------------------------------------------------------------------------------
kf, addr, q0, miss, status, keep = be.chop_many_device(eng, bam, chroms, st, en, fl)
print("device: %d regions, %d kept reads, status counts %s" % (len(rows), int(kf[-1]), dict(zip(*np.unique(status, return_counts=True)))), flush=True)
bad = 0
lens = []
for g in range(len(rows)):
    if status[g]:
        continue
    a, e = int(kf[g]), int(kf[g + 1])
    hr, hm = host[g]
    if e - a != len(hr) or list(map(int, miss[a:e])) != hm:
        bad += 1
        if bad <= 5:
            print("region %d: device %d reads miss %s, host %d reads miss %s" % (g, e - a, miss[a:e].tolist(), len(hr), hm))
    for i in range(a, e):
        lens.append(int(en[g] - st[g] - miss[i]))
print("regions whose kept reads / miss_bp differ: %d" % bad, flush=True)
if bad == 0 and len(addr):
    lens = np.asarray(lens, dtype=np.int64)
    dev = eng.seqset_raw(addr, lens, None, keepalive=keep, src_kind=np.ones(len(addr), dtype=np.uint8), src_first=q0)
    flat = [r for g in range(len(rows)) if not status[g] for r in host[g][0]]
    ref = eng.seqset(flat)
    diff = 0
    for i in range(len(flat)):
        pa, pb = dev.planes(i), ref.planes(i)
        if not all(np.array_equal(x, y) for x, y in zip(pa, pb)):
            diff += 1
    print("reads whose bit planes differ from the host upload's: %d of %d; n_exc equal %s, n_invalid equal %s" % (
        diff, len(flat), np.array_equal(dev.n_exc, ref.n_exc), np.array_equal(dev.n_invalid, ref.n_invalid)), flush=True)
    dev.close(); ref.close()
for k in keep:
    k.close()

# ---- timing --------------------------------------------------------------------------------------------------------------------
for name, fn in (("device", lambda: be.chop_many_device(eng, bam, chroms, st, en, fl)), ("host", lambda: be.chop_many(bam, chroms, st, en, fl))):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        for k in (out[5] if name == "device" else []):
            k.close()
        best = min(best, dt)
    print("%s chop of %d regions: %.4f s -> %.0f loci/s" % (name, len(rows), best, len(rows) / best), flush=True)
