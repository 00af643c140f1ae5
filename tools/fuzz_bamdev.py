"""Fuzz of the read extraction on the device against the host's: random BAM files (block sizes 700 B - 64 KB, every zlib level and
strategy incl. stored and fixed-code blocks, reads of 300 - 20 000 bases with clips / insertions / deletions / N, seeded or absent
qualities, now and then a read whose CIGAR goes to CG:B,I) and random regions around and between the reads; for every region
the device answers, the kept reads, miss_bp and bases (bit planes) are the host reader's - tests/test_gpu_bamdev.py's `compare`.
  python tools/fuzz_bamdev.py [seconds] [seed]"""
import os, struct, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vapor_amd import bamio, synth
from vapor_amd.engine import Engine
import test_gpu_bamdev as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng = Engine(0)
t_end = time.time() + budget
files = regions_n = reads_n = host_route = 0
orig_block = bamio._bgzf_block
while time.time() < t_end:
    level = int(rng.integers(0, 10))
    strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]

    def block(data, level=level, strat=strat):
        comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
        cdata = comp.compress(data) + comp.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25)
                + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    bs = int(rng.choice([700, 1500, 4096, 16384, 30000, 0xFF00]))
    if level == 0 or strat in (zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED, zlib.Z_RLE):
        bs = min(bs, 30000)                                  # (the payload must fit BSIZE)
    n_contigs = int(rng.integers(1, 4))
    refs, recs = [], []
    for c in range(n_contigs):
        clen = int(rng.integers(20000, 120000))
        contig = synth.random_dna(rng, clen)
        refs.append(("k%d" % c, clen))
        for i in range(int(rng.integers(20, 90))):
            n = int(rng.integers(300, 20000))
            pos = int(rng.integers(0, max(clen - 400, 1)))
            read, cg = synth.mutate(rng, contig[pos:pos + n])
            pre = int(rng.integers(0, 60)) if rng.random() < 0.4 else 0
            lead = "%dD" % int(rng.integers(1, 300)) if rng.random() < 0.15 else ""
            seq = synth.random_dna(rng, pre) + read
            if rng.random() < 0.1:
                k = int(rng.integers(0, len(seq)))
                seq = seq[:k] + "N" * min(5, len(seq) - k) + seq[k + 5:]
            recs.append(("r%d_%d" % (c, i), c, pos, ("%dS" % pre if pre else "") + lead + cg, seq))
    tmp = tempfile.mkdtemp()
    bam = os.path.join(tmp, "f.bam")
    bamio._bgzf_block = block
    try:
        bamio.write_bam(bam, refs, recs, block_size=bs, qual_seed=(int(rng.integers(1, 1 << 30)) if rng.random() < 0.5 else None))
    finally:
        bamio._bgzf_block = orig_block
    regions = []
    for _ in range(int(rng.integers(20, 80))):
        c = int(rng.integers(0, n_contigs))
        a = int(rng.integers(2, refs[c][1]))
        f = int(rng.choice([20, 100, 300, 500, 1000]))
        regions.append((refs[c][0], max(a - f, 1), a + int(rng.integers(1, 6000)) + f, f))
    status, n = T.compare(eng, bam, regions, max_keep=int(rng.choice([1, 5, 20, 60])))
    files += 1; regions_n += len(regions); reads_n += n; host_route += int((status != 0).sum())
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
print("fuzz_bamdev seed %d: %d files, %d regions (%d left to the host route), %d kept reads compared with the host reader's - numbers and bit planes all equal"
      % (seed, files, regions_n, host_route, reads_n), flush=True)
