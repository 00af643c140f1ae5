"""The read extraction on the device (vapor_bam_chop_device: bgzf_inflate_kernel, bam_chop_kernel; vapor_seqset_create_mixed:
bam_expand_kernel) against the host's (vapor_bam_chop, itself pinned on the Python statement and on an independent encoder in
tests/test_bamio.py): kept reads, miss_bp and the bases themselves - as the bit planes of a set made from the device addresses
against those of the same reads uploaded as text - on files with blocks of 64 KB and of 1.5 KB (records over many blocks),
stored and fixed-code blocks, a 70 000-operation CIGAR in CG:B,I, aux fields, several references, regions without reads and on
contigs the file lacks, more candidates than minimize_pacbio_read_list keeps, more kept reads than a region's slot holds, and
damaged blocks (the region goes to the host route, its neighbours do not); then the CLI's table both ways."""
import ctypes
import os
import shutil
import struct
import zlib

import numpy as np
import pytest

import test_bamio as TB
from vapor_amd import _lib as L
from vapor_amd import bamio, cli, pipeline, seqio, synth
from vapor_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def compare(eng, bam, regions, max_keep=20):
    """regions: (chrom, start, end, flank).  Returns the device's status per region; for every region it answered, the kept
    reads are the host's (numbers and bases)."""
    be = seqio.InProcessBam()
    b = be._open(bam)
    chroms = [r[0] for r in regions]
    st = np.asarray([r[1] for r in regions], dtype=np.int64)
    en = np.asarray([r[2] for r in regions], dtype=np.int64)
    fl = np.asarray([r[3] for r in regions], dtype=np.int64)
    dkf, daddr, dq0, dmiss, dstatus, batches = be.chop_many_device(eng, bam, chroms, st, en, fl, max_keep)
    texts, lens, sel = [], [], []
    try:
        for g in range(len(regions)):
            if dstatus[g]:
                continue
            r = b.chop_native_raw(chroms[g], int(st[g]), int(en[g]), int(fl[g]))
            a, e = int(dkf[g]), int(dkf[g + 1])
            if r is None:
                assert e == a, g
                continue
            whole, off, ln, miss = r
            order = np.arange(len(off))
            if len(order) > max_keep:
                order = np.argsort(miss, kind="stable")[:max_keep]
            assert e - a == len(order) and dmiss[a:e].tolist() == miss[order].tolist(), (g, regions[g])
            for t, i in enumerate(order):
                assert int(ln[i]) == int(en[g] - st[g] - miss[i])
                texts.append(whole[int(off[i]):int(off[i]) + int(ln[i])])
                lens.append(int(ln[i]))
                sel.append(a + t)
        if texts:
            sel = np.asarray(sel)
            dev = eng.seqset_raw(daddr[sel], np.asarray(lens, dtype=np.int64), None, src_kind=np.ones(len(sel), dtype=np.uint8), src_first=dq0[sel])
            ref = eng.seqset(texts)
            try:
                for t in range(len(texts)):
                    assert all(np.array_equal(x, y) for x, y in zip(dev.planes(t), ref.planes(t))), t
                assert np.array_equal(dev.n_exc, ref.n_exc) and np.array_equal(dev.n_invalid, ref.n_invalid)
            finally:
                dev.close()
                ref.close()
    finally:
        for bt in batches:
            bt.close()
        b.close()
    return dstatus, len(texts)


def test_world_files_large_and_small_blocks(eng, tmp_path):
    w = synth.make_world(seed=61, n_loci=40, svtypes=("DEL", "INV", "INS", "TANDUP"), span_range=(80, 3000), read_len=7000, n_reads=26)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    for block in (0xFF00, 1500):
        d = tmp_path / ("b%d" % block)
        d.mkdir()
        fa, bam = synth.write_world_files(w, str(d), block_size=block)
        regions = [(l.chrom, max(l.start - 400, 1), l.start + 900, 400) for l in w.loci] + [("no_such_contig", 5, 900, 100), (w.loci[0].chrom, 1, 40, 10)]
        status, n = compare(eng, bam, regions)
        assert status.tolist() == [0] * len(regions) and n > 300, (block, status.tolist(), n)


def test_independent_encoder_long_cigar_aux_fields_several_references(eng, tmp_path):
    rng = np.random.default_rng(31)
    refs = [("chrA", 200000), ("chrB", 90000)]
    reads = []
    for i in range(40):
        tid = int(rng.integers(0, 2)); pos = int(rng.integers(0, 60000))
        ops, seq_len = [], 0
        for _ in range(int(rng.integers(1, 30))):
            o = "MIDS=X"[int(rng.integers(0, 6))]; n = int(rng.integers(1, 400))
            ops.append((n, o)); seq_len += n if o in "MIS=X" else 0
        if seq_len == 0:
            ops.append((5, "M")); seq_len = 5
        seq = "".join("ACGTN"[j] for j in rng.integers(0, 5, seq_len))
        reads.append(("q%d" % i, tid, pos, ops, seq, b"NMC\x05RGZgrp1\0" if i % 2 else b""))
    n_ops = 70000
    reads.append(("qlong", 0, 1000, [(1, "M") if j % 2 == 0 else (1, "I") for j in range(n_ops)], "ACGT" * (n_ops // 4), b"NMC\x01"))
    reads.append(("qun", -1, -1, [], "ACGT", b""))
    reads.sort(key=lambda r: (r[1] if r[1] >= 0 else 1 << 30, r[2]))
    p = str(tmp_path / "ind.bam")
    TB._encode_bam(p, refs, reads)
    regions = [("chrA", 1001, 1400, 100), ("chrA", 20000, 26000, 500), ("chrB", 5000, 9000, 500), ("chrA", 1, 100000, 500),
               ("chrB", 30000, 30100, 40), ("chrA", 1050, 1100, 20), ("chrZ", 1, 10, 5), ("chrA", 1100, 30000, 300)]
    status, n = compare(eng, p, regions)
    assert status.tolist() == [0] * len(regions) and n >= 3


def test_more_candidates_than_are_kept_and_more_kept_than_a_slot_holds(eng, tmp_path):
    rng = np.random.default_rng(8)
    contig = synth.random_dna(rng, 60000)
    # 90 reads over one window with different starts (different miss_bp): the 20 smallest, in file order inside one value
    recs = []
    for i in range(90):
        pos = 4000 + int(rng.integers(0, 900))
        pre = int(rng.integers(0, 40))
        read, cg = synth.mutate(rng, contig[pos:pos + 6000])
        recs.append(("m%d" % i, 0, pos, ("%dS" % pre if pre else "") + "%dD" % int(rng.integers(1, 700)) + cg, synth.random_dna(rng, pre) + read))
    p = str(tmp_path / "many.bam")
    bamio.write_bam(p, [("c", 60000)], recs, block_size=0xFF00)
    status, n = compare(eng, p, [("c", 4900, 6100, 1000), ("c", 5200, 5900, 1400)])
    assert status.tolist() == [0, 0] and n == 40
    # 300 kept reads in one region: beyond the 256 a region's slot holds - the host route's
    big = [("b%d" % i, 0, 100 + i, "30000M", "ACGT" * 7500) for i in range(300)]
    p3 = str(tmp_path / "big.bam")
    bamio.write_bam(p3, [("c", 60000)], big)
    status, n = compare(eng, p3, [("c", 500, 25000, 500), ("c", 150, 900, 30)])
    assert status[0] == 4 and status[1] == 0 and n == 20


@pytest.mark.parametrize("kind", ["stored", "fixed", "huffman_only", "level1"])
def test_blocks_of_every_deflate_kind(eng, tmp_path, kind, monkeypatch):
    def block(data):
        level, strat = {"stored": (0, zlib.Z_DEFAULT_STRATEGY), "fixed": (6, zlib.Z_FIXED), "huffman_only": (6, zlib.Z_HUFFMAN_ONLY),
                        "level1": (1, zlib.Z_DEFAULT_STRATEGY)}[kind]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
        cdata = comp.compress(data) + comp.flush()
        if len(cdata) + 26 > 65536:                        # (a stored block of 64 KB does not fit BSIZE: smaller pieces)
            raise OverflowError
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25)
                + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    monkeypatch.setattr(bamio, "_bgzf_block", block)
    w = synth.make_world(seed=62, n_loci=10, svtypes=("DEL", "INS"), span_range=(100, 1500), read_len=5000, n_reads=22)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=30000)
    monkeypatch.undo()
    status, n = compare(eng, bam, [(l.chrom, max(l.start - 300, 1), l.start + 700, 300) for l in w.loci])
    assert status.tolist() == [0] * len(w.loci) and n > 60


@pytest.mark.parametrize("field", ["crc", "payload_bit", "isize_small"])
def test_a_damaged_block_sends_its_regions_to_the_host_route_and_no_other(eng, tmp_path, field):
    rng = np.random.default_rng(5)
    contig = synth.random_dna(rng, 400000)
    recs = []
    for i in range(160):
        pos = 2000 * i + int(rng.integers(0, 500))
        read, cg = synth.mutate(rng, contig[pos:pos + 5000])
        recs.append(("m%d" % i, 0, pos, cg, read))
    good = str(tmp_path / "good.bam")
    bamio.write_bam(good, [("c", 400000)], recs, block_size=20000)
    raw = bytearray(open(good, "rb").read())
    bl = TB._blocks(bytes(raw))
    off, bsize, xlen = bl[len(bl) // 2]
    if field == "crc":
        raw[off + bsize - 8] ^= 0x40
    elif field == "payload_bit":
        raw[off + 12 + xlen + (bsize - xlen - 20) // 2] ^= 0x04
    else:
        struct.pack_into("<I", raw, off + bsize - 4, 17)
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    shutil.copy(good + ".bai", bad + ".bai")
    regions = [("c", 2000 * i + 600, 2000 * i + 1500, 300) for i in range(4, 150, 3)]
    be = seqio.InProcessBam()
    st = np.asarray([r[1] for r in regions]); en = np.asarray([r[2] for r in regions]); fl = np.asarray([r[3] for r in regions])
    dkf, daddr, dq0, dmiss, dstatus, batches = be.chop_many_device(eng, bad, ["c"] * len(regions), st, en, fl)
    for bt in batches:
        bt.close()
    b = be._open(bad)
    n_bad = 0
    for g, r in enumerate(regions):
        try:
            host = b.chop_native_raw(*r)
            host_n = 0 if host is None else len(host[1])
            assert dstatus[g] == 0 and int(dkf[g + 1] - dkf[g]) == min(host_n, 20), (g, dstatus[g])
        except ValueError:
            n_bad += 1
            assert dstatus[g] != 0, g                      # (what the host refuses, the device has not answered)
    assert 1 <= n_bad <= 12 and (dstatus != 0).sum() == n_bad


def test_addresses_outside_a_live_batch_are_refused(eng, tmp_path):
    w = synth.make_world(seed=63, n_loci=6, svtypes=("DEL",), span_range=(100, 900), read_len=3000, n_reads=22)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    be = seqio.InProcessBam()
    l = w.loci[0]
    kf, addr, q0, miss, status, batches = be.chop_many_device(eng, bam, [l.chrom], [l.start - 200], [l.start + 500], [200])
    assert int(kf[-1]) > 0
    lens = np.full(len(addr), 100, dtype=np.int64)
    one = np.ones(len(addr), dtype=np.uint8)
    ss = eng.seqset_raw(addr, lens, None, src_kind=one, src_first=q0)
    ss.close()
    with pytest.raises(L.VaporHipError):                   # beyond the arena's end
        eng.seqset_raw(addr + np.uint64(1 << 33), lens, None, src_kind=one, src_first=q0)
    for bt in batches:
        bt.close()
    with pytest.raises(L.VaporHipError):                   # the batch is gone
        eng.seqset_raw(addr, lens, None, src_kind=one, src_first=q0)


def test_cli_table_is_the_same_with_reads_by_device_address(eng, tmp_path):
    w = synth.make_world(seed=64, n_loci=60, svtypes=("DEL", "INV", "INS", "TANDUP", "DEL"), span_range=(80, 2500), read_len=6000, n_reads=24)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    bed = tmp_path / "in.bed"
    bed.write_text(synth.bed_text(w))
    seqio.set_backend(seqio.InProcessBam())
    tables = {}
    try:
        for dev in ("1", "0"):
            out = tmp_path / ("out%s.vapor" % dev)
            os.environ["VAPOR_BAM_DEVICE"] = dev
            os.environ["VAPOR_QC_SEED"] = "7"
            assert cli.main(["bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam, "--output-path", str(tmp_path / "figs"),
                             "--output-file", str(out), "--no-figures"]) == 0
            tables[dev] = out.read_text()
    finally:
        os.environ.pop("VAPOR_BAM_DEVICE", None)
        os.environ.pop("VAPOR_QC_SEED", None)
        seqio.set_backend(None)
    assert tables["1"] == tables["0"] and tables["1"].count("\n") == 61 and tables["1"].count("\tNA") < 10


def test_fuzz_of_the_device_extraction_for_a_few_seconds(tmp_path):
    """tools/fuzz_bamdev.py (random files of every block size, zlib level and strategy, clips / insertions / deletions / N,
    seeded and absent qualities; random regions): a short run here (VAPOR_FUZZ_SECONDS for a long one; eight minutes of it -
    2 086 files, 385 k kept reads - are logged in profiles/r05_fuzz_bamdev.txt)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bamdev.py"), os.environ.get("VAPOR_FUZZ_SECONDS", "12"), "5"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all equal" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])


def test_extraction_on_a_stream_masked_to_a_share_of_the_cus(eng, tmp_path):
    """`bam_cu_share` (what cli.py sets while several chunks are scored at once): the same kept reads from a stream masked to five,
    one and seven eighths of the CUs as from the context's own; the parameter's range; vapor_bam_last_stats says what the call did."""
    w = synth.make_world(seed=65, n_loci=24, svtypes=("DEL", "INS"), span_range=(100, 1500), read_len=6000, n_reads=22)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    regions = [(l.chrom, max(l.start - 300, 1), l.start + 700, 300) for l in w.loci]
    try:
        for share in (0, 5, 1, 7, 8):
            eng.set_param("bam_cu_share", share)
            status, n = compare(eng, bam, regions)
            assert status.tolist() == [0] * len(regions) and n > 200, share
            st = eng.bam_last_stats()
            assert st["regions"] == len(regions) and st["blocks"] > 20 and st["inflated_bytes"] > st["compressed_bytes"] > 0 and st["inflate_ms"] > 0
        for bad in (-1, 9):
            with pytest.raises(L.VaporHipError):
                eng.set_param("bam_cu_share", bad)
    finally:
        eng.set_param("bam_cu_share", 0)


def test_three_threads_extract_at_once_each_on_its_context(tmp_path):
    """What cli.py does with three chunks in flight: three threads, a context each, vapor_bam_chop_device of different region sets
    of one file at the same time (one of them on a masked stream) - every thread's kept reads are the host reader's."""
    import threading
    w = synth.make_world(seed=66, n_loci=45, svtypes=("DEL", "INS", "INV"), span_range=(100, 2000), read_len=6000, n_reads=22)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    regions = [(l.chrom, max(l.start - 300, 1), l.start + 800, 300) for l in w.loci]
    shares = [regions[k::3] for k in range(3)]
    out, errs = [None] * 3, []

    def work(k):
        try:
            e = Engine(0)
            try:
                if k == 1:
                    e.set_param("bam_cu_share", 5)
                for _ in range(3):
                    out[k] = compare(e, bam, shares[k])
            finally:
                e.close()
        except BaseException as ex:       # noqa: BLE001
            errs.append(ex)
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for k in range(3):
        status, n = out[k]
        assert status.tolist() == [0] * len(shares[k]) and n > 150
