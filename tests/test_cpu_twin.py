"""libvapor_cpu.so (oracle/cpu_twin.cpp): the C ABI of include/vapor_hip.h on the CPU oracle - SURVEY.md 8b's "same
symbols exported by a CPU build".  TEST INFRASTRUCTURE: bound here explicitly, never by vapor_amd.  It lets the
CPU-only suite run the real ctypes bindings and the real Engine / Plan / pipeline objects (on the GPU box the same
code runs on libvapor_hip.so), and it is itself pinned by the reference's vectors: the bodies of the GPU parity
tests are run against it."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def twin(oracle):
    """vapor_amd._lib bound to the CPU twin for the duration of this module."""
    from vapor_amd import _lib
    so = oracle.build_twin()
    saved = _lib._lib
    _lib._lib = _lib.bind(ctypes.CDLL(so))
    yield so
    _lib._lib = saved


@pytest.fixture()
def eng(twin):
    from vapor_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_twin_exports_the_whole_header(twin):
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vapor_hip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(vapor_[a-z_0-9]+)\s*\(", src)))
    lib = ctypes.CDLL(twin)
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libvapor_cpu.so does not export %s" % n


def test_gpu_parity_bodies_on_the_twin(eng, oracle):
    """The reference-vector checks of tests/test_gpu_parity.py, executed on the twin through the same Engine class."""
    import test_gpu_parity as G
    G.test_dotdata_golden(eng)
    G.test_cleaners_golden(eng, oracle)
    G.test_scorers_golden_end_to_end(eng)
    G.test_scorer_inputs_vs_oracle(eng, oracle)
    G.test_queue_edge(eng, oracle)
    G.test_deep_loci_device_finish_vs_reference(eng)
    G.test_device_finish_matches_host_finish(eng, "tiny")


def test_derived_sequences_on_the_twin(eng, oracle):
    """The bodies of tests/test_gpu_derived.py on the twin, which builds a derived sequence the way the reference does
    (slices, reverse(complementary()), str.upper()) and restates pack_kernel's symbol codes: the descriptors of the test
    cases say what their texts say, plane for plane; and the dot plots against them are the oracle's."""
    import test_gpu_derived as D
    D.check_planes(eng)
    D.test_revcomp_of_a_window_with_iupac_codes_is_refused(eng)
    assert D.check_shared_joins(eng, oracle, ks=(10, 30), want_shared=False) == 0      # (the twin joins pair by pair)
    assert D.check_random_structures(eng, oracle, n_windows=3) == 0


def test_pipeline_and_cli_on_the_twin(eng, tmp_path):
    """`vapor bed` through pipeline.run_batch with the real Engine on the twin: the reference's table."""
    from conftest import load_golden
    from vapor_amd import cli, pipeline, seqio, synth
    case = [c for c in load_golden("locus_bed.json.gz")["cases"] if c["name"] == "bed_hom_alt"][0]
    pipeline.set_engine(eng)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        bed = tmp_path / "in.bed"
        bed.write_text(case["bed"])
        out = tmp_path / "out.vapor"
        assert cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                         "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
        assert out.read_text() == case["vapor_text"]
    finally:
        pipeline.set_engine(None)
        seqio.set_backend(None)


def test_complex_types_on_the_twin(eng, tmp_path):
    import complex_cases as cx
    from vapor_amd import pipeline
    pipeline.set_engine(eng)
    try:
        cx.check_records(cx.CX["cases"][1], tmp_path)
        cx.check_disdup_driver()
    finally:
        pipeline.set_engine(None)


def test_error_codes_and_async_surface(eng):
    from vapor_amd import _lib as L
    ss = eng.seqset(["ACGTACGTACGTTTGACCA", "ACGTACGTACGTXACGT", "ACGT"])
    assert ss.n_invalid.tolist() == [0, 1, 0]
    st = eng.score(ss, eng.make_pairs([(1, 0, 0, 10, 7), (0, 0, 0, 11, 7), (0, 7, 0, 10, 7), (2, 0, 0, 10, 7)]))
    assert st[:, 15].tolist() == [L.E_KEYERROR, L.E_ARG, L.E_ARG, 0] and st[3, 0] == 0
    plan = eng.plan(ss, eng.make_pairs([(0, 0, 0, 10, 3)]))
    t = np.zeros(1, dtype=L.READ_DTYPE)
    t["kind"], t["len_ref"], t["len_alt"] = 1, 19, 19
    plan.set_reads(t, 1)
    a = plan.run_loci().copy()
    plan.run_loci_async()
    plan.then(0); plan.after(0)
    b = plan.sync().copy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    plan.close(); ss.close()


def test_timed_workloads_body_on_the_twin(eng, oracle):
    """The body of tests/test_gpu_workloads.py (the cfg2 / cfg3 batches bench.py times, against the oracle's scorers and
    finish) on the twin, at a few loci of each shape: the twin's float64 finish and the oracle-side expectation agree."""
    import test_gpu_workloads as W
    for name, n in (("cfg2", 4), ("cfg3", 5)):
        w = W.workload(name, n_loci=n)
        n_pairs, n_loci, _routes = W.check_workload(eng, oracle, name, w, routes=(1,), want_shared=False)
        assert n_loci == n and n_pairs == len(w.pairs)


def test_allele_with_more_blocks_than_a_descriptor_holds(eng):
    """a cannot_classify / SVelter alt allele of many blocks (the reference takes any number, SF:1490-1556) gives
    more segments than VAPOR_MAX_SEGMENTS; the library refuses such a descriptor, so pipeline.describe() must send that allele
    as bytes - same scores as the same text without segments - and _cat() merges slices that lie end to end."""
    from vapor_amd import _lib as L
    from vapor_amd import drivers, pipeline, synth
    rng = np.random.default_rng(77)
    ref = synth.random_dna(rng, 6000)
    blocks = [(400 + 300 * t, 400 + 300 * (t + 1)) for t in range(17)]
    order = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]        # no two neighbours lie end to end
    parts = [(ref, None, 400)] + [(ref, blocks[t][0], blocks[t][1], t % 3 == 0) for t in order] + [(ref, 400 + 300 * 17, None)]
    alt = drivers._cat(*parts)
    assert alt.segs is not None and len(alt.segs) == 19 > L.MAX_SEGMENTS
    plain = str(alt)
    merged = drivers._cat((ref, None, 400), (ref, 400, 700), (ref, 700, 1000), (ref, 1000, 1300, True), (ref, 1300, None))
    assert [(o, n, rc) for _p, o, n, rc in merged.segs] == [(0, 1000, False), (1000, 300, True), (1300, 4700, False)]
    reads = [[synth.mutate(rng, src[200:3200], 0.01, 0.05, 0.03)[0], 0, "r%d" % t] for t, src in enumerate((ref, plain, ref, plain))]
    got = pipeline.score_requests(eng, [drivers.Score("s1", ref, alt, reads, 10)])
    want = pipeline.score_requests(eng, [drivers.Score("s1", ref, plain, reads, 10)])
    assert got == want and any(v is not None for v in got[0])
    # sixteen segments still travel as a descriptor
    alt16 = drivers._cat(*(parts[:15] + [(ref, blocks[order[14]][0], None)]))
    assert len(alt16.segs) == 16
    assert pipeline.score_requests(eng, [drivers.Score("s1", ref, alt16, reads, 10)]) == pipeline.score_requests(eng, [drivers.Score("s1", ref, str(alt16), reads, 10)])


def test_reads_by_device_address_through_the_object_layer(eng, tmp_path):
    """vapor_bam_chop_device / vapor_seqset_create_mixed through Engine, InProcessBam.chop_many_device and the fast route, on the
    twin (its "device" addresses are host memory; every other read starts at an odd base): the kept reads and miss_bp of
    chop_many, the bit planes of the same reads uploaded as text, and the same table from the CLI as with VAPOR_BAM_DEVICE=0."""
    from vapor_amd import cli, pipeline, seqio, synth
    w = synth.make_world(seed=53, n_loci=14, svtypes=("DEL", "INV", "INS", "TANDUP", "DEL"), span_range=(80, 1500), read_len=5200, n_reads=24)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    be = seqio.InProcessBam()
    loci = w.loci
    chroms = [l.chrom for l in loci] + ["no_such_contig"]
    starts = np.asarray([l.start - 300 for l in loci] + [100], dtype=np.int64)
    ends = np.asarray([l.start + 300 for l in loci] + [700], dtype=np.int64)
    fl = np.full(len(chroms), 300, dtype=np.int64)
    kf, addr, q0, miss, status, keep = be.chop_many(bam, chroms, starts, ends, fl)
    dkf, daddr, dq0, dmiss, dstatus, batches = be.chop_many_device(eng, bam, chroms, starts, ends, fl)
    assert dstatus.tolist() == [0] * len(chroms) and dkf.tolist() == kf.tolist() and dmiss.tolist() == miss.tolist() and int(kf[-1]) > 100
    assert set(dq0.tolist()) == {0, 1}
    lens = np.concatenate([np.full(kf[g + 1] - kf[g], ends[g] - starts[g], dtype=np.int64) for g in range(len(chroms))]) - miss
    texts = [ctypes.string_at(int(addr[t] + q0[t]), int(lens[t])).decode() for t in range(len(addr))]
    dev = eng.seqset_raw(daddr, lens, None, keepalive=batches, src_kind=np.ones(len(daddr), dtype=np.uint8), src_first=dq0)
    ref = eng.seqset(texts)
    try:
        for t in range(len(texts)):
            assert all(np.array_equal(a, b) for a, b in zip(dev.planes(t), ref.planes(t))), t
        assert np.array_equal(dev.n_exc, ref.n_exc) and np.array_equal(dev.n_invalid, ref.n_invalid)
    finally:
        dev.close(); ref.close()
        for bt in batches:
            bt.close()
    # a source that is not a device source and one described badly
    with pytest.raises(Exception):
        eng.seqset_raw(daddr[:1], lens[:1], None, src_kind=np.asarray([2], dtype=np.uint8), src_first=np.zeros(1, dtype=np.int64))
    # the CLI both ways
    pipeline.set_engine(eng)
    seqio.set_backend(be)
    bed = tmp_path / "in.bed"
    bed.write_text(synth.bed_text(w))
    tables = {}
    try:
        for dev_on in ("1", "0"):
            out = tmp_path / ("out%s.vapor" % dev_on)
            os.environ["VAPOR_BAM_DEVICE"] = dev_on
            os.environ["VAPOR_QC_SEED"] = "7"
            assert cli.main(["bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam,
                             "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
            tables[dev_on] = out.read_text()
    finally:
        os.environ.pop("VAPOR_BAM_DEVICE", None)
        os.environ.pop("VAPOR_QC_SEED", None)
        pipeline.set_engine(None)
        seqio.set_backend(None)
    assert tables["1"] == tables["0"] and tables["1"].count("\n") == 15


def test_device_extraction_in_groups_and_a_region_too_large_for_a_call(eng, tmp_path, monkeypatch):
    """chop_many_device sends its regions in size-bounded groups and halves a group the library refuses for its size; a single
    region that is refused goes to the host route (status != 0), the others keep their answers."""
    from vapor_amd import _lib, seqio, synth
    w = synth.make_world(seed=54, n_loci=9, svtypes=("DEL", "INS"), span_range=(100, 900), read_len=3000, n_reads=22)
    for c in w.reads:
        w.reads[c] = sorted(w.reads[c], key=lambda r: r.pos)
    fa, bam = synth.write_world_files(w, str(tmp_path), block_size=0xFF00)
    be = seqio.InProcessBam()
    chroms = [l.chrom for l in w.loci]
    st = np.asarray([l.start - 200 for l in w.loci], dtype=np.int64)
    en = st + 700
    fl = np.full(len(chroms), 200, dtype=np.int64)
    want = be.chop_many_device(eng, bam, chroms, st, en, fl)
    for bt in want[5]:
        bt.close()
    real = eng.bam_chop_device
    calls = []

    def picky(native, tids, starts, *rest):
        calls.append(len(tids))
        if len(tids) > 2 or int(starts[0]) == int(st[4]):          # refuses groups of three and more, and locus 4 by itself
            raise _lib.VaporHipError(_lib.E_ARG, "vapor_bam_chop_device: more than 2 GB of block data in one call (use smaller batches)")
        return real(native, tids, starts, *rest)
    monkeypatch.setattr(eng, "bam_chop_device", picky)
    got = be.chop_many_device(eng, bam, chroms, st, en, fl)
    for bt in got[5]:
        bt.close()
    assert max(calls) == 9 and calls.count(1) >= 2
    status = got[4].tolist()
    assert status[4] != 0 and [s for t, s in enumerate(status) if t != 4] == [0] * 8
    n_want, n_got = np.diff(want[0]), np.diff(got[0])
    assert n_got[4] == 0 and np.array_equal(np.delete(n_got, 4), np.delete(n_want, 4))
    keep = np.ones(len(want[3]), dtype=bool)
    keep[int(want[0][4]):int(want[0][5])] = False
    assert np.array_equal(got[3], want[3][keep]) and np.array_equal(got[2], want[2][keep])
