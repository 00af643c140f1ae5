"""`vapor bed` end to end on the GPU: table identical to the reference's, byte for byte."""
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LOCUS = load_golden("locus_bed.json.gz")["cases"] + load_golden("locus_long.json.gz")["cases"]     # (+ spans of 20-99 kb: the junction-window branches)


@pytest.mark.parametrize("case", [c for c in LOCUS if not any("error" in p["scores"] for p in c["per_locus"])],
                         ids=lambda c: c["name"])
def test_bed_cli_rows_gpu(case, tmp_path):
    from vapor_amd import cli, pipeline, seqio, synth
    pipeline.set_engine(None)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        bed = tmp_path / "in.bed"
        bed.write_text(case["bed"])
        out = tmp_path / "out.vapor"
        rc = cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                       "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"])
        assert rc == 0
        assert out.read_text() == case["vapor_text"]
    finally:
        seqio.set_backend(None)


def test_reference_named_functions_gpu():
    from vapor_amd import pipeline
    from vapor_vali.Simple_function import dotdata, window_size_refine, clean_dotdata_diagnal_and_anti_diagnal
    pipeline.set_engine(None)
    kat = "ACGTACGTACGTTTGACCA"
    assert dotdata(10, kat, kat) == [(0, 0), (0, 2), (1, 1), (1, 1), (2, 0), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6),
                                      (7, 7), (8, 8), (9, 9)]
    with pytest.raises(KeyError):
        dotdata(10, "ACGTACGTACGTXACGT", kat)
    assert window_size_refine("ACGTAC") == ["Error", "Error"]
    dots = [(t, t) for t in range(11)]
    assert clean_dotdata_diagnal_and_anti_diagnal(dots) == dots
    assert clean_dotdata_diagnal_and_anti_diagnal(dots[:10]) == []


def test_figures_render(tmp_path):
    from vapor_amd import drivers, figures, pipeline, synth
    import numpy as np
    pipeline.set_engine(None)
    rng = np.random.default_rng(4)
    ref = synth.random_dna(rng, 1500)
    alt = ref[:500] + ref[-500:]
    read, _ = synth.mutate(rng, alt)
    out = tmp_path / "x.DEL.c1__1__2__DEL.png"
    figures.make_event_figure_1(drivers.Figure([0.5], [read[:990], 0, "r"], 10, ref, alt, str(out)))
    assert out.exists() and out.stat().st_size > 1000


OTHER = load_golden("locus_other.json.gz")


@pytest.mark.parametrize("case", OTHER["cases"], ids=lambda c: c["name"])
def test_cannot_classify_driver_gpu(case):
    from vapor_amd import pipeline, seqio, synth
    from vapor_amd import simple_function as SF
    pipeline.set_engine(None)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(OTHER["world"])))
    try:
        got = SF.vapor_CANNOT_CLASSIFY_VapoR(3, 1, "x.bam", "ref.fa", list(case["sv_info"]), "f.png")
        assert [float(v) for v in got] == [float(v) for v in case["scores"]["ok"]]
    finally:
        seqio.set_backend(None)


VCF = load_golden("locus_vcf.json.gz")["cases"]


@pytest.mark.parametrize("name", ["vcf_simple_nohdr", "vcf_tiny_span_nohdr"])
def test_vcf_cli_gpu(name, tmp_path):
    from vapor_amd import cli, pipeline, seqio, synth
    case = [c for c in VCF if c["name"] == name][0]
    pipeline.set_engine(None)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        vcf = tmp_path / "in.vcf"
        vcf.write_text(case["vcf"])
        assert cli.main(["vcf", "--sv-input", str(vcf), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                         "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"]) == 0
        assert (tmp_path / "in.vcf.vapor").read_text() == case["final"]
    finally:
        seqio.set_backend(None)


def test_config1_vapor_test_bed_gpu(tmp_path):
    from vapor_amd import cli, pipeline, seqio, synth
    cfg = load_golden("config1_bed.json.gz")
    pipeline.set_engine(None)
    seqio.set_backend(seqio.MemorySamtools(synth.make_world_from_bed(cfg["bed_rows"], seed=cfg["seed"])))
    try:
        bed = tmp_path / "vapor_test.bed"
        bed.write_text(cfg["bed"])
        out = tmp_path / "vapor_test.bed.vapor"
        assert cli.main(["bed", "--sv-input", str(bed), "--reference", "hg19.fa", "--pacbio-input", "x.bam",
                         "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
        assert out.read_text() == cfg["cases"][0]["vapor_text"]
    finally:
        seqio.set_backend(None)


def test_melt_ins_mode_gpu(tmp_path):
    from vapor_amd import cli, pipeline, seqio, synth
    d = load_golden("melt_ins.json.gz")
    pipeline.set_engine(None)
    world = synth.world_from_json(d["world"])
    world.contigs.update(d["fasta"])
    seqio.set_backend(seqio.MemorySamtools(world))
    try:
        (tmp_path / "S1.melt.sites.vcf").write_text(d["vcf"])
        assert cli.main(["ins", "--sv-input", str(tmp_path / "S1.melt.sites"), "--reference", "ref.fa", "--pacbio-input",
                         "x.bam", "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"]) == 0
        assert (tmp_path / "S1.melt.sites.vapor").read_text() == d["cases"][0]["vapor_text"]
    finally:
        seqio.set_backend(None)


def test_figure_specs_match_what_the_reference_plots_gpu():
    """The four plotted point sets, tick lists, titles and the clamped file name of every make_event_figure_1 call of
    the reference (recorded with matplotlib.pyplot replaced inside it), from dot plots the HIP library computes."""
    import figure_cases
    from vapor_amd import pipeline
    pipeline.set_engine(None)
    figure_cases.check_specs()
    figure_cases.check_driver_requests()


@pytest.mark.parametrize("in_flight", ["1", "2", "3"])
def test_bed_cli_chunks_in_flight_gpu(in_flight, tmp_path, monkeypatch):
    """`vapor bed --chunk 3` on the eight-locus world: several chunks of a run scored at once (cli.score_jobs: a thread and a
    library context per chunk in flight, VAPOR_CHUNKS_IN_FLIGHT) give the reference's table byte for byte, whatever the
    number in flight."""
    from vapor_amd import cli, pipeline, seqio, synth
    case = [c for c in LOCUS if c["name"] == "bed_small_mix"][0]
    monkeypatch.setenv("VAPOR_CHUNKS_IN_FLIGHT", in_flight)
    pipeline.set_engine(None)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        bed = tmp_path / "in.bed"
        bed.write_text(case["bed"])
        out = tmp_path / "out.vapor"
        rc = cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam", "--chunk", "3",
                       "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"])
        assert rc == 0
        assert out.read_text() == case["vapor_text"]
        assert len(pipeline.get_engines(int(in_flight))) == int(in_flight)
    finally:
        seqio.set_backend(None)
        pipeline.set_engine(None)


def test_bed_cli_figures_with_chunks_in_flight_gpu(tmp_path, monkeypatch):
    """with figures on (the CLI's default) and more loci than --chunk, two chunks are in flight, each on its
    own thread and library context - the figure pass of a chunk must run on that chunk's context too (it used the default
    one: two threads inside one vapor_ctx).  The table and every PNG equal those of the one-chunk-at-a-time run."""
    import hashlib
    import os
    from vapor_amd import cli, figures, pipeline, seqio, synth
    case = [c for c in LOCUS if c["name"] == "bed_small_mix"][0]
    monkeypatch.setenv("VAPOR_HOST_PROCS", "2")
    got = {}
    for in_flight in ("1", "2"):
        monkeypatch.setenv("VAPOR_CHUNKS_IN_FLIGHT", in_flight)
        pipeline.set_engine(None)
        seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
        engines_used = []
        real = figures.figure_specs

        def spy(reqs, engine=None, _real=real, _seen=engines_used):
            _seen.append(engine)
            return _real(reqs, engine)
        monkeypatch.setattr(figures, "figure_specs", spy)
        try:
            d = tmp_path / ("run" + in_flight)
            d.mkdir()
            bed = d / "in.bed"
            bed.write_text(case["bed"])
            out = d / "out.vapor"
            rc = cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam", "--chunk", "3",
                           "--output-path", str(d / "figs"), "--output-file", str(out)])
            assert rc == 0
            assert out.read_text() == case["vapor_text"]
            pngs = {f: hashlib.sha256(open(os.path.join(d, "figs", f), "rb").read()).hexdigest()
                    for f in sorted(os.listdir(d / "figs")) if f.endswith(".png")}
            got[in_flight] = pngs
            assert engines_used and all(e is not None for e in engines_used)       # the chunk's engine, never the default by omission
            if in_flight == "2":
                assert len({id(e) for e in engines_used}) == 2                      # both contexts drew their own chunk's figures
        finally:
            monkeypatch.setattr(figures, "figure_specs", real)
            seqio.set_backend(None)
            pipeline.set_engine(None)
    assert got["1"] and got["1"] == got["2"]
