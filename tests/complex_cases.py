"""Shared by the CPU (fake engine) and GPU tests of BASELINE.json configs[3]'s complex SV types: replays the
reference-generated fixture tests/golden/locus_complex.json.gz (oracle/gen_golden.py gen_complex) through this
repository's drivers / CLI, whatever engine pipeline.get_engine() holds."""
import pytest

from conftest import load_golden

CX = load_golden("locus_complex.json.gz")


def _driver_for(x, y):
    from vapor_amd import drivers
    fn = {"DISDUP": drivers.vapor_simple_disdup, "DEL_INV": drivers.vapor_del_inv, "DUP_INV": drivers.vapor_dup_inv,
          "Other": drivers.vapor_cannot_classify}[x]
    return fn(3, 1, "x.bam", "ref.fa", y, "f.png")


def check_records(case, tmp_path):
    """Every record of the fixture's VCF through its one-locus driver: same scores (float for float) or the same
    exception type as the reference."""
    from vapor_amd import cli, pipeline, seqio, synth
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        vcf = tmp_path / "in.vcf"
        vcf.write_text(case["vcf"])
        vcf_list, _rec = cli.vcf_list_readin(str(vcf))
        it = iter(case["per_record"])
        n = 0
        for x in list(vcf_list.keys()):
            for y in vcf_list[x]:
                p = next(it)
                assert p["type"] == x and p["item"] == [list(i) if isinstance(i, list) else i for i in y]
                gen = _driver_for(x, y)
                if "error" in p["scores"]:
                    with pytest.raises(Exception) as ei:
                        pipeline.run_sync(gen)
                    assert type(ei.value).__name__ == p["scores"]["error"], p["key"]
                else:
                    got = pipeline.run_sync(gen)
                    assert [float(v) for v in got] == [float(v) for v in p["scores"]["ok"]], p["key"]
                n += 1
        assert n == len(case["per_record"])
    finally:
        seqio.set_backend(None)


def check_cli(case, tmp_path):
    """`vapor vcf` on a fixture whose records all score: the annotated table equals the reference's text."""
    from vapor_amd import cli, seqio, synth
    assert all("ok" in p["scores"] for p in case["per_record"])
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    try:
        vcf = tmp_path / "in.vcf"
        vcf.write_text(case["vcf"])
        assert cli.main(["vcf", "--sv-input", str(vcf), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                         "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"]) == 0
        assert (tmp_path / "in.vcf.vapor").read_text() == case["final"]
    finally:
        seqio.set_backend(None)


def check_disdup_driver():
    """vapor_simple_disdup_Vapor with integer coordinates (whole-region branch, SF:1795-1821)."""
    from vapor_amd import drivers, pipeline, seqio, synth
    d = CX["disdup_driver"]
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(d["world"])))
    try:
        for c in d["cases"]:
            gen = drivers.vapor_simple_disdup(3, 1, "x.bam", "ref.fa", list(c["sv_info"]), "f.png")
            if "error" in c["scores"]:
                with pytest.raises(Exception) as ei:
                    pipeline.run_sync(gen)
                assert type(ei.value).__name__ == c["scores"]["error"]
            else:
                assert [float(v) for v in pipeline.run_sync(gen)] == [float(v) for v in c["scores"]["ok"]], c["sv_info"]
    finally:
        seqio.set_backend(None)
