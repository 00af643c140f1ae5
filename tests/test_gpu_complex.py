"""BASELINE.json configs[3]: the complex SV types of the VCF path through the HIP library, against vectors the
reference produced (tests/golden/locus_complex.json.gz; X-means never entered, so they are reproducible)."""
import pytest

import complex_cases as cx

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def real_engine():
    from vapor_amd import pipeline
    pipeline.set_engine(None)          # the process-wide HIP engine
    yield
    pipeline.set_engine(None)


@pytest.mark.parametrize("case", cx.CX["cases"], ids=lambda c: c["name"])
def test_complex_records_gpu(case, tmp_path):
    cx.check_records(case, tmp_path)


def test_complex_vcf_cli_gpu(tmp_path):
    cx.check_cli([c for c in cx.CX["cases"] if c["name"] == "vcf_cx_b"][0], tmp_path)


def test_disdup_driver_integer_coordinates_gpu():
    cx.check_disdup_driver()
