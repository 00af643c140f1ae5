"""Complex SV types (DISDUP, DUP_INV, DEL_INV, Other=) on the CPU: host logic against the reference's vectors,
device work answered by tests/fake_engine.py (oracle-backed, test only)."""
import pytest

import complex_cases as cx
from fake_engine import FakeEngine
from vapor_amd import pipeline, seqio


@pytest.fixture()
def fake(oracle):
    pipeline.set_engine(FakeEngine(oracle))
    yield
    pipeline.set_engine(None)
    seqio.set_backend(None)


@pytest.mark.parametrize("case", cx.CX["cases"], ids=lambda c: c["name"])
def test_complex_records(fake, case, tmp_path):
    cx.check_records(case, tmp_path)


def test_complex_vcf_cli(fake, tmp_path):
    cx.check_cli([c for c in cx.CX["cases"] if c["name"] == "vcf_cx_b"][0], tmp_path)


def test_disdup_driver_integer_coordinates(fake):
    cx.check_disdup_driver()
