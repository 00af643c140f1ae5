"""Host float64 finishing (L2/L3) against reference-generated vectors.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from vapor_amd import finish


@pytest.mark.parametrize("case", load_golden("genotype.json.gz")["cases"], ids=lambda c: str(len(c["scores"])))
def test_result_organize_and_gt(case):
    r = finish.result_organize_ins(["key", case["scores"]])
    exp = case["organize"]["ok"]
    assert r[0] == exp[0] and r[3] == exp[3]
    if exp[1] == "NA":
        assert r[1:] == ["NA"] * 3
        assert finish.locus_summary(case["scores"]) is None
        return
    assert float(r[1]) == float(exp[1]) and float(r[2]) == float(exp[2])
    gt = finish.gt_estimate_log_likelihood(r)
    assert gt[0] == case["gt"]["ok"][0]
    assert float(gt[1]) == float(case["gt"]["ok"][1])
    qs, gs, idx, gq = finish.locus_summary(case["scores"])
    assert float(qs) == float(exp[1]) and gs == float(exp[2])
    assert finish.gt_name(idx) == case["gt"]["ok"][0] and float(gq) == float(case["gt"]["ok"][1])


def test_rounded_nonpositive_matches_python_round():
    rng = np.random.default_rng(9)
    v = np.concatenate([rng.normal(0, 0.01, 20000), np.array([0.005, 0.0049999999999999, 0.00500000000000001,
                        -0.0, 0.0, 0.015, 0.004999999999999999, np.nextafter(0.005, 0), np.nextafter(0.005, 1)])])
    exp = np.array([not float(str(round(float(x), 2))) > 0 for x in v])
    assert (finish.rounded_nonpositive(v) == exp).all()


def test_genotype_table_equals_the_scalar_statement():
    """finish.gt_table (what vapor_plan_set_reads hands the device-side finish; built a row of l at a time) against
    _gt_from_counts - log_likelihood_calcu and the arg-max / quality of gt_estimate_log_likelihood (SF:2054-2077) term by
    term - for every (k, l) of the table: genotype index and quality equal bit for bit."""
    from vapor_amd import _lib as L
    t = finish._gt_table_rows(L.GT_TABLE_N)
    assert t.shape == (L.GT_TABLE_N, L.GT_TABLE_N, 2)
    n = 0
    with np.errstate(divide="ignore"):
        for k in range(1, L.GT_TABLE_N):
            for l in range(k + 1):
                idx, gq = finish._gt_from_counts(k, l)
                assert t[k, l, 0] == idx and (t[k, l, 1] == gq or (np.isnan(gq) and np.isnan(t[k, l, 1]))), (k, l)
                n += 1
            assert not t[k, k + 1:].any()
    assert n == (L.GT_TABLE_N - 1) * (L.GT_TABLE_N + 2) // 2
    assert finish.gt_table() is finish.gt_table()
