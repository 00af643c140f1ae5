"""Pins the CPU oracle (oracle/) to vectors produced by the reference itself
(oracle/gen_golden.py -> tests/golden/*.json.gz).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden


def _close(a, b):
    if isinstance(b, str):
        return a == b
    return a == pytest.approx(b, rel=0, abs=1e-12) or a == b


@pytest.mark.parametrize("case", load_golden("kmerhits.json.gz")["cases"], ids=lambda c: c["name"])
def test_dotdata(oracle, case):
    if "error" in case:
        with pytest.raises(KeyError):
            oracle.dotdata_array(case["k"], case["seq1"], case["seq2"])
        return
    h = oracle.dotdata_array(case["k"], case["seq1"], case["seq2"])
    assert len(h) == case["n_hits"]
    assert hashlib.sha256(np.ascontiguousarray(h, dtype=np.int32).tobytes()).hexdigest() == case["sha256"]
    if "hits" in case:
        assert [list(map(int, t)) for t in h] == case["hits"]


@pytest.mark.parametrize("case", load_golden("cleaners.json.gz")["cases"], ids=lambda c: c["name"])
def test_cleaners(oracle, case):
    hits = np.asarray(case["hits"], dtype=np.int32).reshape(-1, 2)
    k1 = oracle.clean_c1_flags(hits)
    assert [list(map(int, t)) for t in hits[k1 > 0]] == case["c1"]["ok"]
    k2 = oracle.clean_c2_flags(hits)
    # the reference lists C2 survivors cluster by cluster; compare as multisets
    got_d = sorted(map(tuple, hits[k2 == 1].tolist()))
    assert got_d == sorted(map(tuple, case["c2_diag"]["ok"]))
    got_a = sorted(map(tuple, hits[k2 == 2].tolist()))
    assert got_a == sorted(map(tuple, case.get("c2_anti_on_left", {"ok": []})["ok"]))
    if "count10" in case:
        assert oracle.eu_dis_dots_within_10perc([tuple(t) for t in hits[k2 > 0].tolist()]) == case["count10"]
    if "meanabs" in case:
        kept = [tuple(t) for t in hits[k1 > 0].tolist()]
        assert float(oracle.eu_dis_abs_calcu(kept)) == case["meanabs"]
        r4 = oracle.dis_to_diagnal_most_abundant_defined(kept)
        assert float(r4) == float(case["r4"]["ok"])
        assert float(oracle.eu_dis_dir_calcu([[d[0] + r4, d[1]] for d in kept])) == case["dir"]["ok"]


@pytest.mark.parametrize("case", load_golden("scorers.json.gz")["cases"], ids=lambda c: c["name"])
def test_scorers(oracle, case):
    x = [case["read"], case["miss"], case["name"]]
    for key, fn in (("s1", oracle.score_abs_dis_m1b), ("s2", oracle.score_within_10Perc_m1b),
                    ("s3", oracle.score_directed_dis_m1b_redefine_diagnal)):
        exp = case[key]
        if "error" in exp:
            with pytest.raises(KeyError):
                fn(case["ref"], case["alt"], x, case["k"])
        else:
            got = fn(case["ref"], case["alt"], x, case["k"])
            assert [float(v) for v in got] == [float(v) for v in exp["ok"]], key


@pytest.mark.parametrize("case", load_golden("window.json.gz")["cases"], ids=lambda c: c["name"])
def test_selfplot_counts(oracle, case):
    """Integer part of window_size_refine: hit / diagonal / lower-triangle counts of each
    self dot-plot the reference evaluated (k = 10, 20, ...)."""
    seq = "".join(ch for ch in case["seq"] if ch != "X")
    for step, (n, d, lo) in enumerate(case["qc_trace"]):
        h = oracle.dotdata_array(10 + 10 * step, seq, seq)
        assert oracle.qual_check_counts(h) == (n, d, lo)


@pytest.mark.parametrize("case", load_golden("genotype.json.gz")["cases"], ids=lambda c: str(len(c["scores"])))
def test_genotype(oracle, case):
    r = oracle.result_organize_ins(["key", case["scores"]])
    exp = case["organize"]["ok"]
    assert r[0] == exp[0] and r[3] == exp[3]
    if exp[1] == "NA":
        assert r[1:] == ["NA"] * 3
        return
    assert float(r[1]) == float(exp[1]) and float(r[2]) == float(exp[2])
    gt = oracle.gt_estimate_log_likelihood(r)
    assert gt[0] == case["gt"]["ok"][0]
    assert float(gt[1]) == float(case["gt"]["ok"][1])
