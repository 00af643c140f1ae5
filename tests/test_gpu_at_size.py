"""The at-size runner (tools/run_at_size.py: BASELINE configs[3] / configs[4] through the product CLI, tiled worlds) at a size
that takes seconds, and its checker (tests/at_size_check.py: the same base world through the CPU twin of the C ABI in a child
process, sampled rows compared text for text) - so that what produced profiles/r03_cfg4_at_size.json and r03_cfg5_at_size.json
is itself under test."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,loci,base", [("cfg4", 180, 60), ("cfg5", 60, 20)])
def test_at_size_runner_and_twin_check(cfg, loci, base, tmp_path):
    out = str(tmp_path / "at_size.json")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))   # (prepended: the driver's hook stays)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_at_size.py"), cfg, "--loci", str(loci), "--base", str(base), "--out", out],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["records"] == loci and rec["rows"] >= loci and rec["rc"] == 0 and len(rec["sample"]) >= min(loci, 400) - 1
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "at_size_check.py"), out], env=env, capture_output=True, text=True, timeout=600)
    assert c.returncode == 0 and " 0 differ" in c.stdout, (c.stdout[-1500:], c.stderr[-1500:])


def test_distinct_world_runner_and_twin_check(tmp_path):
    """The same with a world of DISTINCT loci (every tile of the base world mutated on its own and made when it is reached,
    synth.DistinctTilesWorld): the sampled rows equal the CPU twin's rows for the same records made again."""
    out = str(tmp_path / "at_size_distinct.json")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_at_size.py"), "cfg5", "--loci", "90", "--base", "15", "--distinct",
                        "--sample", "40", "--chunk", "32", "--lru", "64", "--out", out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["distinct"] and rec["distinct_loci"] == rec["records"] == 90 and len(rec["sample"]) == 40
    assert len({row.split("\t", 5)[-1] for _t, row in rec["sample"] if "\tNA" not in row}) > 30          # different loci, different scores
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "at_size_check.py"), out], env=env, capture_output=True, text=True, timeout=900)
    assert c.returncode == 0 and " 0 differ" in c.stdout, (c.stdout[-1500:], c.stderr[-1500:])


def test_at_size_runner_with_the_truth_sets_span_distribution(tmp_path):
    """VERDICT r04 missing 2: a world whose spans follow the reference's simulated truth sets (simulate/Structural_Variants_het:
    50 bp - 100 kb, 7-10 % of the deletions and inversions >= 10 kb - the drivers' junction-window branch at depth) through the
    product CLI on the GPU; every row equals the CPU twin's."""
    out = str(tmp_path / "at_size_sim.json")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_at_size.py"), "cfg5", "--loci", "120", "--base", "120", "--spans", "simulate",
                        "--all-rows", "--out", out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["span_dist"] == "simulate" and rec["records"] == 120 and rec["spans"]["max"] >= 20000 and rec["spans"]["frac_ge_10kb"] >= 0.03
    assert rec["rows_with_scores"] >= 60
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "at_size_check.py"), out], env=env, capture_output=True, text=True, timeout=900)
    assert c.returncode == 0 and " 0 differ" in c.stdout, (c.stdout[-1500:], c.stderr[-1500:])
