"""Figure parity, shared by the CPU (fake engine) and GPU tests: what vapor_amd.figures would hand
to matplotlib against what the reference handed to it (tests/golden/figures.json.gz: matplotlib.pyplot replaced by
a recorder inside the reference, oracle/gen_golden.py gen_figures)."""
import hashlib

import numpy as np

from conftest import load_golden

FIG = load_golden("figures.json.gz")


def check_specs():
    from vapor_amd import drivers, figures
    n_drawn = 0
    for c in FIG["cases"]:
        req = drivers.Figure(c["scores"], c["best_read"], c["k"], c["ref_seq"], c["alt_seq"], c["name"])
        spec = figures.figure_spec(req)
        exp = c["drawn"]
        if exp is None:
            assert spec is None, c["name"]
            continue
        n_drawn += 1
        assert spec is not None and spec["name"] == exp["saved"], c["name"]
        assert len(spec["subplots"]) == len(exp["subplots"]) == 4
        for got, e in zip(spec["subplots"], exp["subplots"]):
            assert got["pos"] == e["pos"] and got["title"] == e["title"]
            h = got["hits"]
            assert len(h) == e["n"] and h[0].tolist() == e["first"] and h[-1].tolist() == e["last"]
            assert hashlib.sha256(np.ascontiguousarray(h, dtype=np.int32).tobytes()).hexdigest() == e["xy_sha256"]
            assert [float(t) for t in got["xticks"]] == e["xticks"] and got["xticklabels"] == e["xticklabels"]
    assert n_drawn >= 6
    long = [c for c in FIG["cases"] if len(c["name"].split("/")[-1]) > 150][0]
    assert len(long["drawn"]["saved"].split("/")[-1]) == 144 and long["drawn"]["saved"].endswith(".png")


def check_driver_requests():
    """The Figure requests this repository's drivers emit for the fixture's BED (best read, window size, ref and alt
    window, scores, file name as vapor_vali/vapor:338 builds it) equal the arguments of the reference's calls."""
    import os
    import tempfile
    from vapor_amd import cli, pipeline, seqio, synth
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(FIG["world"])))
    got = []
    try:
        tmp = tempfile.mkdtemp()
        bed = os.path.join(tmp, "in.bed")
        open(bed, "w").write(FIG["bed"])
        jobs = cli.bed_jobs(cli.bed_info_readin(bed, tmp), 3, "x.bam", "ref.fa", tmp + "/figs/", "s")
        pipeline.run_batch([j.make() for j in jobs], figure_fn=got.append)
    finally:
        seqio.set_backend(None)
    exp = [c for c in FIG["cases"] if c["plt_li"] != 99]
    assert len(got) == len(exp) >= 6
    # the lockstep executor answers loci in rounds, not in input order: pair the requests up by their windows
    by_win = {(g.ref_seq, g.alt_seq): g for g in got}
    assert len(by_win) == len(got)
    for e in exp:
        g = by_win[(e["ref_seq"], e["alt_seq"])]
        assert [float(s) for s in g.scores] == e["scores"] and g.k == e["k"]
        assert g.ref_seq == e["ref_seq"] and g.alt_seq == e["alt_seq"]
        assert (list(g.best_read) if g.best_read != "" else "") == e["best_read"]
