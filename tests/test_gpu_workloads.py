"""The configurations bench.py TIMES, as workloads, against the CPU oracle.

bench.py's headline is workload.make_workload("cfg2") with derived alt windows -> shared joins -> the records of every served
pair cut out of the shared dot plot by its clean workgroup (remap_in_clean = 1 picks that route for a plan of a few rounds);
its sub-record is "cfg3", whose plan takes the remap_kernel route.  Here the whole cfg2 batch (100 loci, 4 000 dot plots of
10 kb x 20 kb) and the first 100 loci of the cfg3 batch (DEL / DEL / TANDUP / INV / INS, 8 000 dot plots of 15 kb x 20 kb) go
through w.upload(eng) -> plan.run() / run_loci() / run_loci_async() under remap_in_clean 0, 1 and 2 and with two plans in
flight; every pair's integer statistics equal oracle.pair_stats', every read's score and every locus's QS / GS /
GT / GQ equal what the oracle's scorers, result_organize_ins and gt_estimate_log_likelihood give (tests/workload_oracle.py).

The body takes an engine: tests/test_cpu_twin.py runs it on the CPU twin at a reduced size."""
import time

import numpy as np
import pytest

import workload_oracle as wo

pytestmark = pytest.mark.gpu

_EXPECT = {}


@pytest.fixture(scope="module")
def eng():
    from vapor_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def workload(name, n_loci=None, seed=None):
    from vapor_amd import workload as wl
    spec = dict(wl.WORKLOADS[name])
    if n_loci is not None:
        spec["n_loci"] = n_loci       # (the generator draws locus after locus from one stream: a prefix of the full batch)
    # the seeds bench.py uses for these shapes (headline: 1000 + rank; sub-record: 3000)
    seed = seed if seed is not None else {"cfg2": 1000, "cfg3": 3000}.get(name, 5)
    w = wl.make_workload(name, seed=seed, **spec)
    w.__dict__["_made_from"] = (name, spec, seed)
    return w


def expected(oracle, name, w):
    key = (name, w.n_loci)
    if key not in _EXPECT:
        t0 = time.perf_counter()
        _EXPECT[key] = wo.expect_parallel(oracle, *w.__dict__["_made_from"], w)
        print("oracle: %s, %d loci, %d dot plots in %.1f s" % (name, w.n_loci, len(w.pairs), time.perf_counter() - t0), flush=True)
    return _EXPECT[key]


def check_workload(eng, oracle, name, w, routes=(0, 1, 2), want_shared=True, two_plans=True):
    from vapor_amd import workload as wl
    exp = expected(oracle, name, w)
    table = wl.read_table(w)
    seen_routes = {}
    for route in routes:
        eng.set_param("remap_in_clean", route)
        try:
            ss = w.upload(eng)
            plans = []
            for _ in range(2 if two_plans else 1):
                p = eng.plan(ss, w.pairs)
                p.set_reads(table, w.n_loci)
                plans.append(p)
            p = plans[0]
            st = p.run().copy()
            tm = p.timings()
            if want_shared:
                assert tm["shared_joins"] > 0 and tm["pairs_served_by_shared_joins"] == len(w.pairs), (name, route, tm)
                assert tm["remap_in_clean"] == {0: 0, 2: 1}.get(route, tm["remap_in_clean"])
                seen_routes[route] = tm["remap_in_clean"]
            loci = p.run_loci(want_scores=True).copy()
            sc = p.read_scores[:len(table)].copy()
            n_pairs, n_loci = wo.check_plan(w, st, sc, loci, exp, tag="%s route %d blocking" % (name, route))
            # the asynchronous steps bench.py times: two plans in flight, a few passes each; the last records of each
            for q in plans[1:]:
                q.run_loci(want_host=False)                 # (sizes the slots)
            for i in range(6):
                plans[i % len(plans)].run_loci_async()
            for q in plans:
                again = q.sync().copy()
                wo.check_plan(w, st, None, again, exp, tag="%s route %d async" % (name, route))
            # and the statistics a plan leaves after those steps are still the oracle's
            st2 = plans[-1].run().copy()
            wo.check_plan(w, st2, None, loci, exp, tag="%s route %d second plan" % (name, route))
            for q in plans:
                q.close()
            ss.close()
        finally:
            eng.set_param("remap_in_clean", 1)
    return n_pairs, n_loci, seen_routes


def test_cfg2_whole_batch_against_the_oracle(eng, oracle):
    """BASELINE configs[1] as bench.py runs it: 100 DEL / TANDUP loci x 20 reads of 10 kb x windows of 20 kb, all 4 000 pairs."""
    w = workload("cfg2")
    assert w.derived and len(w.pairs) == 4000
    n_pairs, n_loci, routes = check_workload(eng, oracle, "cfg2", w)
    # (the default takes the clean workgroups on a plan of a few rounds of them: the route the headline times)
    assert n_pairs == 4000 and n_loci == 100 and routes == {0: 0, 1: 1, 2: 1}
    rec = expected(oracle, "cfg2", w)[2]
    assert int(np.isfinite(rec[:, 0]).sum()) >= 95          # (the batch is scored, not gated away)


def test_cfg3_first_100_loci_against_the_oracle(eng, oracle):
    """BASELINE configs[2]'s shape and type mix: the first 100 loci of bench.py's cfg3 batch (40 reads of 15 kb per locus)."""
    w = workload("cfg3", n_loci=100)
    assert set(w.svtypes) == {"DEL", "TANDUP", "INV", "INS"} and len(w.pairs) == 8000
    n_pairs, n_loci, routes = check_workload(eng, oracle, "cfg3", w)
    # (the default takes remap_kernel on a plan of many rounds - as the full 80 000-pair batch of the sub-record does)
    assert n_pairs == 8000 and n_loci == 100 and routes == {0: 0, 1: 0, 2: 1}
    rec = expected(oracle, "cfg3", w)[2]
    assert int(np.isfinite(rec[:, 0]).sum()) >= 95
