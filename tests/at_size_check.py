"""TEST INFRASTRUCTURE (not collected by pytest): checks the rows sampled by tools/run_at_size.py against the CPU twin.

tools/run_at_size.py runs BASELINE.json configs[3] / configs[4] at their stated sizes through the product CLI on the GPU and
keeps 400 of the output rows.  This script builds the same base world from the recorded seed, runs its loci through the SAME
CLI with the CPU twin of the C ABI behind it (oracle/_build/libvapor_cpu.so: the oracle; VAPOR_HIP_LIB), and compares every
sampled row with the twin's row for the locus it is a tile of - text for text (QS, GS, GT, GQ and the per-read scores).

usage: python tests/at_size_check.py gpurun_out/r3_cfg5_at_size.json"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def twin_rows(rec):
    """The base world's rows from the CPU twin (a child process: the library is chosen at load time)."""
    from oracle import oracle as orc
    orc.build()
    twin = orc.build_twin()
    out = tempfile.mktemp(suffix=".json")
    # (PYTHONPATH prepended, not replaced: the driver's hook that records which native libraries a process loads rides on it)
    env = dict(os.environ, VAPOR_HIP_LIB=twin, VAPOR_ALLOW_TWIN="1", VAPOR_QC_SEED=str(rec["qc_seed"]),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), VAPOR_HOST_PROCS="0")
    n = rec["base_loci"] if rec["config"] == "cfg5" else rec["base_loci"]
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_at_size.py"), rec["config"], "--loci", str(n), "--base", str(rec["base_loci"]),
           "--out", out, "--all-rows"] + (["--spans", rec["span_dist"]] if rec.get("span_dist", "uniform") != "uniform" else [])
    subprocess.check_call(cmd, env=env, stdout=subprocess.DEVNULL)
    return json.load(open(out))


def twin_rows_distinct(rec, procs=4):
    """A distinct world (tools/run_at_size.py --distinct): the sampled records alone - the same seeded tiles made again - through
    the CLI on the CPU twin, in a few child processes; returns the rows in the sample's order."""
    from oracle import oracle as orc
    orc.build()
    twin = orc.build_twin()
    env = dict(os.environ, VAPOR_HIP_LIB=twin, VAPOR_ALLOW_TWIN="1", VAPOR_QC_SEED=str(rec["qc_seed"]),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), VAPOR_HOST_PROCS="0", VAPOR_CHUNKS_IN_FLIGHT="1")
    idx = [t for t, _row in rec["sample"]]
    shares = [idx[k::procs] for k in range(procs)]
    jobs = []
    for k, sh in enumerate(shares):
        if not sh:
            continue
        only = tempfile.mktemp(suffix=".only.json")
        json.dump(sh, open(only, "w"))
        out = tempfile.mktemp(suffix=".json")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "run_at_size.py"), rec["config"], "--loci", str(rec["records"]), "--base", str(rec["base_loci"]),
               "--distinct", "--only", only, "--out", out, "--all-rows", "--chunk", "256"] + (
                   ["--spans", rec["span_dist"]] if rec.get("span_dist", "uniform") != "uniform" else [])
        jobs.append((sh, out, subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL)))
    rows = {}
    for sh, out, p in jobs:
        assert p.wait() == 0
        got = json.load(open(out))["sample"]
        assert len(got) == len(sh), (len(got), len(sh))
        for t, (_q, row) in zip(sh, got):
            rows[t] = row
    return rows


def main():
    rec = json.load(open(sys.argv[1]))
    if rec.get("distinct"):
        tw = twin_rows_distinct(rec)
        bad = 0
        for t, row in rec["sample"]:
            if tw.get(t) != row:
                bad += 1
                if bad <= 5:
                    print("row %d differs:\n  gpu : %s\n  twin: %s" % (t, row[:300], (tw.get(t) or "<missing>")[:300]))
        scored = sum(1 for _t, r in rec["sample"] if "\tNA" not in r)
        print("%s: %d sampled rows of %d distinct loci (%d with scores) against the CPU twin's rows of the same records: %d differ"
              % (rec["config"], len(rec["sample"]), rec["rows"], scored, bad))
        return 1 if bad else 0
    tw = twin_rows(rec)
    strip = re.compile(r"\.t\d+")
    by_key = {}
    for _t, row in tw["sample"]:
        by_key[strip.sub("", row.split("\tVaPoR", 1)[0]).split("\t")[0:5].__str__()] = strip.sub("", row)
    bad = 0
    for t, row in rec["sample"]:
        want = by_key.get(strip.sub("", row).split("\t")[0:5].__str__())
        if want is None or want != strip.sub("", row):
            bad += 1
            if bad <= 5:
                print("row %d differs:\n  gpu : %s\n  twin: %s" % (t, strip.sub("", row)[:300], (want or "<no such locus>")[:300]))
    scored = sum(1 for _t, r in rec["sample"] if "\tNA" not in r)
    print("%s: %d sampled rows of %d (%d with scores) against the CPU twin's %d base rows: %d differ"
          % (rec["config"], len(rec["sample"]), rec["rows"], scored, len(tw["sample"]), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
