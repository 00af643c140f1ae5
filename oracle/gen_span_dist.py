"""Build container only: reads the reference's simulated truth sets (/root/reference/simulate/Structural_Variants_het: the BED files
of simple SVs, the VCF files of complex ones) and writes what SURVEY.md section 8d asks the synthetic worlds to follow - the SV
type mix and the distribution of spans per type - as DATA: counts and quantile tables, no text of any reference file.

    python oracle/gen_span_dist.py   ->   vapor_amd/data/simulate_spans.json

vapor_amd.synth.make_world(span_dist="simulate") samples it (inverse CDF over the quantile table, linear between quantiles)."""
import collections
import glob
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/simulate/Structural_Variants_het"
N_Q = 200


def table(values):
    v = np.sort(np.asarray(values, dtype=np.int64))
    q = [int(round(float(x))) for x in np.quantile(v, np.linspace(0.0, 1.0, N_Q + 1))]
    return {"n": int(len(v)), "min": int(v[0]), "median": float(np.median(v)), "max": int(v[-1]),
            "frac_ge_10kb": round(float((v >= 10000).mean()), 4), "quantiles": q}


def main():
    spans = collections.defaultdict(list)
    ins_len = []
    n_files = 0
    for f in sorted(glob.glob(SRC + "/*.bed")):
        n_files += 1
        for ln in open(f):
            p = ln.split()
            if len(p) < 4:
                continue
            if p[3].startswith("INS"):
                m = re.search(r"_(\d+)$", p[3])           # INS:ALU_124 - the inserted element's length
                if m:
                    ins_len.append(int(m.group(1)))
                continue
            spans[p[3]].append(int(p[2]) - int(p[1]))
    cx = collections.defaultdict(list)
    for f in sorted(glob.glob(SRC + "/*.vcf")):
        n_files += 1
        for ln in open(f):
            if ln.startswith("#"):
                continue
            p = ln.rstrip("\n").split("\t")
            if len(p) < 8:
                continue
            t = re.search(r"SVTYPE=([A-Za-z_:]+)", p[7])
            e = re.search(r"(?:^|;)END=(\d+)", p[7])
            if t and e:
                cx[t.group(1)].append(int(e.group(1)) - int(p[1]))
    out = {"source": "simulate/Structural_Variants_het/*.bed and *.vcf of the reference (%d files): spans END - POS per SV type; "
                     "data only (counts and %d-step quantile tables)" % (n_files, N_Q),
           "simple": {t: table(v) for t, v in sorted(spans.items())},
           "insertion_length": table(ins_len),
           "complex": {t: table(v) for t, v in sorted(cx.items())}}
    allv = np.concatenate([np.asarray(v) for v in spans.values()])
    out["simple_all"] = {"n": int(len(allv)), "median": float(np.median(allv)), "frac_ge_10kb": round(float((allv >= 10000).mean()), 4)}
    path = os.path.join(ROOT, "vapor_amd", "data", "simulate_spans.json")
    json.dump(out, open(path, "w"), indent=None, separators=(",", ":"))
    print(path, {t: (x["n"], x["median"], x["frac_ge_10kb"]) for t, x in out["simple"].items()}, out["insertion_length"]["n"], out["simple_all"])


if __name__ == "__main__":
    main()
