"""Reference CPU path vs this repository's C port of it, on the same (read, allele) pairs of a bench shape
(cfg2: 10 kb reads x 20 kb windows, DEL/TANDUP; cfg3: 15 kb reads x 20 kb windows, DEL/DEL/TANDUP/INV/INS).
DEVELOPMENT CONTAINER ONLY: imports the reference from /root/reference (as oracle/gen_golden.py does) and, with
--cython, a scratch Cython build of it under /tmp (never inside the repository).
Writes profiles/r03_cpu_ratio_<cfg>.json, which bench.py quotes next to its own C-port timing (the reference cannot
travel to the GPU box).  usage: python tools/cpu_ratio.py [--cython] [--cfg cfg3] [n_reads]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import numpy as np

from vapor_amd import workload as wl
from oracle import oracle as orc
_spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "oracle", "gen_golden.py"))
gen_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen_golden)

n_reads = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 6
CFG = sys.argv[sys.argv.index("--cfg") + 1] if "--cfg" in sys.argv else "cfg2"
TYPES = wl.WORKLOADS[CFG]["svtypes"]
w = wl.make_workload(CFG, seed=1000, **dict(wl.WORKLOADS[CFG], n_loci=len(TYPES)))
orc.build()
mods = {"python": gen_golden.load_reference()}
if "--cython" in sys.argv:
    scratch = "/tmp/vapor_ref_cython"
    os.makedirs(scratch, exist_ok=True)
    so = [f for f in os.listdir(scratch) if f.startswith("Simple_function") and f.endswith(".so")]
    if not so:
        import shutil
        shutil.copy("/root/reference/vapor_vali/Simple_function.pyx", os.path.join(scratch, "Simple_function.pyx"))
        subprocess.check_call([sys.executable, "-m", "cython", "-3", "Simple_function.pyx"], cwd=scratch)
        import sysconfig
        inc = sysconfig.get_paths()["include"]
        ext = sysconfig.get_config_var("EXT_SUFFIX")
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-I" + inc, "-I" + np.get_include(), "Simple_function.c", "-o", "Simple_function" + ext], cwd=scratch)
    sys.path.insert(0, scratch)
    os.environ.setdefault("MPLBACKEND", "Agg")
    import Simple_function as cy
    mods["cython"] = cy

# locus 0 is a DEL (abs_dis_m1b + within_10Perc_m1b per read, SF:1718-1726), locus 1 a TANDUP
# (directed_dis_m1b_redefine_diagnal, SF:1761), as the cfg2 batch scores them
rpl = wl.WORKLOADS[CFG]["reads_per_locus"]
n_reads = min(n_reads, rpl)
loci = []
for li in range(len(TYPES)):
    base = li * (2 + rpl)
    loci.append((w.svtypes[li], w.seqs[base], w.seqs[base + 1], [[w.seqs[base + 2 + r], 0, "r%d" % r] for r in range(n_reads)]))
out = {"shape": "%s: %d reads of %d bp per locus x ref/alt windows of ~%d bp, k = 10; one locus per entry of the shape's type "
                "cycle %s (DEL: abs_dis_m1b + within_10Perc_m1b per read; TANDUP: directed_dis_m1b_redefine_diagnal; INV / INS: "
                "abs_dis_m1b)" % (CFG, n_reads, len(loci[0][3][0][0]), len(loci[0][1]), "/".join(TYPES)),
       "host": "development container, 1 core", "reads_per_locus_of_the_shape": rpl}
res = {}
for name, m in mods.items():
    r = []
    per_locus = []
    for t, ref, alt, reads in loci:
        t0 = time.perf_counter()
        if t == "DEL":
            r.append([(m.calcu_vapor_single_read_score_abs_dis_m1b(ref, alt, x, 10), m.calcu_vapor_single_read_score_within_10Perc_m1b(ref, alt, x, 10)) for x in reads])
        elif t == "TANDUP":
            r.append([(m.calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal(ref, alt, x, 10),) for x in reads])
        else:
            r.append([(m.calcu_vapor_single_read_score_abs_dis_m1b(ref, alt, x, 10),) for x in reads])
        per_locus.append((t, (time.perf_counter() - t0) / n_reads))
        print(name, t, per_locus[-1][1], flush=True)
    res[name] = r
    out["reference_%s_s_per_read" % name] = [[t, v] for t, v in per_locus]
    mean = sum(v for _t, v in per_locus) / len(per_locus)          # the type cycle weights the types as the shape does
    out["reference_%s_loci_per_s" % name] = 1.0 / (mean * rpl)
# the port as bench.py times it: one statistics record (fill + C1/C2 clean + counts) per (read, window) pair
t0 = time.perf_counter()
for t, ref, alt, reads in loci:
    for x in reads:
        orc.pair_stats(10, x[0], ref)
        orc.pair_stats(10, x[0], alt)
dt = (time.perf_counter() - t0) / (len(loci) * n_reads)
out["port_s_per_read"] = dt
out["port_loci_per_s"] = 1.0 / (dt * rpl)
for name in mods:
    out["port_over_reference_%s" % name] = out["port_loci_per_s"] / out["reference_%s_loci_per_s" % name]
if "cython" in res:
    assert json.dumps(gen_golden.jsonable(res["python"])) == json.dumps(gen_golden.jsonable(res["cython"]))
out["note"] = ("reference = /root/reference/vapor_vali/Simple_function.pyx, imported as plain Python and (cython) compiled "
               "unchanged with Cython in a scratch directory; port = oracle/vapor_oracle.c through oracle.pair_stats, "
               "one record per (read, window) pair as in bench.py's cpu_baseline leg, gcc -O2, one thread; both per locus of "
               "%d reads, mean over the loci of the type cycle" % rpl)
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_cpu_ratio_%s.json" % CFG), "w"), indent=1)
print(json.dumps(out, indent=1))
