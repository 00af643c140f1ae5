"""A/B of library builds on the DEVICE-FINISHED path bench.py times (vapor_plan_run_loci_async, one and two plans in flight), one
box: every variant is a child process with VAPOR_HIP_LIB set (tools/ab.py --build makes tools/libvapor_ab_<name>.so).

  python tools/ab_loci.py base new [new@remap_in_clean=2] [--workloads cfg2,cfg3] [--rounds 2] [--seconds 1.0]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--child" in sys.argv:
    import numpy as np
    from vapor_amd import workload as wl
    from vapor_amd.engine import Engine
    name = sys.argv[sys.argv.index("--child") + 1]
    seconds = float(sys.argv[sys.argv.index("--child") + 2])
    w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
    eng = Engine(0)
    for kv in os.environ.get("VAPOR_AB_PARAMS", "").split(","):
        if kv:
            eng.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    ss = w.upload(eng)
    plans = []
    for _ in range(2):
        p = eng.plan(ss, w.pairs)
        p.set_reads(wl.read_table(w), w.n_loci)
        rec = p.run_loci().copy()
        plans.append(p)
    out = []
    for ps in (plans[:1], plans):
        for i in range(4 * len(ps)):
            ps[i % len(ps)].run_loci_async()
        for p in ps:
            p.sync(want_host=False)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for i in range(16):
                ps[(n + i) % len(ps)].run_loci_async()
            n += 16
        for p in ps:
            p.sync(want_host=False)
        dt = time.perf_counter() - t0
        tm = ps[0].timings()
        out.append("%d plan: %.4f ms/pass %8.0f loci/s (join %.4f clean %.4f)" % (len(ps), dt / n * 1e3, w.n_loci * n / dt, tm["join_ms"], tm["clean_ms"]))
    chk = plans[0].sync().copy()
    print(" | ".join(out) + " | checksum %.6f route %d" % (float(np.nansum(chk[:, :4])), plans[0].timings()["remap_in_clean"]), flush=True)
    sys.exit(0)

args = [a for a in sys.argv[1:]]
def opt(name, default):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return v
    return default
workloads = opt("--workloads", "cfg2,cfg3").split(",")
rounds = int(opt("--rounds", "2"))
seconds = opt("--seconds", "1.0")
for r in range(rounds):
    for wn in workloads:
        for n in args:
            kvs = [kv for kv in n.partition("@")[2].split(",") if kv]
            env = dict(os.environ, VAPOR_HIP_LIB=os.path.join(ROOT, "tools", "libvapor_ab_%s.so" % n.split("@")[0]),
                       VAPOR_AB_PARAMS=",".join(kv for kv in kvs if not kv[0].isupper()))
            env.update(kv.split("=") for kv in kvs if kv[0].isupper())
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wn, seconds], env=env, capture_output=True, text=True)
            print("%-5s %-28s %s %s" % (wn, n, out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else ""), flush=True)
