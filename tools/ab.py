"""A/B timing of library variants on ONE box: each variant is this source tree compiled with extra -D flags.

  python tools/ab.py --build base: peel:-DVAPOR_AB_PEEL        (here; the .so files travel with gpurun)
  python tools/ab.py --run base peel [--rounds 3] [--workload cfg2]   (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def so(name):
    return os.path.join(ROOT, "tools", "libvapor_ab_%s.so" % name)


if "--build" in sys.argv:
    from vapor_amd import build as B
    for spec in sys.argv[sys.argv.index("--build") + 1:]:
        name, _, flags = spec.partition(":")
        cmd = [B.hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"] + B.EXTRA_FLAGS + (["-DVAPOR_DEV_BUILD"] if flags else []) + [
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "vapor_amd", "csrc"),
               "-Wno-unused-function", "-o", so(name)] + [f for f in flags.split(",") if f] + B.SOURCES + ["-lz"]
        subprocess.check_call(cmd)
        print("built", so(name))
    sys.exit(0)

if "--child" in sys.argv:
    import numpy as np
    from vapor_amd import workload as wl
    from vapor_amd.engine import Engine
    name = sys.argv[sys.argv.index("--child") + 1]
    w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
    eng = Engine(0)
    for kv in os.environ.get("VAPOR_AB_PARAMS", "").split(","):       # e.g. VAPOR_AB_PARAMS=remap_in_clean=0
        if kv:
            eng.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    plan = eng.plan(w.upload(eng), w.pairs)          # (derived alt windows: the plan shares its joins)
    for _ in range(5):
        plan.run()
    tj, tc = [], []
    for _ in range(30):
        plan.run()
        t = plan.timings()
        tj.append(t["join_ms"]); tc.append(t["clean_ms"])
    st = plan.run()
    print("join %.4f clean %.4f ms (median of 30)  checksum %d" % (np.median(tj), np.median(tc), int(st.sum())), flush=True)
    sys.exit(0)

names = [a for a in sys.argv[sys.argv.index("--run") + 1:] if not a.startswith("-")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cfg2"
names = [n for n in names if n not in (str(rounds), workload)]
for r in range(rounds):
    for n in names:
        # name@param=value,...  (engine parameters; an upper-case name is an environment variable of the child: a dev-build switch)
        kvs = [kv for kv in n.partition("@")[2].split(",") if kv]
        env = dict(os.environ, VAPOR_HIP_LIB=so(n.split("@")[0]), VAPOR_AB_PARAMS=",".join(kv for kv in kvs if not kv[0].isupper()))
        env.update(kv.split("=") for kv in kvs if kv[0].isupper())
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", workload], env=env, capture_output=True, text=True)
        print("%-34s %s %s" % (n, out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else ""), flush=True)
