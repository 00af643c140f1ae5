"""Residency / staging sweep of the clean kernel on developer builds (GPU box): every variant x (VAPOR_DEV_HCAP,
VAPOR_DEV_CLEAN_PAD) in a fresh process.  usage: python tools/clean_sweep.py cfg2 basedev:1416:0 basedev:1416:4000 dualdev:1100:0 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--child" in sys.argv:
    import numpy as np
    from vapor_amd import workload as wl
    from vapor_amd.engine import Engine
    name = sys.argv[sys.argv.index("--child") + 1]
    w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
    eng = Engine(0)
    plan = eng.plan(eng.seqset(w.seqs), w.pairs)
    for _ in range(4):
        plan.run()
    tj, tc = [], []
    for _ in range(20):
        plan.run()
        t = plan.timings()
        tj.append(t["join_ms"]); tc.append(t["clean_ms"])
    st = plan.run()
    rc = plan.record_counts()
    hc = int(os.environ.get("VAPOR_DEV_HCAP", "0"))
    print("join %.4f clean %.4f ms  checksum %d  records p50 %d p90 %d p99 %d max %d  above hcap %d"
          % (np.median(tj), np.median(tc), int(st.sum()), *np.percentile(rc, [50, 90, 99]).astype(int), rc.max(),
             int((rc > hc).sum()) if hc else -1), flush=True)
    sys.exit(0)

workload = sys.argv[1]
for spec in sys.argv[2:]:
    name, hcap, pad = spec.split(":")
    env = dict(os.environ, VAPOR_HIP_LIB=os.path.join(ROOT, "tools", "libvapor_ab_%s.so" % name))
    if int(hcap):
        env["VAPOR_DEV_HCAP"] = hcap
    if int(pad):
        env["VAPOR_DEV_CLEAN_PAD"] = pad
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", workload], env=env, capture_output=True, text=True)
    print("%-22s %s %s" % (spec, out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else ""), flush=True)
