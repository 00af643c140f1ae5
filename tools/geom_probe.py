import sys
sys.path.insert(0, '.')
from vapor_amd import workload as wl
from vapor_amd.engine import Engine
eng = Engine(0)
for name in ("cfg2", "cfg3", "cfg1", "tiny"):
    if name not in wl.WORKLOADS: continue
    w = wl.make_workload(name, seed=1000, **wl.WORKLOADS[name])
    plan = eng.plan(w.upload(eng), w.pairs)
    plan.run()
    t = plan.timings()
    print(name, len(w.pairs), {k: t[k] for k in ("clean_workgroups_per_cu", "remap_in_clean", "shared_joins")}, flush=True)
