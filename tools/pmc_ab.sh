#!/bin/bash
# usage: tools/pmc_ab.sh <lib name (tools/libvapor_ab_<name>.so)> <tag> [workload]   (GPU box, repo root)
# SQ issue counters of one library variant on the blocking path (tools/ab.py --child), one rocprofv3 --pmc pass.
R=$GRAFT_REPO_ROOT
export VAPOR_HIP_LIB=$R/tools/libvapor_ab_$1.so
W=${3:-cfg2}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_$2 -- python3 $R/tools/ab.py --child $W > $R/gpurun_out/pmcab_$2.log 2>&1
echo "pmc_ab $2 rc=$?"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmcab_$2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        for n in ("join_kernel", "clean_kernel", "remap_kernel", "clean_big"):
            if n in k:
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print("$2", n, "launches", len(next(iter(c.values()))), {k: round(v / 1e6, 3) for k, v in m.items()},
          "valu_busy", round((m.get("SQ_INSTS_VALU", 0) - m.get("SQ_ACTIVE_INST_VALU2", 0)) / max(m.get("SQ_BUSY_CU_CYCLES", 1), 1), 3))
PY
