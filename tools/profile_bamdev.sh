#!/bin/bash
# usage: tools/profile_bamdev.sh   (GPU box, from the repo root) - evidence for the device read extraction on the final sources:
# rocprofv3 --kernel-trace --stats of the files path (tools/files_ab.py: device and host extraction, tables compared) and one
# counter pass over the extraction alone (tools/bamdev_probe.py); the summaries go to gpurun_out/profiles_r05/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r05
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_files_stats -- python3 $R/tools/files_ab.py 1000 > $O/r05_files_path_device.txt 2> $R/gpurun_out/r05_files_stats.err ) || exit 2
F=$(find $R/gpurun_out/r05_files_stats -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $O/r05_files_kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/r05_bamdev_pmc -- python3 $R/tools/bamdev_probe.py 1000 > $O/r05_bamdev_probe.txt 2> $R/gpurun_out/r05_bamdev_pmc.err ) || exit 3
python3 - <<PY > $O/r05_bamdev_pmc.txt
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/r05_bamdev_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "bamdev" not in k and "pack_kernel" not in k: continue
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in rows.items():
    l = max(n[k], 1)
    print(k, "launches", n[k], {a: round(b / l / 1e6, 3) for a, b in c.items()}, "(millions per launch)")
    if c.get("SQ_BUSY_CU_CYCLES"):
        print("   valu_busy", round((c["SQ_INSTS_VALU"] - c.get("SQ_ACTIVE_INST_VALU2", 0)) / c["SQ_BUSY_CU_CYCLES"], 3), " salu per busy CU cycle", round(c["SQ_INSTS_SALU"] / c["SQ_BUSY_CU_CYCLES"], 3))
PY
rm -rf $R/gpurun_out/r05_files_stats $R/gpurun_out/r05_bamdev_pmc
find $R/gpurun_out -name "*kernel_trace.csv" -size +1M -delete
echo "bamdev profile done"
