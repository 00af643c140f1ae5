"""rocprofv3 --kernel-trace --stats summary -> profiles/<round>_<workload>_kernel_stats.json: per kernel the calls, the average
duration and the share of GPU time, stamped with the kernel source id (bench.py's roofline.kernel is the largest share of the
summary measured on its own source).  usage: stats_to_json.py <round> <workload> <dir with *kernel_stats.csv> [csv copy target]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

rnd, workload, d = sys.argv[1:4]
files = glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv")
assert files, "no kernel_stats.csv under " + d
kern = {}
for r in csv.DictReader(open(files[0])):
    k = r["Name"]
    name = ("join_kernel" if "join_kernel" in k else "clean_big_kernel" if "clean_big_kernel" in k else "clean_kernel" if "clean_kernel" in k
            else "remap_kernel" if "remap_kernel" in k else "finish_kernel" if "finish_kernel" in k else None)
    if name is None:
        continue
    e = kern.setdefault(name, {"calls": 0, "total_ns": 0.0, "percentage": 0.0})
    e["calls"] += int(r["Calls"]); e["total_ns"] += float(r["TotalDurationNs"]); e["percentage"] += float(r["Percentage"])
for e in kern.values():
    e["avg_us"] = round(e["total_ns"] / max(e["calls"], 1) * 1e-3, 3)
    e["percentage"] = round(e["percentage"], 3)
    del e["total_ns"]
out = {"kernels": kern, "source_id": bench.kernel_source_id(),
       "measured_with": "rocprofv3 --kernel-trace --stats of bench.py (tools/profile_r05.sh), instantiations of a kernel summed"}
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.json" % (rnd, workload)), "w"), indent=1)
shutil.copy(files[0], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (rnd, workload)))
print(json.dumps(out))
