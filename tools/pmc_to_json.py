"""Folds rocprofv3 --pmc passes (tools/pmc_run.sh) into the two small JSON files bench.py quotes:
  profiles/<round>_<workload>_traffic.json  fabric-side bytes per launch (FETCH_SIZE doubled per the gfx950 correction
                                            of MI355X_MICROARCH.md §HBM, WRITE_SIZE as read; both in KB units)
  profiles/<round>_<workload>_util.json     what the SQ counters say limits each kernel
usage: pmc_to_json.py <round> <workload> <pmc dir> [<pmc dir> ...]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # (kernel_source_id: bench.py quotes these files only for the source they were measured on)
rnd, workload, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            name = "join_kernel" if "join_kernel" in k else "clean_kernel" if (k.endswith("clean_kernel") or "clean_kernel<" in k) else "remap_kernel" if "remap_kernel" in k else None
            if name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            name = "join_kernel" if "join_kernel" in k else "clean_kernel" if (k.endswith("clean_kernel") or "clean_kernel<" in k) else "remap_kernel" if "remap_kernel" in k else None
            if name:
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
traffic, util = {}, {}
for name, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    us = sorted(dur[name])[len(dur[name]) // 2] if dur[name] else None
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        traffic[name] = {"bytes": int(2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024), "fetch_kb_raw": m["FETCH_SIZE"],
                         "write_kb": m["WRITE_SIZE"], "note": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KB units"}
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        u = {"us_under_pmc": round(us, 1) if us else None,
             "wave_time": {"parked_on_waitcnt_or_barrier": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
                           "issue_stalled": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                           "issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)},
             "valu_wave_instructions": int(m.get("SQ_INSTS_VALU", m.get("SQ_ACTIVE_INST_VALU", 0))),
             "lds_wave_instructions": int(m.get("SQ_INSTS_LDS", 0)), "salu_wave_instructions": int(m.get("SQ_INSTS_SALU", 0))}
        # The vector unit of a SIMD takes one wave64 instruction per quad-cycle, or two from different waves when both are
        # of the simple class (SQ_ACTIVE_INST_VALU2 counts those quad-cycles; tools/micro/valu_ops.hip says which opcodes
        # pair).  Quad-cycles the vector ALUs were occupied = instructions - paired quad-cycles; a CU has four SIMDs, so
        # its capacity over SQ_BUSY_CU_CYCLES cycles is that many quad-cycle slots.
        if "SQ_ACTIVE_INST_VALU2" in m and m.get("SQ_BUSY_CU_CYCLES"):
            iv = m.get("SQ_INSTS_VALU", m.get("SQ_ACTIVE_INST_VALU", 0))
            u["valu_busy"] = round((iv - m["SQ_ACTIVE_INST_VALU2"]) / m["SQ_BUSY_CU_CYCLES"], 3)
            u["valu_instructions_issued_in_pairs"] = round(2 * m["SQ_ACTIVE_INST_VALU2"] / max(iv, 1), 3)
            u["cu_busy_cycles_per_cu"] = int(m["SQ_BUSY_CU_CYCLES"] / 256)
        if us:
            cyc = us * 1e-6 * 2.4e9                        # at the 2.4 GHz maximum clock: utilisations are lower bounds on idle time
            if "SQ_LDS_IDX_ACTIVE" in m:
                u["lds_array_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 3)
        if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
            u["lds_bank_conflict_share_of_lds_cycles"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 3)
        vb = u.get("valu_busy")
        u["limit"] = (("vector issue: the vector ALUs are occupied %.2f of the time a CU is busy" % vb if vb else "vector issue") +
                      (" (%.0f %% of the instructions share a quad-cycle with another wave's)" % (100 * u["valu_instructions_issued_in_pairs"]) if vb else "") +
                      ("; the rest is LDS round trips at 4 waves per SIMD (one 159 KB table per CU); LDS array under half busy, "
                       "fabric traffic 2 % of HBM peak" if name == "join_kernel" else
                       "; the rest is workgroup start-up and dependent global/LDS round trips"))
        util[name] = u
SRC = bench.kernel_source_id()
if traffic:
    traffic["source_id"] = SRC
    traffic["measured_with"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_run.sh: bench.py --plans 1 --passes-per-step 1"
if util:
    util["source_id"] = SRC
if traffic:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, workload)), "w"), indent=1)
if util:
    json.dump(util, open(os.path.join(ROOT, "profiles", "%s_%s_util.json" % (rnd, workload)), "w"), indent=1)
print(json.dumps({"traffic": traffic, "util": util}, indent=1))
