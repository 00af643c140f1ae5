import csv, collections, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        for r in rows:
            agg[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if any(x in k[0].lower() for x in ("join", "clean", "finish", "remap")):
                print("%-30s %-28s n=%d avg=%.1f" % (k[0], k[1], len(v), sum(v) / len(v)))
