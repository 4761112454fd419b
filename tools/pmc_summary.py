#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, launches and average counter value per launch."""
import csv, glob, sys, collections, json
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]; n = n.replace("(anonymous namespace)::", "").replace("sqk::", "")
        n = n[: n.index("(")] if "(" in n else n
        n = n.replace("void ", "").split("<")[0].strip()
        a = agg[n][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, cs in agg.items():
    out[k] = {c: {"launches": v[0], "avg_per_launch": v[1] / v[0], "total": v[1]} for c, v in cs.items()}
keys = sorted(out, key=lambda k: -max(x["total"] for x in out[k].values()))
for k in keys[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-40s %s" % (k[:40], "  ".join("%s: n=%d avg=%.1f" % (c, x["launches"], x["avg_per_launch"]) for c, x in out[k].items())))
if len(sys.argv) > 3: json.dump({k: out[k] for k in keys}, open(sys.argv[3], "w"), indent=1)
