"""Prints the GPU timeline around the end-of-job tail (export -> EM preparation) from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last k_prep_cw launch = the timed EM's preparation
idx = max(i for i, n in enumerate(names) if "k_prep_cw" in n and i < len(names) - 1)
cands = [i for i, n in enumerate(names) if "k_prep_cw" in n]
print("k_prep_cw launches at rows", cands)
for target in cands:
    t0 = int(rows[target]["Start_Timestamp"])
    print("---- around row", target)
    for r in rows[max(0, target - 12): target + 30]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%10.3f ms  dur %9.3f us  q=%s  %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
