#!/usr/bin/env python3
"""Find the longest hipMemcpyAsync in a rocprofv3 --hip-trace db and list what other threads / the GPU were doing meanwhile."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
print("views:", views)
cols = [r[1] for r in c.execute("pragma table_info(regions)")]
print("regions cols:", cols)
rows = c.execute("select name, tid, start, end from regions where name like 'hipMemcpyAsync%' order by (end-start) desc limit 3").fetchall()
for name, tid, s, e in rows:
    print("\nLONG %s tid=%s dur=%.3f ms" % (name, tid, (e - s) / 1e6))
    oth = c.execute("select name, tid, start, end from regions where end > ? and start < ? and not (tid = ? and start = ?) order by start", (s, e, tid, s)).fetchall()
    print("  overlapping API calls: %d" % len(oth))
    agg = {}
    for n, t, s2, e2 in oth:
        k = (n, t); a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (min(e, e2) - max(s, s2)) / 1e6
    for (n, t), (cnt, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:15]:
        print("   %-40s tid=%s n=%d overlap=%.3f ms" % (n, t, cnt, ms))
    ks = c.execute("select name, start, end, queue_id from kernels where end > ? and start < ? order by start", (s, e)).fetchall()
    print("  kernels running in the window: %d; busy %.3f ms" % (len(ks), sum((min(e, k[2]) - max(s, k[1])) for k in ks) / 1e6))
    for k in ks[:5] + ks[-5:]:
        print("    %-50s q=%s start=+%.3f ms dur=%.3f ms" % (k[0][:50], k[3], (k[1] - s) / 1e6, (k[2] - k[1]) / 1e6))
    # what did the same thread do right before
    prev = c.execute("select name, start, end from regions where tid = ? and end <= ? order by end desc limit 6", (tid, s)).fetchall()
    for n, s2, e2 in prev: print("   prev: %-40s end=-%.3f ms dur=%.3f ms" % (n, (s - e2) / 1e6, (e2 - s2) / 1e6))
