#!/usr/bin/env python3
"""Experiment: the late first synchronisation of the EM set-up.  From a rocprofv3 --hip-trace --hsa-trace --kernel-trace --memory-copy-trace database: the longest
hipStreamSynchronize calls that have (almost) no device work inside, and what the process — every thread, HIP and HSA level — and the device did during them."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
def cols(t): return [r[1] for r in c.execute("pragma table_info(%s)" % t)]
print("objects:", [n for n in names if not n.startswith("rocpd_")][:40])
reg = "regions" if "regions" in names else None
print("regions cols:", cols(reg)); print("kernels cols:", cols("kernels"))
mc = "memory_copies" if "memory_copies" in names else None
if mc: print("memory_copies cols:", cols(mc))
rows = c.execute("select name, tid, start, end from regions where name = 'hipStreamSynchronize' order by (end-start) desc limit 12").fetchall()
for name, tid, s, e in rows:
    ks = c.execute("select name, start, end, queue_id from kernels where end > ? and start < ? order by start", (s, e)).fetchall()
    busy = sum((min(e, k[2]) - max(s, k[1])) for k in ks) / 1e6
    dur = (e - s) / 1e6
    if busy > 0.5 * dur: continue
    print("\n==== hipStreamSynchronize tid=%s dur=%.3f ms, kernels inside: %d busy %.3f ms" % (tid, dur, len(ks), busy))
    for k in ks[:12]: print("    kernel %-46s q=%s start=+%.3f ms dur=%.3f ms" % (k[0][:46], k[3], (k[1] - s) / 1e6, (k[2] - k[1]) / 1e6))
    if mc:
        cp = c.execute("select name, start, end from %s where end > ? and start < ? order by start" % mc, (s, e)).fetchall()
        for n, s2, e2 in cp[:12]: print("    copy   %-46s start=+%.3f ms dur=%.3f ms" % (str(n)[:46], (s2 - s) / 1e6, (e2 - s2) / 1e6))
    oth = c.execute("select name, tid, start, end from regions where end > ? and start < ? and not (tid = ? and start = ?) order by start", (s, e, tid, s)).fetchall()
    agg = {}
    for n, t, s2, e2 in oth:
        a = agg.setdefault((n, t), [0, 0.0, 1e18]); a[0] += 1; a[1] += (min(e, e2) - max(s, s2)) / 1e6; a[2] = min(a[2], (s2 - s) / 1e6)
    for (n, t), (cnt, ms, first) in sorted(agg.items(), key=lambda x: -x[1][1])[:25]:
        print("   api %-44s tid=%s n=%d overlap=%.3f ms first at +%.3f ms" % (n[:44], t, cnt, ms, first))
    prev = c.execute("select name, start, end from regions where tid = ? and end <= ? order by end desc limit 8", (tid, s)).fetchall()
    for n, s2, e2 in prev: print("   before: %-40s ended -%.3f ms dur=%.3f ms" % (n[:40], (s - e2) / 1e6, (e2 - s2) / 1e6))
    lastk = c.execute("select name, end from kernels where end <= ? order by end desc limit 3", (s,)).fetchall()
    for n, e2 in lastk: print("   last kernel before: %-40s ended -%.3f ms" % (n[:40], (s - e2) / 1e6))
