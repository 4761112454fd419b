#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace db: how fast does the eq-stage stream progress under each mapping kernel?"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
packs = [i for i, r in enumerate(rows) if 'k_pack' in r[0]]
a, b = packs[4], packs[8]
mainq = rows[a][3]
q2 = [r for r in rows[a:b] if r[3] == mainq]
q3 = [r for r in rows[a:b] if r[3] != mainq]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('sqk::', '')
    return n[:n.index('(')][:28] if '(' in n else n[:28]
# time covered by each main kernel; eq kernels (by name) that START during it with durations
cover = collections.defaultdict(float); eqn = collections.defaultdict(lambda: collections.defaultdict(list))
idle_eq = collections.defaultdict(list)
import bisect
starts = [r[1] for r in q2]
for r in q2: cover[short(r[0])] += (r[2] - r[1]) / 1e3
for e in q3:
    i = bisect.bisect_right(starts, e[1]) - 1
    if i >= 0 and q2[i][2] > e[1]: eqn[short(q2[i][0])][short(e[0])].append((e[2] - e[1]) / 1e3)
    else: idle_eq[short(e[0])].append((e[2] - e[1]) / 1e3)
print("window %.1f ms, main busy %.1f ms" % ((rows[b][1] - rows[a][1]) / 1e6, sum(cover.values()) / 1e3))
for k, t in sorted(cover.items(), key=lambda x: -x[1]):
    d = eqn.get(k, {})
    s = ", ".join("%s n=%d avg=%.0fus" % (n, len(v), sum(v) / len(v)) for n, v in d.items() if len(v) >= 3)
    print("%-28s %8.1f ms | %s" % (k, t / 1e3, s))
print("main idle | " + ", ".join("%s n=%d avg=%.0fus" % (n, len(v), sum(v) / len(v)) for n, v in idle_eq.items() if len(v) >= 3))
