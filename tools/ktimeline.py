#!/usr/bin/env python3
"""Kernel timeline around an anchor kernel from a rocprofv3 --kernel-trace SQLite db: python tools/ktimeline.py db anchor_substring [n_after] [occurrence]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); anchor = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 60; occ = int(sys.argv[4]) if len(sys.argv) > 4 else -1
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if not idx: sys.exit("anchor not found")
i0 = idx[occ]; t0 = rows[i0][1]
for r in rows[max(0, i0 - 3): i0 + n]:
    print("%10.3f ms  +%8.1f us  %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, r[0][:70]))
