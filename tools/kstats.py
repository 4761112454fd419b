#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace SQLite db: per-kernel count / total / avg / min / max."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else None
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
print("total kernel ms %.3f" % sum(r[2] for r in rows))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    if pat and pat not in r[0]: continue
    print("%-64s n=%6d total=%9.3f ms avg=%9.2f us min=%8.2f max=%9.2f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
