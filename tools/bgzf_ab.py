#!/usr/bin/env python
"""A/B of the BGZF leg of the host reader: one inflate state per worker thread (default) against one per member (SQ_BGZF_TLS=0), alternating, reader alone.
   python tools/bgzf_ab.py [pairs]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
code = ("import sys; sys.path.insert(0, %r); import reader_bench as rb; d = '/dev/shm/sq_reader_bench'\n"
        "for i in range(3):\n    n, dt = rb.drain(d + '/b_1.fq.gz', d + '/b_2.fq.gz', 1000000); print('%%.3f s %%.2f M pairs/s' %% (dt, n / dt / 1e6))\n") % os.path.dirname(os.path.abspath(__file__))
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "reader_bench.py"), str(N)], stdout=subprocess.DEVNULL)   # writes the files
for rnd in range(3):
    for tls in ("1", "0"):
        env = dict(os.environ, SQ_BGZF_TLS=tls, SQ_READER_DEVICE="0")
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split("\n")
        print("state per %s: %s" % ("thread" if tls == "1" else "member", " | ".join(x for x in out if x)), flush=True)
