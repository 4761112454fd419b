#!/usr/bin/env python
"""A/B of the BGZF leg of the host reader, reader alone: the own byte-mode inflate (default) against zlib's (SQ_BGZF_ZLIB=1), alternating.
   python tools/bgzf_ab.py [pairs]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, HERE)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
import numpy as np
d = "/dev/shm/sq_reader_bench"; os.makedirs(d, exist_ok=True)
import reader_bench as rb
for m in (1, 2):
    r = np.random.default_rng(m); L = 100
    row = np.empty((N, 3 + L + 3 + L + 1), np.uint8); row[:, 0:3] = np.frombuffer(b"@r\n", np.uint8); row[:, 3:3 + L] = r.choice(np.frombuffer(b"ACGT", np.uint8), size=(N, L))
    row[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", np.uint8); row[:, 6 + L:6 + 2 * L] = r.choice(np.frombuffer(b"F:,#", np.uint8), size=(N, L), p=[0.86, 0.09, 0.04, 0.01]); row[:, -1] = 10
    rb.bgzf_write("%s/b_%d.fq.gz" % (d, m), row.tobytes())
code = ("import sys; sys.path.insert(0, %r); import reader_bench as rb; d = '/dev/shm/sq_reader_bench'\n"
        "for i in range(4):\n    n, dt = rb.drain(d + '/b_1.fq.gz', d + '/b_2.fq.gz', 1000000); print('%%.3f s %%.2f M pairs/s' %% (dt, n / dt / 1e6))\n") % HERE
for rnd in range(2):
    for z in ("0", "1"):
        env = dict(os.environ, SQ_BGZF_ZLIB=z, SQ_READER_DEVICE="0")
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split("\n")
        print("%s inflate: %s" % ("own " if z == "0" else "zlib", " | ".join(x for x in out if x)), flush=True)
