#!/usr/bin/env python3
"""End-to-end rate of the stand-alone driver from FASTQ files (host read pipeline + H2D + mapping lanes + EM):
generates a synthetic transcriptome and N read pairs, writes plain and gzip FASTQ, runs `salmon-hip index|quant`."""
import gzip, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from salmon_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/cli_e2e"
os.makedirs(out, exist_ok=True)
tx = synth.Txome(seed=1, n_genes=4000, iso_per_gene=10, threads=32)
with open(out + "/t.fa", "w") as f:
    for n, s in zip(tx.names(), tx.seqs()): f.write(">%s\n%s\n" % (n, s.decode()))
seq, off, _, _ = tx.reads(N, read_len=100, seed=2, threads=32, truth=False)
arr = seq.reshape(N, 2, 100)
q = b"I" * 100
for m in (0, 1):
    rows = [b"@r%d/%d\n" % (i, m + 1) + arr[i, m].tobytes() + b"\n+\n" + q + b"\n" for i in range(N)]
    blob = b"".join(rows)
    open(out + "/r_%d.fq" % (m + 1), "wb").write(blob)
    with gzip.open(out + "/r_%d.fq.gz" % (m + 1), "wb", compresslevel=4) as g: g.write(blob)
exe = os.path.join(ROOT, "salmon_amd", "bin", "salmon-hip")
subprocess.check_call([exe, "index", "-t", out + "/t.fa", "-i", out + "/idx", "-p", "32"], stderr=subprocess.DEVNULL)
for tag, ext in (("plain", ".fq"), ("gzip", ".fq.gz")):
    for lanes in ("1", "2"):
        t0 = time.time()
        subprocess.check_call([exe, "quant", "-i", out + "/idx", "-l", "IU", "-1", out + "/r_1" + ext, "-2", out + "/r_2" + ext, "-o", out + "/q_" + tag, "--lanes", lanes], stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        print("%s FASTQ, %s lane(s): %d pairs in %.2f s wall (incl. index load, EM) = %.2f M pairs/s" % (tag, lanes, N, dt, N / dt / 1e6), flush=True)
