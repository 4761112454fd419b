#!/usr/bin/env python3
"""Decoy-genome scale check of the index builder: a small synthetic transcriptome plus N random "chromosomes" (with copies of
transcripts embedded in the first one) written as a gentrome FASTA + decoys.txt, indexed by the stand-alone driver (peak resident memory from getrusage).
   python tools/genome_scale_check.py [total_Gnt=0.3] [chrom_Mnt=125] [workdir=/tmp/sq_genome] [repeat_copies_per_chrom=0]
With repeat copies: 2000 families of 300-nt elements, every copy with ~2 % substitutions, scattered over each chromosome (many short unitigs,
crowded minimizer buckets).
Prints build time, peak resident memory and the index statistics (info.json).  No GPU needed."""
import json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from salmon_amd import synth
total = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
chrom = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 125_000_000
wd = sys.argv[3] if len(sys.argv) > 3 else "/tmp/sq_genome"
nrep = int(sys.argv[4]) if len(sys.argv) > 4 else 0
os.makedirs(wd, exist_ok=True)
fa, dec = os.path.join(wd, "gentrome.fa"), os.path.join(wd, "decoys.txt")
rng = np.random.default_rng(5)
tx = synth.Txome(seed=21, n_genes=500, iso_per_gene=8, threads=8)
names = [n if isinstance(n, str) else n.decode() for n in tx.names()]; seqs = list(tx.seqs())
t0 = time.time(); nchr = max(1, int(round(total * 1e9 / chrom))); A = np.frombuffer(b"ACGT", np.uint8)
with open(fa, "wb") as f, open(dec, "w") as d:
    for n, s in zip(names, seqs): f.write(b">" + n.encode() + b"\n" + s + b"\n")
    for c in range(nchr):
        g = A[rng.integers(0, 4, chrom, dtype=np.uint8)]
        if nrep:
            if c == 0: fams = A[rng.integers(0, 4, (2000, 300), dtype=np.uint8)]
            pos = rng.integers(0, chrom - 300, nrep); fam = rng.integers(0, 2000, nrep)
            for q0 in range(0, nrep, 100000):   # vectorised in slabs
                pp = pos[q0:q0 + 100000]; el = fams[fam[q0:q0 + 100000]].copy()
                mut = rng.random(el.shape) < 0.02; el[mut] = A[rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)]
                g[(pp[:, None] + np.arange(300)[None, :]).ravel()] = el.ravel()
        if c == 0:
            for j in rng.choice(len(seqs), min(300, len(seqs)), replace=False):
                s = np.frombuffer(seqs[j], np.uint8); p = int(rng.integers(0, chrom - len(s) - 10)); g[p:p + len(s)] = s
        f.write(b">chr%d\n" % c); f.write(g.tobytes()); f.write(b"\n"); d.write("chr%d\n" % c)
print("wrote %s: %d transcripts + %d x %d nt decoys in %.0f s" % (fa, len(seqs), nchr, chrom, time.time() - t0), flush=True)
exe = os.path.join(ROOT, "salmon_amd", "bin", "salmon-hip"); out = os.path.join(wd, "idx")
t0 = time.time()
r = subprocess.run([exe, "index", "-t", fa, "-d", dec, "-i", out, "-p", str(os.cpu_count())], capture_output=True, text=True)
dt = time.time() - t0
print("exit", r.returncode, "build %.0f s" % dt, "peak resident %.1f GB" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6))
print("\n".join(l for l in r.stderr.splitlines() if "sq-timing" in l or "salmon-hip" in l))
if r.returncode == 0: print(open(os.path.join(out, "info.json")).read())
