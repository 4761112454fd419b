"""CPU-only scale check of the decoy-aware index builder: ~7 M nt of transcripts + three 10 M-nt decoy
"chromosomes" (random sequence with embedded transcripts and a repeat family), cDBG invariants verified by the checker, then a
few hundred pairs mapped by the checker.  8 cores: build 18 s, check 3 min.  python tools/decoy_scale_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from salmon_amd import api, synth
import orc
rng = np.random.default_rng(11)
tx = synth.Txome(seed=21, n_genes=500, iso_per_gene=8, threads=8)
names = list(tx.names()); seqs = [s.decode() for s in tx.seqs()]
print("transcripts", len(seqs), sum(map(len, seqs)))
A = np.frombuffer(b"ACGT", np.uint8)
chroms = []
for c in range(3):
    g = A[rng.integers(0, 4, 10_000_000)].copy()
    # embed 300 transcripts (as exons with introns) + a repeat family
    rep = A[rng.integers(0, 4, 800)]
    for _ in range(400):
        p = int(rng.integers(0, len(g) - 1000)); g[p:p+800] = rep
    for j in rng.choice(len(seqs), 300, replace=False):
        s = np.frombuffer(seqs[j].encode(), np.uint8); p = int(rng.integers(0, len(g) - len(s) - 10)); g[p:p+len(s)] = s
    chroms.append(g.tobytes().decode())
t = time.time()
idx = api.SalmonIndex.build_mem(names + ["chr%d" % i for i in range(3)], seqs + chroms, threads=8, first_decoy=len(seqs), keep_duplicates=True)
print("build %.1f s  refs %d  kmers %d" % (time.time() - t, idx.num_refs, idx.num_kmers if hasattr(idx, 'num_kmers') else -1))
t = time.time(); rc = orc.check_cdbg(idx); print("check_cdbg rc", rc, "%.1f s" % (time.time() - t))
# map a few reads from a chromosome and from transcripts with the checker
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
recs = []
def pair(s):
    fl = int(rng.integers(180, 320)); p = int(rng.integers(0, len(s) - fl))
    r1 = s[p:p + 100]; r2 = "".join(comp[c] for c in reversed(s[p + fl - 100:p + fl])); return (r1, r2)
for _ in range(300): recs += pair(chroms[int(rng.integers(3))])
long_seqs = [s for s in seqs if len(s) > 400]
for _ in range(300): recs += pair(long_seqs[int(rng.integers(len(long_seqs)))])
seq = np.frombuffer("".join(recs).encode(), np.uint8).copy(); off = np.arange(0, len(recs) + 1, dtype=np.uint64) * np.uint64(100)
oidx = orc.OrcIndex(idx); opts = api.quant_opts(); rb = api.make_read_batch(seq, off, 600, paired=True)
ro, aln, mt, st = orc.map_batch(oidx, opts, rb, threads=8); print(st)
