#!/usr/bin/env python3
"""Spot check of an index built by tools/genome_scale_check.py: k-mers sampled from the decoy chromosomes (and from transcripts) must be found by the
host look-up, the unitig occurrence it names must be listed in the contig table at the sampled reference position, and random k-mers must miss.
   python tools/genome_scale_verify.py [workdir=/tmp/sq_genome] [samples=20000]     (no GPU: sq_index_load with device -1)"""
import mmap, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from salmon_amd import api
wd = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sq_genome"; N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
t0 = time.time(); idx = api.SalmonIndex.load(os.path.join(wd, "idx"), device=-1); v = idx.view(); k = idx.k
print("loaded in %.0f s: %d refs (first decoy %d), %d unitigs, %d k-mers" % (time.time() - t0, idx.num_refs, idx.first_decoy, idx.num_unitigs, idx.num_kmers), flush=True)
names = idx.ref_names(); rid = {n: i for i, n in enumerate(names)}
uoff = np.ctypeslib.as_array(v.uoff, shape=(v.num_unitigs + 1,)); ctab_off = np.ctypeslib.as_array(v.ctab_off, shape=(v.num_unitigs + 1,))
ctab = np.ctypeslib.as_array(v.ctab, shape=(int(ctab_off[-1]),))
code = np.full(256, 255, np.uint8); code[[65, 67, 71, 84]] = [0, 1, 2, 3]
def pack(b):
    c = code[np.frombuffer(b, np.uint8)].astype(np.uint64); return int((c << (2 * np.arange(len(b), dtype=np.uint64))).sum())
rng = np.random.default_rng(9); found = placed = 0; tested = 0
with open(os.path.join(wd, "gentrome.fa"), "rb") as f:
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    chrs = [n for n in names if n.startswith("chr")]
    for name in rng.choice(chrs, min(len(chrs), 8), replace=False):
        h = mm.find(b">" + name.encode() + b"\n"); s0 = h + len(name) + 2; e = mm.find(b"\n", s0); L = e - s0
        for p in rng.integers(0, L - k, N // 8):
            km = pack(mm[s0 + int(p): s0 + int(p) + k]); r = idx.lookup_host(km); tested += 1
            if r is None: continue
            found += 1; u, off, fw = r
            occ = ctab[int(ctab_off[u]): int(ctab_off[u + 1])]
            ulen = int(uoff[u + 1] - uoff[u])
            for o in occ:       # occurrence = ref << 32 | fw << 31 | pos of the unitig on the reference
                ref, ofw, pos = int(o) >> 32, (int(o) >> 31) & 1, int(o) & 0x7FFFFFFF
                kpos = pos + off if ofw else pos + (ulen - k - off)
                if ref == rid[name] and kpos == int(p): placed += 1; break
miss = sum(idx.lookup_host(int(x)) is None for x in rng.integers(0, 1 << 62, 5000))
print("decoy k-mers: %d sampled, %d found, %d placed at the sampled position by the contig table; random k-mers: %d of 5000 miss" % (tested, found, placed, miss))
print("OK" if found == tested and placed == tested and miss >= 4990 else "FAILED")
