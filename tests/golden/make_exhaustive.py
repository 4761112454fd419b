"""Makes tests/golden/exhaustive_labels.npz: per-fragment transcript labels (and alignment scores) from the exhaustive all-positions aligner
(oracle/exhaustive.cpp — no index, no seeds, no chains, no band: full affine DP of every read end against every position of every
transcript, then the in-tree pairing / filtering rules) for
  * C1: every pair of the reference's bundled sample data (tests/golden/c1/), and
  * S1: 20 000 synthetic 2x100 bp pairs against a 300-gene / ~1300-transcript synthetic transcriptome (isoforms share exons, paralog families),
  * S2: 6 000 NOISY pairs against the same transcriptome (2 % substitutions, 0.4 % indels: four times and forty times S1's rates) — the banded,
    chain-guided DP of row a4 against the full DP where gaps are common.
tests/test_exhaustive.py holds the checker's and the HIP path's labels to these.  ~6 minutes on 8 cores:  python tests/golden/make_exhaustive.py"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import exh, fixtures
from salmon_amd import api, synth

S1 = dict(seed=41, n_genes=300, iso=5, n_pairs=20000, read_seed=5)
S2 = dict(n_pairs=6000, read_seed=77, sub_rate=0.02, indel_rate=0.004)


def s1_world(threads=4):
    tx = synth.Txome(seed=S1["seed"], n_genes=S1["n_genes"], iso_per_gene=S1["iso"], threads=threads)
    names, seqs, lens = tx.tables()
    idx = api.SalmonIndex.build_mem_raw(tx.n, names, seqs, lens, threads=threads)
    seq, off, tt, tp = tx.reads(S1["n_pairs"], read_len=100, seed=S1["read_seed"], threads=threads)
    raw = dict(zip(tx.names(), tx.seqs()))
    # the index's references: duplicates dropped, poly-A tails clipped (a reference is a prefix of its FASTA record)
    refs = [raw[n][:l] for n, l in zip(idx.ref_names(), idx.ref_lens())]
    return dict(tx=tx, idx=idx, seq=seq, off=off, n=S1["n_pairs"], refs=refs, truth=tt)


def s2_world(s1):
    seq, off, tt, tp = s1["tx"].reads(S2["n_pairs"], read_len=100, seed=S2["read_seed"], sub_rate=S2["sub_rate"], indel_rate=S2["indel_rate"], threads=4)
    return dict(s1, seq=seq, off=off, n=S2["n_pairs"], truth=tt)


def c1_world():
    d = fixtures.c1_load()
    idx = api.SalmonIndex.build_mem(d["names"], d["seqs"], threads=2)
    raw = dict(zip(d["names"], d["seqs"]))
    refs = [raw[n][:l].encode() for n, l in zip(idx.ref_names(), idx.ref_lens())]
    return dict(idx=idx, seq=d["seq"], off=d["off"], n=d["n"], refs=refs)


if __name__ == "__main__":
    thr = os.cpu_count() or 8
    out = {}
    s1 = s1_world()
    for tag, w in (("c1", c1_world()), ("s1", s1), ("s2", s2_world(s1))):
        t = time.time()
        lo, lt, ls, kind = exh.labels(w["refs"], w["seq"], w["off"], w["n"], api.quant_opts(), threads=thr)
        print("%s: %d pairs against %d transcripts (%d nt) in %.0f s; %d labels, %d unmapped, %d orphan-only" % (tag, w["n"], len(w["refs"]),
              sum(map(len, w["refs"])), time.time() - t, len(lt), int((kind == 0).sum()), int((kind == 2).sum())), flush=True)
        out.update({tag + "_off": lo.astype(np.uint32), tag + "_tid": lt.astype(np.uint32), tag + "_score": ls.astype(np.int16), tag + "_kind": kind})
    np.savez_compressed(os.path.join(HERE, "exhaustive_labels.npz"), **out)
