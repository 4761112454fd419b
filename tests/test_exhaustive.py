"""Independent anchor for rows a1-a4 (seeding, uni-MEMs, chaining, joining, selective-alignment score), which live in pufferfish and have no
reference vectors: the labels of an EXHAUSTIVE aligner (oracle/exhaustive.cpp — no index, no seeds, no chains, no band: every read end
against every position of every transcript by full affine DP with salmon's selective-alignment scoring, then the in-tree pairing / filtering
rules), committed as tests/golden/exhaustive_labels.npz by tests/golden/make_exhaustive.py, for
  C1: all 10 000 pairs of the reference's bundled sample data,   S1: 20 000 synthetic 2x100 pairs against ~1300 synthetic transcripts,
  S2: 6 000 noisy pairs (2 % substitutions, 0.4 % indels) against the same transcripts.
The checker's labels (CPU) and the HIP path's (GPU) are held to them: C1 — every fragment; S1 — >= 99.9 % of the fragments for which the
exhaustive aligner finds a concordant pair, >= 99.3 % of all fragments (measured 99.97 % / 99.44 %: the gap is ONE class — a pair candidate
whose mate overhangs a clipped poly-A tail fails validation and the mapping is dropped, SalmonQuantify.cpp:1524-1529, where the exhaustive
tool falls back to the orphan), and identical alignment scores on every (fragment, transcript) both name; S2 — all 16 616 shared scores
equal (the banded, chain-guided DP finds the full DP's optimum wherever it aligns at all), 96.0 % of the label sets (97.0 % where a concordant pair
exists): errors closer than a k-mer leave ends without a seed, which the seed-and-extend scheme — the reference's as much as this one — cannot map.  oracle/SPEC.md §X lists the
disagreement classes with their counts."""
import os, sys
import numpy as np
import pytest
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import exh, orc
from salmon_amd import api

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exhaustive_labels.npz"))


@pytest.fixture(scope="module")
def worlds(built):
    import make_exhaustive as mk
    s1 = mk.s1_world()
    return dict(c1=mk.c1_world(), s1=s1, s2=mk.s2_world(s1))


def _check(tag, w, read_off, aln, floor, floor_pairs=None):
    lo = G[tag + "_off"].astype(np.uint64); lt = G[tag + "_tid"]; ls = G[tag + "_score"]; kind = G[tag + "_kind"]
    assert len(lo) == w["n"] + 1
    c = exh.compare(lo, lt, read_off, aln["tid"])
    assert c["agreement"] >= floor, {k: v for k, v in c.items() if k != "examples"}
    if floor_pairs is not None:   # fragments with a concordant pair of valid end alignments: what seeding + chaining + joining + scoring must find
        pairs = np.flatnonzero(kind == 1); bad = {e[0] for e in exh.compare(lo, lt, read_off, aln["tid"], max_examples=1 << 30)["examples"]}
        agree = 1.0 - len([f for f in pairs if int(f) in bad]) / max(1, len(pairs))
        assert len(pairs) > 0.9 * w["n"] and agree >= floor_pairs, (len(pairs), agree)
        c["pair_agreement"] = agree
    # wherever both name a transcript the alignment score is the same number: the heuristic found the optimal alignment
    differs = shared = 0
    for f in range(w["n"]):
        a = dict(zip(lt[int(lo[f]):int(lo[f + 1])].tolist(), ls[int(lo[f]):int(lo[f + 1])].tolist()))
        for x in aln[int(read_off[f]):int(read_off[f + 1])]:
            t = int(x["tid"])
            if t in a and (x["mate_status"] == 3) == (kind[f] == 1):      # a pair's sum against a pair's sum, an orphan's score against an orphan's
                shared += 1
                differs += a[t] != int(x["score"]) + (int(x["mate_score"]) if x["mate_status"] == 3 else 0)
    assert shared > w["n"] and differs <= (0 if tag == "s2" else shared // 2000), (shared, differs)   # S2 (noisy reads: gaps are common): every one of the 16 616 shared scores is the full DP's optimum
    return c


def test_golden_labels_are_what_the_exhaustive_aligner_produces(worlds):
    # a live run on a slice of each set (the whole of it takes minutes: make_exhaustive.py)
    for tag, k in (("c1", 150), ("s1", 40), ("s2", 30)):
        w = worlds[tag]
        lo, lt, ls, kind = exh.labels(w["refs"], w["seq"], w["off"], k, api.quant_opts(), threads=os.cpu_count() or 8)
        glo = G[tag + "_off"]
        assert np.array_equal(lo, glo[:k + 1].astype(np.uint64)) and np.array_equal(lt, G[tag + "_tid"][:int(glo[k])]) and np.array_equal(ls, G[tag + "_score"][:int(glo[k])])
        assert np.array_equal(kind, G[tag + "_kind"][:k])


def test_checker_labels_agree_with_the_exhaustive_aligner(worlds):
    res = {}
    for tag, floor, fp in (("c1", 1.0, 1.0), ("s1", 0.993, 0.999), ("s2", 0.955, 0.965)):
        w = worlds[tag]; oidx = orc.OrcIndex(w["idx"])
        rb = api.make_read_batch(w["seq"], w["off"], w["n"], paired=True)
        ro, aln, mt, st = orc.map_batch(oidx, api.quant_opts(), rb, threads=os.cpu_count() or 8)
        res[tag] = _check(tag, w, ro, aln, floor, fp)
    assert res["c1"]["equal"] == 10000


@pytest.mark.gpu
def test_hip_labels_agree_with_the_exhaustive_aligner(worlds):
    for tag, floor, fp in (("c1", 1.0, 1.0), ("s1", 0.993, 0.999), ("s2", 0.955, 0.965)):
        w = worlds[tag]; w["idx"].to_device(0)
        ctx = api.QuantContext(w["idx"], api.quant_opts(), device=0, max_batch_reads=max(4096, w["n"]))
        ro, aln, mt, st = ctx.map_batch(api.make_read_batch(w["seq"], w["off"], w["n"], paired=True))
        _check(tag, w, ro, aln, floor, fp)
        ctx.free()
