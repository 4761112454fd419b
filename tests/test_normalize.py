"""normalizeAlphas + projectToPolytope (a14): the product's host implementation (parallel lock-free union-find,
clusters on several threads) against the checker's sequential restatement — bit for bit, on clustered random classes."""
import numpy as np
import pytest
from salmon_amd import api
import orc


def clustered_classes(M, E, seed):
    rng = np.random.default_rng(seed)
    bounds = [0]
    while bounds[-1] < M:
        bounds.append(min(M, bounds[-1] + int(rng.integers(1, 25))))
    groups = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    labels, counts = [], []
    for _ in range(E):
        lo, hi = groups[int(rng.integers(0, len(groups)))]
        k = int(rng.integers(1, min(8, hi - lo) + 1))
        labels.append(np.sort(rng.choice(np.arange(lo, hi), size=k, replace=False)).astype(np.uint32)); counts.append(int(rng.integers(1, 500)))
    eq = api.EqClasses.alloc(E, sum(len(l) for l in labels))
    off = np.zeros(E + 1, np.uint64); off[1:] = np.cumsum([len(l) for l in labels])
    eq.off[:] = off; eq.tid[:] = np.concatenate(labels); eq.count[:] = np.array(counts, np.uint64)
    eq.w[:] = 1.0
    return eq


@pytest.mark.parametrize("M,E,seed", [(50, 40, 1), (3000, 5000, 2), (40000, 90000, 3)])
def test_normalize_alphas_matches_checker(built, M, E, seed):
    eq = clustered_classes(M, E, seed)
    rng = np.random.default_rng(seed + 100)
    lm = np.log(rng.random(M) * 50 + 1e-6)
    unseen = np.ones(M, bool); unseen[eq.tid] = False; lm[unseen] = -np.inf       # a transcript without fragments has no mass
    tc = rng.integers(0, 400, M).astype(np.uint64); uq = (tc * rng.random(M) * 0.6).astype(np.uint64)
    want = orc.normalize_alphas(M, eq, lm, uq, tc)
    for _ in range(3):          # the thread interleaving must not matter
        got = api.normalize_alphas(eq, lm, uq, tc)
        assert np.array_equal(got, want, equal_nan=True)
    assert not np.any(np.isnan(got)) and np.all(got >= 0) and got.sum() > 0


def test_normalize_alphas_rejects_bad_labels(built):
    eq = clustered_classes(100, 50, 9)
    eq.tid[3] = 100
    with pytest.raises(Exception):
        api.normalize_alphas(eq, np.zeros(100), np.zeros(100, np.uint64), np.ones(100, np.uint64))


def _py_normalize(M, off, tid, count, log_mass, uniq, total):
    """Plain-Python restatement of normalizeAlphas (SalmonUtils.cpp:461-529) + projectToPolytope
    (TranscriptCluster.hpp:46-102); clusters = connected components of the labels, members ascending."""
    import math
    parent = list(range(M))
    def find(x):
        while parent[x] != x: parent[x] = parent[parent[x]]; x = parent[x]
        return x
    for c in range(len(count)):
        for i in range(off[c] + 1, off[c + 1]):
            a, b = find(int(tid[off[c]])), find(int(tid[i]))
            if a != b: parent[max(a, b)] = min(a, b)
    members, hits = {}, {}
    for t in range(M): members.setdefault(find(t), []).append(t)
    for c in range(len(count)):
        if off[c + 1] > off[c]: r = find(int(tid[off[c]])); hits[r] = hits.get(r, 0) + int(count[c])
    proj = [0.0] * M
    for r, mem in members.items():
        ch = float(hits.get(r, 0))
        lcm = math.inf                                            # LOG_0 is +HUGE_VAL in salmon; logAdd treats it as "nothing"
        for t in mem:
            m = log_mass[t]
            if math.isinf(m): continue
            lcm = m if math.isinf(lcm) else max(lcm, m) + math.log1p(math.exp(-abs(lcm - m)))
        need = False
        for t in mem:
            if math.isinf(log_mass[t]) or ch <= 0: proj[t] = 0.0; continue
            proj[t] = math.exp((log_mass[t] - lcm) + math.log(ch))
            need |= proj[t] > float(total[t]) or proj[t] < float(uniq[t])
        if len(mem) > 1 and need:
            bound = [False] * len(mem); rnd = 0
            while True:
                ub = bd = 0.0
                for i, t in enumerate(mem):
                    if proj[t] > float(total[t]): proj[t] = float(total[t]); bound[i] = True
                    elif proj[t] < float(uniq[t]): proj[t] = float(uniq[t]); bound[i] = True
                    if bound[i]: bd += proj[t]
                    else: ub += proj[t]
                if abs(ub + bd - ch) <= 0.375e-10: break
                if ub == 0: bound = [False] * len(mem); ub, bd = bd, 0.0
                s = (ch - bd) / ub
                for i, t in enumerate(mem):
                    if not bound[i]: proj[t] *= s
                rnd += 1
                if rnd > 5000: break
    return np.array(proj)


def test_checker_normalize_alphas_matches_plain_python_restatement(built):
    M, E = 400, 700
    eq = clustered_classes(M, E, 21)
    rng = np.random.default_rng(5)
    lm = np.log(rng.random(M) * 50 + 1e-6)
    unseen = np.ones(M, bool); unseen[eq.tid] = False; lm[unseen] = np.inf
    tc = rng.integers(0, 400, M).astype(np.uint64); uq = (tc * rng.random(M) * 0.6).astype(np.uint64)
    want = _py_normalize(M, [int(x) for x in eq.off], eq.tid, eq.count, [float(x) for x in lm], uq, tc)
    got = orc.normalize_alphas(M, eq, lm, uq, tc)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-12)
    assert np.any(got != np.round(got)) and np.any((got == tc) | (got == uq))     # free and clamped members both occur
