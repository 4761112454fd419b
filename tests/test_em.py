"""EM / VBEM: CPU checker properties (no GPU) and GPU-vs-checker bit parity (-m gpu)."""
import numpy as np
import pytest
from salmon_amd import api
import orc
from conftest import random_eq_classes


def test_oracle_em_single_class_fixed_point(built):
    # one class {0,1} with equal weights and equal lengths: EM keeps a uniform split of the count
    eq = api.EqClasses(np.array([0, 2], np.uint64), np.array([0, 1], np.uint32), np.array([0.5, 0.5]), np.array([10], np.uint64))
    a, rep = orc.em_optimize(eq, np.array([100.0, 100.0]), opts=api.em_opts(use_vbem=0, init_uniform=1))
    assert np.allclose(a, [5.0, 5.0]) and rep["iters"] == 100


def test_oracle_em_conserves_counts(built):
    eq = random_eq_classes(500, 3000, seed=3)
    eff = np.random.default_rng(1).uniform(50, 3000, 500)
    a, rep = orc.em_optimize(eq, eff, opts=api.em_opts(use_vbem=0, init_uniform=1))
    assert abs(a.sum() - float(eq.count.sum())) < 1e-6 * float(eq.count.sum())
    assert rep["iters"] >= 100


def test_oracle_vbem_matches_scipy_reference(built):
    # independent numpy/scipy restatement of VBEMUpdate_ (CollapsedEMOptimizer.cpp:241-328), 3 steps
    from scipy.special import digamma
    M, E = 60, 200
    eq = random_eq_classes(M, E, seed=5)
    eff = np.random.default_rng(2).uniform(80, 2000, M)
    o = api.em_opts(init_uniform=1)
    cw = np.zeros(len(eq.tid))
    for c in range(E):
        a, b = int(eq.off[c]), int(eq.off[c + 1])
        x = float(eq.count[c]) * eq.w[a:b] / np.maximum(eff[eq.tid[a:b]], 1.0)
        cw[a:b] = x / x.sum()
    alpha = np.full(M, 100.0)
    for _ in range(3):
        ap = alpha + 1e-2
        th = np.where(ap > 1e-10, np.exp(digamma(ap) - digamma(ap.sum())), 0.0)
        out = np.zeros(M)
        for c in range(E):
            a, b = int(eq.off[c]), int(eq.off[c + 1])
            if b - a == 1:
                out[eq.tid[a]] += float(eq.count[c]); continue
            v = th[eq.tid[a:b]] * cw[a:b]
            out[eq.tid[a:b]] += float(eq.count[c]) * v / v.sum()
        alpha = out
    got = orc.em_steps(eq, eff, np.full(M, 100.0), 3, o)
    assert np.allclose(got, alpha, rtol=1e-12, atol=1e-12)


def test_oracle_em_matches_numpy_reference(built):
    # the same for EMUpdate_ (CollapsedEMOptimizer.cpp:178-234): alpha itself in place of expTheta, no prior
    M, E = 60, 200
    eq = random_eq_classes(M, E, seed=7)
    eff = np.random.default_rng(3).uniform(80, 2000, M)
    o = api.em_opts(init_uniform=1, use_vbem=0)
    cw = np.zeros(len(eq.tid))
    for c in range(E):
        a, b = int(eq.off[c]), int(eq.off[c + 1])
        x = float(eq.count[c]) * eq.w[a:b] / np.maximum(eff[eq.tid[a:b]], 1.0)
        cw[a:b] = x / x.sum()
    alpha = np.full(M, 100.0)
    for _ in range(4):
        out = np.zeros(M)
        for c in range(E):
            a, b = int(eq.off[c]), int(eq.off[c + 1])
            if b - a == 1:
                out[eq.tid[a]] += float(eq.count[c]); continue
            v = alpha[eq.tid[a:b]] * cw[a:b]
            if v.sum() > 0: out[eq.tid[a:b]] += float(eq.count[c]) * v / v.sum()
        alpha = out
    got = orc.em_steps(eq, eff, np.full(M, 100.0), 4, o)
    assert np.allclose(got, alpha, rtol=1e-12, atol=1e-12)


def test_canonical_sum_is_a_sum(built):
    x = np.random.default_rng(0).uniform(0, 1e6, 70001)
    s = orc.lib().orc_canonical_sum(x.ctypes.data, len(x))
    assert abs(s - float(np.sum(x))) < 1e-9 * float(np.sum(x))


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [0, 1])
@pytest.mark.parametrize("M,E", [(64, 300), (1000, 20000), (70000, 150000)])
def test_gpu_em_steps_bit_exact(built, vb, M, E):
    eq = random_eq_classes(M, E, seed=M + vb, max_size=12 if M > 64 else 6)
    eff = np.random.default_rng(4).uniform(50, 5000, M)
    a0 = np.random.default_rng(5).uniform(0, 50, M)
    o = api.em_opts(use_vbem=vb)
    want = orc.em_steps(eq, eff, a0, 7, o)
    got, rep = api.em_steps(eq, eff, a0, 7, o)
    assert np.array_equal(got, want), float(np.max(np.abs(got - want)))


@pytest.mark.gpu
@pytest.mark.parametrize("vb,uni", [(1, 0), (0, 1), (1, 1)])
def test_gpu_em_optimize_bit_exact(built, vb, uni):
    M, E = 3000, 40000
    eq = random_eq_classes(M, E, seed=17)
    eff = np.random.default_rng(6).uniform(50, 5000, M)
    proj = np.random.default_rng(7).uniform(0, 30, M)
    o = api.em_opts(use_vbem=vb, init_uniform=uni)
    want, wrep = orc.em_optimize(eq, eff, proj, o)
    got, grep = api.em_optimize(eq, eff, proj, o)
    assert grep["iters"] == wrep["iters"] and grep["converged"] == wrep["converged"]
    assert np.array_equal(got, want)
    assert grep["max_rel_diff"] == wrep["max_rel_diff"]


@pytest.mark.gpu
def test_gpu_em_eqclass_mode_and_degenerate(built):
    # `salmon quant -e` seam (SalmonQuantifyAlignments.cpp:1407-1441): weights verbatim, uniform init;
    # plus empty-ish inputs: a transcript in no class, a class of size 1
    off = np.array([0, 1, 3, 6], np.uint64); tid = np.array([2, 0, 1, 0, 2, 4], np.uint32)
    w = np.array([1.0, 0.3, 0.7, 0.2, 0.5, 0.3]); cnt = np.array([5, 10, 7], np.uint64)
    eq = api.EqClasses(off, tid, w, cnt)
    eff = np.array([100.0, 200.0, 50.0, 10.0, 0.5])
    o = api.em_opts(eq_class_mode=1, init_uniform=1)
    want, _ = orc.em_optimize(eq, eff, None, o)
    got, _ = api.em_optimize(eq, eff, None, o)
    assert np.array_equal(got, want) and got[3] == 0.0


def _degenerate_case():
    # class 1 carries all-zero file weights: its combined weights are 0 * inf = NaN, markDegenerateClasses (CollapsedEMOptimizer.cpp:
    # 330-394) skips the NaN terms, finds denom = 0 and drops the class; without that the NaNs would reach every alpha
    off = np.array([0, 1, 3, 6, 8], np.uint64); tid = np.array([2, 0, 1, 0, 2, 4, 1, 3], np.uint32)
    w = np.array([1.0, 0.0, 0.0, 0.2, 0.5, 0.3, 0.6, 0.4]); cnt = np.array([5, 10, 7, 9], np.uint64)
    return api.EqClasses(off, tid, w, cnt), np.array([100.0, 200.0, 50.0, 10.0, 0.5])


@pytest.mark.parametrize("vb", [0, 1])
def test_oracle_marks_degenerate_classes(built, vb):
    eq, eff = _degenerate_case()
    a, rep = orc.em_optimize(eq, eff, None, api.em_opts(eq_class_mode=1, init_uniform=1, use_vbem=vb))
    assert rep["num_degenerate"] == 1 and np.all(np.isfinite(a))
    assert abs(a.sum() - (5 + 7 + 9)) < (1e-6 if not vb else 0.5)      # the dropped class's 10 fragments take no part


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [0, 1])
def test_gpu_marks_degenerate_classes_like_the_checker(built, vb):
    eq, eff = _degenerate_case()
    o = api.em_opts(eq_class_mode=1, init_uniform=1, use_vbem=vb)
    want, wrep = orc.em_optimize(eq, eff, None, o)
    got, grep = api.em_optimize(eq, eff, None, o)
    assert grep["num_degenerate"] == wrep["num_degenerate"] == 1 and grep["iters"] == wrep["iters"] and np.array_equal(got, want)
    eq2 = random_eq_classes(800, 5000, seed=9)                           # nothing to drop in an ordinary table
    _, r2 = api.em_optimize(eq2, np.random.default_rng(3).uniform(50, 3000, 800), None, api.em_opts(init_uniform=1, use_vbem=vb))
    assert r2["num_degenerate"] == 0


@pytest.mark.gpu
def test_gpu_alternative_init_mode_matches_checker(built):
    # --alternativeInitMode / --meta (CollapsedEMOptimizer.cpp:790-792, 817-818): the online estimate is mixed with
    # (uniqueCount + 0.5) * 1e-3 * effLen instead of the uniform abundance
    M, E = 3000, 20000
    eq = random_eq_classes(M, E, seed=21); rng = np.random.default_rng(22)
    eff = rng.uniform(50, 3000, M); proj = rng.gamma(0.3, 200.0, M); uq = rng.integers(0, 50, M).astype(np.uint64)
    o = api.em_opts(alt_init_mode=1, min_iter=3, max_iter=3)               # three steps: the starting point still shows
    want, wrep = orc.em_optimize(eq, eff, proj, o, unique=uq)
    got, grep = api.em_optimize(eq, eff, proj, o, unique=uq)
    assert grep["iters"] == wrep["iters"] == 3 and np.array_equal(got, want)
    full_c, rc = orc.em_optimize(eq, eff, proj, api.em_opts(alt_init_mode=1), unique=uq)
    full_g, rg = api.em_optimize(eq, eff, proj, api.em_opts(alt_init_mode=1), unique=uq)
    assert rc["iters"] == rg["iters"] and np.array_equal(full_c, full_g)
    plain, _ = orc.em_optimize(eq, eff, proj, api.em_opts(min_iter=3, max_iter=3))
    assert not np.array_equal(plain, want)                                # the option changes the starting point
    nouq, _ = orc.em_optimize(eq, eff, proj, o)                           # without unique counts the mode has nothing to use
    assert np.array_equal(nouq, plain)


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [0, 1])
def test_gpu_em_giant_class_and_hot_transcript(built, vb):
    # one class wider than the k_class LDS chunk (2048 labels) stands alone in its block; transcript 0
    # sits in every class (deep blocked-64 plan); a block of single-label classes exercises the -count path
    M, E = 5000, 9000
    rng = np.random.default_rng(23)
    labs = [np.arange(3000, dtype=np.uint32)]                                   # the giant class
    for c in range(1, E):
        if c % 7 == 0: labs.append(np.array([rng.integers(0, M)], np.uint32))
        else: labs.append(np.unique(np.concatenate([[0], rng.integers(1, M, rng.integers(1, 9))]).astype(np.uint32)))
    cnt = np.array([len(l) for l in labs]); off = np.zeros(E + 1, np.uint64); off[1:] = np.cumsum(cnt)
    tid = np.concatenate(labs); x = rng.random(len(tid)) + 0.05
    w = x / np.repeat(np.add.reduceat(x, off[:-1].astype(np.int64)), cnt)
    eq = api.EqClasses(off, tid, w, rng.integers(1, 300, E).astype(np.uint64))
    eff = rng.uniform(50, 5000, M); a0 = rng.uniform(0, 50, M)
    o = api.em_opts(use_vbem=vb)
    want = orc.em_steps(eq, eff, a0, 5, o)
    got, _ = api.em_steps(eq, eff, a0, 5, o)
    assert np.array_equal(got, want), float(np.max(np.abs(got - want)))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(no_rich_eq_classes=1), dict(per_transcript_prior=0, vb_prior=1e-5), dict(vb_prior=1.0),
    dict(use_vbem=0, no_rich_eq_classes=1),
                                dict(rel_diff_tolerance=0.001, max_iter=300), dict(min_iter=10, max_iter=40), dict(num_required_fragments=1000.0)],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_gpu_em_option_variants_bit_exact(built, kw):
    M, E = 2000, 15000
    eq = random_eq_classes(M, E, seed=31)
    eff = np.random.default_rng(8).uniform(50, 5000, M)
    proj = np.random.default_rng(9).uniform(0, 30, M)
    o = api.em_opts(**kw)
    want, wrep = orc.em_optimize(eq, eff, proj, o)
    got, grep = api.em_optimize(eq, eff, proj, o)
    assert grep["iters"] == wrep["iters"] and grep["converged"] == wrep["converged"]
    assert np.array_equal(got, want)
