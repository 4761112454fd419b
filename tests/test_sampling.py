"""Bootstrap (a16) and Gibbs (a17): checker properties on CPU, GPU == checker bit for bit (-m gpu)."""
import numpy as np
import pytest
from salmon_amd import api
import orc
from conftest import random_eq_classes


def _problem(seed=1, M=300, E=2000):
    eq = random_eq_classes(M, E, seed=seed, max_size=6)
    eff = np.random.default_rng(seed).uniform(100, 3000, M)
    return eq, eff


def test_gamma_sampler_moments(built):
    import ctypes as C
    # sq_gamma_draw through the checker's Gibbs with a degenerate problem is awkward; test the moments via a
    # one-class-per-transcript table: counts become Gamma(count + prior, 1/(0.1 + effLen)) draws
    M = 4000
    off = np.arange(M + 1, dtype=np.uint64); tid = np.arange(M, dtype=np.uint32); w = np.ones(M); cnt = np.full(M, 7, np.uint64)
    eq = api.EqClasses(off, tid, w, cnt); eff = np.full(M, 9.9)
    out = orc.gibbs(eq, eff, np.full(M, 7.0), 1, 123, int(cnt.sum()), api.gibbs_opts(thinning_factor=1))
    # mu ~ Gamma(7 + 1, scale 0.1): mean 0.8, var 0.08; alpha = mu*eff*N/sum(mu*eff) -> mean 7, cv^2 = 1/8
    a = out[0]
    assert abs(a.mean() - 7.0) < 1e-9
    assert abs(a.var() / a.mean() ** 2 - 1 / 8) < 0.02


def test_oracle_bootstrap_is_a_resampling(built):
    eq, eff = _problem()
    N = int(eq.count.sum())
    bs = orc.bootstrap(eq, eff, 6, 99, N)
    point, _ = orc.em_optimize(eq, eff, None, api.em_opts(init_uniform=1))
    assert np.allclose(bs.sum(axis=1), N, rtol=1e-9)           # each replicate redistributes exactly N fragments
    assert not np.array_equal(bs[0], bs[1])
    big = point > 50
    assert np.all(np.abs(bs.mean(axis=0)[big] - point[big]) < 6 * np.sqrt(point[big]) + 0.25 * point[big])


def test_oracle_gibbs_conserves_and_tracks_point_estimate(built):
    eq, eff = _problem(seed=2)
    N = int(eq.count.sum())
    point, _ = orc.em_optimize(eq, eff, None, api.em_opts(init_uniform=1))
    g = orc.gibbs(eq, eff, point, 8, 7, N, api.gibbs_opts(thinning_factor=4))
    assert np.allclose(g.sum(axis=1), N, rtol=1e-6)
    big = point > 200
    assert np.corrcoef(g.mean(axis=0)[big], point[big])[0, 1] > 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("vb", [0, 1])
def test_gpu_bootstrap_bit_exact(built, vb):
    eq, eff = _problem(seed=3)
    N = int(eq.count.sum())
    o = api.em_opts(use_vbem=vb)
    assert np.array_equal(api.bootstrap(eq, eff, 3, 1234, N, o), orc.bootstrap(eq, eff, 3, 1234, N, o))


@pytest.mark.gpu
@pytest.mark.parametrize("nogamma", [0, 1])
def test_gpu_gibbs_bit_exact(built, nogamma):
    eq, eff = _problem(seed=4)
    N = int(eq.count.sum())
    point, _ = orc.em_optimize(eq, eff, None, api.em_opts(init_uniform=1))
    go = api.gibbs_opts(thinning_factor=3, no_gamma_draw=nogamma)
    assert np.array_equal(api.gibbs(eq, eff, point, 5, 42, N, go), orc.gibbs(eq, eff, point, 5, 42, N, go))
