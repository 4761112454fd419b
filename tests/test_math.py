"""sq_math.h (shared deterministic fp64 functions) against libm / scipy, via the checker's exports."""
import numpy as np
import math
from scipy.special import digamma
import orc


def _ulp(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / abs(np.nextafter(b, np.inf) - b)


def test_exp_log_within_2ulp(built):
    rng = np.random.default_rng(0)
    L = orc.lib()
    worst_e = worst_l = 0.0
    for x in rng.uniform(-60, 60, 20000):
        worst_e = max(worst_e, _ulp(L.orc_exp(x), math.exp(x)))
    for y in np.exp(rng.uniform(-40, 40, 20000)):
        worst_l = max(worst_l, _ulp(L.orc_log(y), math.log(y)))
    assert worst_e <= 2.0 and worst_l <= 2.0


def test_exp_log_edges(built):
    L = orc.lib()
    assert L.orc_exp(0.0) == 1.0
    assert L.orc_log(1.0) == 0.0
    assert L.orc_exp(-800.0) == 0.0
    assert math.isinf(L.orc_exp(800.0))
    assert L.orc_log(0.0) == -math.inf


def test_digamma(built):
    L = orc.lib()
    xs = np.concatenate([np.exp(np.random.default_rng(1).uniform(-22, 12, 4000)), [1.0, 0.5, 2.0, 1e-10, 1e7]])
    for x in xs:
        r = float(digamma(x))
        assert abs(L.orc_digamma(x) - r) <= 5e-15 * max(1.0, abs(r))


def test_log_add_identity_and_value(built):
    # salmon::math::logAdd (SalmonMath.hpp:55-67): LOG_0 = +inf is the identity
    L = orc.lib()
    assert L.orc_log_add(math.inf, -3.0) == -3.0
    assert L.orc_log_add(-3.0, math.inf) == -3.0
    assert abs(L.orc_log_add(-3.0, -4.0) - math.log(math.exp(-3) + math.exp(-4))) < 1e-15


def test_fld_prior_is_discretised_normal(built):
    # FragmentLengthDistribution.cpp:38-55: hist[i] = log(Phi((i+.5-mu)/sd) - Phi((i-.5-mu)/sd))
    from scipy.stats import norm
    L = orc.lib()
    h = np.zeros(1001); tot = __import__("ctypes").c_double()
    L.orc_fld_prior(250.0, 25.0, h.ctypes.data, __import__("ctypes").byref(tot))
    i = np.arange(150, 351)
    ref = np.log(norm.cdf(i + 0.5, 250, 25) - norm.cdf(i - 0.5, 250, 25))
    assert np.allclose(h[150:351], ref, rtol=0, atol=1e-9)
    assert abs(h[1000] - math.log(0.375e-10)) < 1e-12   # LOG_EPSILON floor where the cdf difference is exactly 0
    assert h[0] < -50                                    # far left tail is tiny but non-zero (erfc form)
    assert abs(tot.value) < 1e-6                         # total mass ~ 1 (alpha = 1)


def test_forgetting_mass_recurrence(built):
    # ForgettingMassCalculator.hpp:30-40: fm_i = fm_{i-1} + ff*log(i-1) - log(i^ff - 1), fm_1 = 0
    L = orc.lib()
    ff = 0.65
    fm = 0.0
    assert L.orc_forgetting_mass(ff, 0) == 0.0
    for b in range(1, 50):
        i = b + 1
        fm += ff * math.log(i - 1) - math.log(i ** ff - 1)
        assert abs(L.orc_forgetting_mass(ff, b) - fm) < 1e-12


def test_effective_lengths_closed_form(small_world):
    # a13: effLen(len) = len - E[fl | fl <= len] under the (prior) fragment-length law, the value at 1000 for longer
    # transcripts, `len` itself when that is below 1 (ReadExperiment.inl:62-94, DistributionUtils.cpp:9-55) — evaluated
    # here with numpy/scipy, independently of the checker's C++
    from scipy.stats import norm
    from salmon_amd import api
    w = small_world
    st = orc.OrcState(w["oidx"], api.quant_opts()); st.finish()          # no fragments: the FLD is the prior
    log_eff = st.model()[3]; st.free()
    i = np.arange(0, 1001)
    nm = norm.cdf(i + 0.5, 250, 25) - norm.cdf(i - 0.5, 250, 25)
    hist = np.where(nm != 0, np.log(np.where(nm != 0, nm, 1.0)), math.log(0.375e-10))
    lp = hist[1:1001] - np.logaddexp.reduce(hist[1:1001])                # dumpPMF over [1, 1000], renormalised
    pmf = np.zeros(1001); pmf[1:1000] = 100.0 * np.exp(lp[:999])         # i < maxVal: bin 1000 stays 0
    vals = np.cumsum(pmf * i); mult = np.cumsum(pmf)
    cf = np.where(mult > 0, vals / np.where(mult > 0, mult, 1.0), 0.0); cf[0] = 0.0
    lens = w["idx"].ref_lens().astype(np.int64)
    c = np.where(lens >= 1001, cf[1000], cf[np.minimum(lens, 1000)])
    eff = lens - c; eff = np.where(eff < 1.0, lens, eff)
    assert np.allclose(np.exp(log_eff), eff, rtol=1e-9, atol=0)
    assert np.all(eff[lens > 600] < lens[lens > 600] - 240) and np.all(eff[lens > 600] > lens[lens > 600] - 260)   # ~ len - 250
