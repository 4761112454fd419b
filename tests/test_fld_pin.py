"""[r4] The fragment-length distribution of the online model (row a12): the checker's FLD (oracle.cpp struct FLD — prior, the placement of the
Binomial(4, 0.5) kernel around an observed length, bin 0 kept empty, the minimum, pmf, cacheCMF / getLockedPMF) against the reference's own
src/model/FragmentLengthDistribution.cpp, compiled from where it lies under /root/reference into oracle/_ref/libfld_ref.so
(oracle/ref_fld_shim.cpp; `make -C oracle ref`).  Boost is absent here: its normal cdf / binomial pdf are stood in for by oracle/_stub (erfc, the
product formula), so the last bits of Boost's prior TABLE stay unpinned — everything the class does with the table is the reference's code.
The reference adds fragments one by one (any order, atomics); the checker adds a mini-batch's fragments of one length at once (SPEC §D3): the
same sums in another order, so the comparison is to 1e-9 in log space, not bit for bit.  Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double); up = C.POINTER(C.c_uint32)


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libfld_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libfld_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path)
    L.ref_fld_new.argtypes = [C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_double]; L.ref_fld_new.restype = C.c_void_p
    L.ref_fld_free.argtypes = [C.c_void_p]; L.ref_fld_add.argtypes = [C.c_void_p, up, C.c_uint64, C.c_double]; L.ref_fld_cache.argtypes = [C.c_void_p]
    L.ref_fld_pmf.argtypes = [C.c_void_p, dp, C.c_uint32]; L.ref_fld_cmf.argtypes = [C.c_void_p, dp, C.c_uint32]
    L.ref_fld_min.argtypes = [C.c_void_p]; L.ref_fld_min.restype = C.c_uint64; L.ref_fld_max.argtypes = [C.c_void_p]; L.ref_fld_max.restype = C.c_uint64
    L.ref_fld_mean.argtypes = [C.c_void_p]; L.ref_fld_mean.restype = C.c_double
    L.ref_eff_lengths.argtypes = [C.c_void_p, up, C.c_uint32, dp]; L.ref_fld_summary.argtypes = [C.c_void_p, dp, dp, up]
    L.ref_eval_log_cmf.argtypes = [C.c_void_p, dp, C.c_uint32]; L.ref_eval_log_cmf.restype = C.c_uint32
    L.ref_ambig_prob.argtypes = [C.c_void_p] + [C.c_int] * 3 + [C.c_int32] * 3; L.ref_ambig_prob.restype = C.c_double
    return L


def _orc():
    O = orc.lib()
    O.orc_fld_new.argtypes = [C.c_double, C.c_double]; O.orc_fld_new.restype = C.c_void_p; O.orc_fld_free.argtypes = [C.c_void_p]
    O.orc_fld_apply.argtypes = [C.c_void_p, up, C.c_double]; O.orc_fld_cache.argtypes = [C.c_void_p]
    O.orc_fld_pmf.argtypes = [C.c_void_p, dp, C.c_uint32]; O.orc_fld_cmf.argtypes = [C.c_void_p, dp, C.c_uint32]
    O.orc_fld_min.argtypes = [C.c_void_p]; O.orc_fld_min.restype = C.c_uint32
    O.orc_fld_eff_lengths.argtypes = [C.c_void_p, up, C.c_uint32, dp]
    O.orc_fld_ambig_prob.argtypes = [C.c_void_p] + [C.c_int] * 3 + [C.c_int32] * 3; O.orc_fld_ambig_prob.restype = C.c_double
    return O


def _vec(fn, h, n=1001):
    out = np.zeros(n); fn(h, out.ctypes.data_as(dp), n); return out


def test_fld_follows_the_reference_class(built):
    L = _ref(); O = _orc(); rng = np.random.default_rng(5)
    # the constructor call of ReadExperiment (alpha 1, max 1000, prior N(250, 25), kernel Binomial(4, 0.5), bin size 1): SalmonDefaults.hpp fragLenDist*
    for mu, sd, pop in ((250.0, 25.0, "library"), (250.0, 25.0, "edges"), (180.0, 40.0, "library")):
        r = L.ref_fld_new(1.0, 1000, mu, sd, 4, 0.5); o = O.orc_fld_new(mu, sd)
        assert L.ref_fld_max(r) == 1000
        # the prior alone
        pr, po = _vec(L.ref_fld_pmf, r), _vec(O.orc_fld_pmf, o)
        assert np.allclose(pr, po, rtol=0, atol=1e-9), np.abs(pr - po).max()
        assert L.ref_fld_min(r) == 1                                             # nothing observed yet: minVal() answers 1 (the checker: orc_state_finish's minV rule)
        # mini-batches of fragments, each with its own forgetting mass
        for b in range(6):
            n = int(rng.integers(50, 5000))
            if pop == "library": lens = np.clip(rng.normal(300, 60, n).round(), 0, 1400).astype(np.uint32)
            else: lens = rng.choice(np.array([0, 1, 2, 3, 4, 997, 998, 999, 1000, 1001, 1500, 300], np.uint32), n)      # both ends of the table: bin 0 stays empty, lengths over the maximum land in the last bin
            log_fm = float(-0.35 * b + rng.normal(0, 0.1))
            L.ref_fld_add(r, lens.ctypes.data_as(up), n, log_fm)
            cnt = np.bincount(np.minimum(lens, 1000), minlength=1001).astype(np.uint32)
            O.orc_fld_apply(o, cnt.ctypes.data_as(up), log_fm)
            pr, po = _vec(L.ref_fld_pmf, r), _vec(O.orc_fld_pmf, o)
            assert np.allclose(pr, po, rtol=0, atol=1e-9), (pop, b, np.abs(pr - po).max())
        assert L.ref_fld_min(r) == O.orc_fld_min(o)
        # burn-in: cacheCMF freezes pmf (renormalised) and cmf
        L.ref_fld_cache(r); O.orc_fld_cache(o)
        pr, po = _vec(L.ref_fld_pmf, r), _vec(O.orc_fld_pmf, o)
        cr, co = _vec(L.ref_fld_cmf, r), _vec(O.orc_fld_cmf, o)
        assert np.allclose(pr, po, rtol=0, atol=1e-9) and np.allclose(cr, co, rtol=0, atol=1e-9), (np.abs(pr - po).max(), np.abs(cr - co).max())
        assert abs(cr[-1]) < 1e-9                                                # the cached CMF ends at log 1
        # lengths beyond the table read its last entry on both sides
        big = np.zeros(1); L.ref_fld_pmf(r, big.ctypes.data_as(dp), 1)
        L.ref_fld_free(r); O.orc_fld_free(o)


def _trained(L, O, rng, n_batches=5):
    r = L.ref_fld_new(1.0, 1000, 250.0, 25.0, 4, 0.5); o = O.orc_fld_new(250.0, 25.0)
    for b in range(n_batches):
        n = int(rng.integers(2000, 9000)); lens = np.clip(rng.normal(280, 70, n).round(), 30, 1200).astype(np.uint32); log_fm = float(-0.3 * b)
        L.ref_fld_add(r, lens.ctypes.data_as(up), n, log_fm)
        cnt = np.bincount(np.minimum(lens, 1000), minlength=1001).astype(np.uint32); O.orc_fld_apply(o, cnt.ctypes.data_as(up), log_fm)
    return r, o


def test_effective_lengths_follow_distribution_utils(built):
    """Row a13: the effective lengths the online phase switches to at burn-in (ReadExperiment::updateTranscriptLengthsAtomic -> correctionFactorsFromMass +
    computeSmoothedEffectiveLengths) — the checker's compute_eff_lengths against the reference's two functions, on the prior alone and on a trained FLD."""
    L = _ref(); O = _orc(); rng = np.random.default_rng(9)
    ref_len = np.concatenate([np.arange(1, 1300), rng.integers(1, 100000, 4000)]).astype(np.uint32)
    for trained in (False, True):
        if trained: r, o = _trained(L, O, rng)
        else: r = L.ref_fld_new(1.0, 1000, 250.0, 25.0, 4, 0.5); o = O.orc_fld_new(250.0, 25.0)
        a = np.zeros(len(ref_len)); b = np.zeros(len(ref_len))
        L.ref_eff_lengths(r, ref_len.ctypes.data_as(up), len(ref_len), a.ctypes.data_as(dp)); O.orc_fld_eff_lengths(o, ref_len.ctypes.data_as(up), len(ref_len), b.ctypes.data_as(dp))
        assert np.allclose(a, b, rtol=0, atol=1e-9), (trained, np.abs(a - b).max(), int(np.abs(a - b).argmax()))
        assert np.all(np.exp(a[:20]) >= 1.0 - 1e-12)                             # a transcript shorter than every fragment keeps its length
        L.ref_fld_free(r); O.orc_fld_free(o)


def test_orphan_fragment_length_probability_follows_logcmfcache(built):
    """Rows a10-a12: what an orphan / single-end alignment gets for its unseen fragment length (LogCMFCache::getAmbigFragLengthProb).  Before burn-in a
    paired-end library reads LogCMFCache's own table, which evaluateLogCMF fills from a vector of LOG_EPSILON (it never copies the pmf it dumped:
    DistributionUtils.cpp:104-118) — the checker keeps that; single-end libraries and burned-in ones read the FLD's cmf."""
    L = _ref(); O = _orc(); rng = np.random.default_rng(13)
    r, o = _trained(L, O, rng)
    tab = np.zeros(1001); assert L.ref_eval_log_cmf(r, tab.ctypes.data_as(dp), 1001) == 1001
    eps = tab[0]; assert np.allclose(tab, eps + np.log(np.arange(1, 1002)), atol=1e-9)      # the cumulative sum of 1001 equal LOG_EPSILON entries
    def both(se, burned):
        worst = 0.0
        for _ in range(400):
            tlen = int(rng.integers(20, 3000)); rlen = int(rng.integers(25, 151)); pos = int(rng.integers(-20, tlen + 20)); fwd = int(rng.integers(0, 2))
            a = L.ref_ambig_prob(r, se, burned, fwd, pos, rlen, tlen); b = O.orc_fld_ambig_prob(o, se, burned, fwd, pos, rlen, tlen)
            if np.isinf(a) or np.isinf(b): assert a == b, (se, burned, fwd, pos, rlen, tlen, a, b)
            else: worst = max(worst, abs(a - b))
        return worst
    assert both(0, 0) < 1e-9 and both(1, 0) < 1e-9          # not burned in: paired-end reads the cache's table, single-end the live cmf
    L.ref_fld_cache(r); O.orc_fld_cache(o)
    assert both(0, 1) < 1e-9 and both(1, 1) < 1e-9          # burned in: the cached cmf
    L.ref_fld_free(r); O.orc_fld_free(o)


def test_fld_summary_of_meta_info_follows_samples_from_log_pmf(built):
    """Row a18: frag_length_mean / frag_length_sd of meta_info.json and the support of aux_info/fld.gz — the product's sq_write_fld_samples (host code) against
    distribution_utils::samplesFromLogPMF on the same distribution (the reference's draws come from the random device: only the summary is comparable)."""
    from salmon_amd import capi
    L = _ref(); O = _orc(); rng = np.random.default_rng(21); P = capi.lib()
    r, o = _trained(L, O, rng); L.ref_fld_cache(r)
    pmf = _vec(L.ref_fld_pmf, r)
    m = np.zeros(1); sd = np.zeros(1); sup = np.zeros(1, np.uint32); L.ref_fld_summary(r, m.ctypes.data_as(dp), sd.ctypes.data_as(dp), sup.ctypes.data_as(up))
    m2 = np.zeros(1); sd2 = np.zeros(1); sup2 = np.zeros(1, np.uint32)
    capi.check(P.sq_write_fld_samples(None, pmf.ctypes.data, int(L.ref_fld_min(r)), 1000, 0, 1, m2.ctypes.data_as(dp), sd2.ctypes.data_as(dp), sup2.ctypes.data_as(up)), "sq_write_fld_samples")
    assert abs(m[0] - m2[0]) < 1e-9 * m[0] and abs(sd[0] - sd2[0]) < 1e-7 * sd[0] and sup[0] == sup2[0] == 1001, (m, m2, sd, sd2, sup, sup2)
    assert 200 < m[0] < 350 and 20 < sd[0] < 120
    L.ref_fld_free(r); O.orc_fld_free(o)

