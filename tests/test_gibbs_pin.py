"""[r5] Row a17 — the Gibbs sampler of configs[4] — against the reference's own `CollapsedGibbsSampler::sample` / `sampleRoundNonCollapsedMultithreaded_`, compiled
from src/inference/CollapsedGibbsSampler.cpp with the stand-ins of the optimiser's pin (oracle/ref_gibbs_shim.cpp -> oracle/_ref/libgibbs_ref.so; its
std::random_device redirected to a seeded counter so that the test is deterministic).  The checker (and the kernels, which equal it bit for bit) draw from
their own counter-based streams, so what is compared is the DISTRIBUTION of the samples: per-transcript means within a few standard errors, spreads within a
factor, the conserved total, zeros where a transcript is in no class — over the option sets that change the sampler's rule (VB per-transcript / per-nucleotide
prior, EM prior, no Gamma draw, counts not extrapolated: the prior and the thinning are what a wrong restatement would get wrong).  No GPU."""
import ctypes as C, os
import numpy as np
import pytest
from salmon_amd import api
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(M, E, seed, idle=4):
    """classes of 1-4 transcripts drawn evenly from all but the last `idle` ones (those stay in no class), normalised weights, 5-400 reads each"""
    rng = np.random.default_rng(seed); off = [0]; tid = []; w = []
    for _ in range(E):
        n = int(rng.integers(1, 5)); t = np.sort(rng.choice(M - idle, n, replace=False)); x = rng.random(n) + 0.05
        tid += list(t); w += list(x / x.sum()); off.append(len(tid))
    return api.EqClasses(np.array(off, np.uint64), np.array(tid, np.uint32), np.array(w), rng.integers(5, 400, E).astype(np.uint64))
REF = os.path.join(ROOT, "oracle", "_ref", "libgibbs_ref.so")


def _ref_gibbs(eq, eff, init, S, seed, N, use_vbem, per_txp, vb_prior, thinning, no_gamma, dont_extrapolate=0):
    if not os.path.exists(REF): pytest.skip("oracle/_ref/libgibbs_ref.so is built where /root/reference exists (make -C oracle ref)")
    L = C.CDLL(REF); L.ref_gibbs.restype = C.c_int
    L.ref_gibbs.argtypes = [C.c_uint64] + [C.c_void_p] * 4 + [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p]
    off = np.ascontiguousarray(eq.off, np.uint64); tid = np.ascontiguousarray(eq.tid, np.uint32); w = np.ascontiguousarray(eq.w, np.float64); cnt = np.ascontiguousarray(eq.count, np.uint64)
    a = np.ascontiguousarray(init, np.float64); e = np.ascontiguousarray(eff, np.float64); out = np.zeros((S, len(e)))
    rc = L.ref_gibbs(len(cnt), off.ctypes.data, tid.ctypes.data, w.ctypes.data, cnt.ctypes.data, len(e), a.ctypes.data, e.ctypes.data, N, use_vbem, per_txp, vb_prior, thinning, no_gamma, dont_extrapolate, S, seed, out.ctypes.data)
    assert rc == 0
    return out


@pytest.mark.parametrize("name,use_vbem,per_txp,vb_prior,no_gamma", [("VB, per-nucleotide prior", 1, 0, 1e-2, 0), ("VB, per-transcript prior (the default)", 1, 1, 1e-2, 0),
                                                                     ("VB, a larger per-transcript prior", 1, 1, 3.0, 0), ("EM", 0, 0, 1e-2, 0), ("no Gamma draw", 1, 0, 1e-2, 1)])
def test_gibbs_samples_follow_the_reference_samplers_distribution(built, name, use_vbem, per_txp, vb_prior, no_gamma):
    M, E, S, thin = 60, 260, 200, 8
    eq = _table(M, E, 11); rng = np.random.default_rng(5); eff = rng.uniform(150, 2500, M)
    N = int(eq.count.sum()); point, _ = orc.em_optimize(eq, eff, None, api.em_opts(init_uniform=1))
    go = api.gibbs_opts(thinning_factor=thin, no_gamma_draw=no_gamma, use_vbem=use_vbem, per_transcript_prior=per_txp, vb_prior=vb_prior)
    ref = np.concatenate([_ref_gibbs(eq, eff, point, S, seed, N, use_vbem, per_txp, vb_prior, thin, no_gamma) for seed in (7, 8)])
    chk = np.concatenate([orc.gibbs(eq, eff, point, S, seed, N, go) for seed in (7, 8)])
    assert np.allclose(ref.sum(axis=1), N, rtol=1e-9) and np.allclose(chk.sum(axis=1), N, rtol=1e-6), name          # extrapolated to the mapped fragments
    active = np.zeros(M, bool); active[np.asarray(eq.tid)] = True
    assert np.all(ref[:, ~active] == 0) and np.all(chk[:, ~active] == 0), name
    mr, mc, sr, sc = ref.mean(0), chk.mean(0), ref.std(0), chk.std(0)
    n_eff = ref.shape[0] / 4.0           # neighbouring samples of a chain are correlated: a quarter of them counted as independent
    se = np.sqrt((sr ** 2 + sc ** 2) / n_eff)
    z = np.abs(mr - mc) / np.maximum(se, 1e-9 * np.maximum(mr, 1.0))
    assert np.all(z[active] < 5.0), (name, float(z[active].max()), int(np.argmax(z)))
    big = active & (mr > 20); assert active.sum() == M - 4 and big.sum() > 40
    assert np.all((sc[big] > 0.6 * sr[big]) & (sc[big] < 1.6 * sr[big])), (name, (sc[big] / sr[big]).min(), (sc[big] / sr[big]).max())
    assert np.corrcoef(mr[active], mc[active])[0, 1] > 0.999, name


def test_counts_not_extrapolated(built):
    """--dontExtrapolateCounts is a flag of the sampler's caller in the product (the hard counts are what the kernel keeps); the reference's own switch returns them: integers that sum to
    the number of fragments in the table."""
    M, E = 40, 150; eq = _table(M, E, 3); eff = np.random.default_rng(2).uniform(200, 2000, M); N = int(eq.count.sum())
    point, _ = orc.em_optimize(eq, eff, None, api.em_opts(init_uniform=1))
    ref = _ref_gibbs(eq, eff, point, 10, 1, N, 1, 0, 1e-2, 4, 0, dont_extrapolate=1)
    assert np.all(ref == np.round(ref)) and np.all(ref.sum(axis=1) == N)
