"""[r5] The default optimiser (row a15) against the reference's own code: src/inference/CollapsedEMOptimizer.cpp compiled from where it lies under
/root/reference into oracle/_ref/libvbem_ref.so (oracle/ref_vbem_shim.cpp; `make -C oracle ref`; TBB / Boost / spdlog / ReadExperiment stood in for
under oracle/_stub/vbem, digamma = the checker's sq_digamma).
  * one serial VBEMUpdate_ (:104-171) = the checker's em_step with use_vbem = 1 on the same combined weights and priors;
  * the whole CollapsedEMOptimizer::optimize (:732-1035) = the checker's em_optimize: initialisation from the projected counts, combined weights,
    markDegenerateClasses, VBEM (and EM) updates over the class vector, the convergence rule and its iteration count, the final cut-off — in the
    default configuration and with every option of the optimiser the path has.
The reference adds into alphaOut class after class, the checker sums a transcript's terms in its canonical order (SPEC §B): equal to rounding, not bit
for bit.  The HIP kernels are bit-exact with the checker (tests/test_em.py).  Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc
from salmon_amd import api
from conftest import random_eq_classes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libvbem_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libvbem_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path); vp = C.c_void_p
    L.ref_vbem_update.argtypes = [C.c_uint64, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]
    L.ref_optimize.argtypes = [C.c_uint64, vp, vp, vp, vp, C.c_uint32, vp, vp, vp] + [C.c_int] * 6 + [C.c_double] * 3 + [C.c_uint32, vp, vp]
    L.ref_optimize.restype = C.c_int64
    return L


def _arrays(eq):
    return (np.ascontiguousarray(eq.off, np.uint64), np.ascontiguousarray(eq.tid, np.uint32), np.ascontiguousarray(eq.w, np.float64), np.ascontiguousarray(eq.count, np.uint64))


@pytest.mark.parametrize("M,E,zeros", [(64, 300, False), (3000, 20000, False), (3000, 20000, True)])
def test_vbem_step_follows_vbemupdate(built, M, E, zeros):
    L = _ref(); rng = np.random.default_rng(7 * M + E + zeros)
    eq = random_eq_classes(M, E, seed=E + 1, max_size=9); off, tid, w, cnt = _arrays(eq)
    eff = rng.uniform(50, 5000, M); a0 = rng.uniform(0, 50, M)
    if zeros: a0[rng.random(M) < 0.6] = 0.0
    for per_txp in (0, 1):
        o = api.em_opts(use_vbem=1, per_transcript_prior=per_txp); t = eq.table(); txp = api.make_txp_in(eff)
        O = orc.lib(); O.orc_em_combined_weights.argtypes = [C.c_void_p] * 4
        cw = np.zeros(len(eq.tid)); O.orc_em_combined_weights(C.byref(t), C.byref(txp), C.byref(o), cw.ctypes.data)
        prior = np.full(M, o.vb_prior) if per_txp else o.vb_prior * eff                # populatePriorAlphas_ (:82-99)
        alpha = a0.copy()
        for it in range(3):
            want = np.zeros(M); th = np.zeros(M)
            L.ref_vbem_update(E, off.ctypes.data, tid.ctypes.data, cw.ctypes.data, cnt.ctypes.data, M, prior.ctypes.data, alpha.ctypes.data, want.ctypes.data, th.ctypes.data)
            got = orc.em_steps(eq, eff, alpha, 1, o)
            assert np.allclose(got, want, rtol=1e-11, atol=1e-9), (per_txp, it, np.abs(got - want).max())
            assert (th[alpha + prior <= 1e-10] == 0).all()
            alpha = want


def gene_classes(M, E, seed, iso=5):
    """Classes inside genes of `iso` isoforms with nearly equal weights: the ambiguous kind that keeps the optimiser going well past its minimum of 100 iterations."""
    rng = np.random.default_rng(seed); G = M // iso; g = rng.integers(0, G, E); masks = rng.integers(1, 1 << iso, E); off = [0]; tid = []; w = []
    for c in range(E):
        ts = [int(g[c]) * iso + j for j in range(iso) if (int(masks[c]) >> j) & 1]; x = rng.random(len(ts)) * 0.2 + 0.9; x /= x.sum()
        if c % 97 == 5: x[:] = 0.0                                                  # a class without weight: its combined weights are NaN and markDegenerateClasses drops it
        tid += ts; w += list(x); off.append(len(tid))
    return api.EqClasses(np.array(off, np.uint64), np.array(tid, np.uint32), np.array(w), rng.integers(1, 400, E).astype(np.uint64))


CASES = [dict(), dict(use_vbem=0), dict(per_transcript_prior=1), dict(init_uniform=1), dict(no_rich_eq_classes=1), dict(alt_init_mode=1), dict(eq_class_mode=1),
         dict(vb_prior=1e-5), dict(use_vbem=0, init_uniform=1), dict(num_required_fragments=2000.0)]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_optimize_follows_the_reference_optimiser(built, case):
    """CollapsedEMOptimizer::optimize itself, run to its own convergence: the iteration count the reference logs and the alphas it leaves in the
    transcripts equal the checker's em_optimize."""
    L = _ref(); kw = CASES[case]; M, E = 2500, 14000; rng = np.random.default_rng(100 + case)
    eq = gene_classes(M, E, 31 + case); off, tid, w, cnt = _arrays(eq)
    eff = rng.uniform(300, 4000, M); eff[rng.random(M) < 0.02] = 0.7               # effective lengths below one are clamped in the weights (:846-848)
    proj = np.zeros(M); np.add.at(proj, tid, np.repeat(cnt.astype(np.float64), np.diff(off).astype(np.int64)) * w)     # something like the online estimates
    proj[rng.random(M) < 0.3] = 0.0                                               # and transcripts the online stage left at zero: degenerate classes appear
    uniq = rng.integers(0, 40, M).astype(np.uint64)
    o = api.em_opts(**kw)
    got, rep = orc.em_optimize(eq, eff, proj, o, unique=uniq)
    want = np.zeros(M); nvalid = C.c_uint64(0)
    it = L.ref_optimize(E, off.ctypes.data, tid.ctypes.data, w.ctypes.data, cnt.ctypes.data, M, proj.ctypes.data, uniq.ctypes.data, eff.ctypes.data,
                        int(o.use_vbem), int(o.per_transcript_prior), int(o.init_uniform), int(o.eq_class_mode), int(o.no_rich_eq_classes), int(o.alt_init_mode),
                        o.vb_prior, o.num_required_fragments, o.rel_diff_tolerance, o.max_iter, want.ctypes.data, C.byref(nvalid))
    assert it >= 100 and it == rep["iters"], (kw, it, rep["iters"])
    assert E - nvalid.value == rep["num_degenerate"], (kw, E - nvalid.value, rep["num_degenerate"])
    # hundreds of iterations along the flat directions of an ambiguous likelihood amplify the last-bit differences of the two summation orders: the
    # iteration count is the sharp statement, the alphas agree to 1e-9 of the transcript's gene-sized scale
    rel = np.abs(got - want) / np.maximum(1.0, np.maximum(np.abs(got), np.abs(want)))
    assert np.array_equal(got == 0, want == 0) and rel.max() < 1e-9, (kw, rel.max())
    assert case != 0 or (rep["num_degenerate"] > 0 and it > 150), (rep["num_degenerate"], it)   # the default case really has dead classes and runs past minIter
