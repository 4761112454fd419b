"""[r4] The EM update (rows a15-a16): the checker's em_step with use_vbem = 0 against the reference's own single-threaded EMUpdate_
(src/inference/EMUtils.cpp — the update every bootstrap replicate runs), compiled from where it lies under /root/reference into
oracle/_ref/libem_ref.so (oracle/ref_em_shim.cpp; `make -C oracle ref`).  The combined weights are the checker's (orc_em_combined_weights: the
formula of CollapsedEMOptimizer.cpp:830-873); what is pinned is the update rule on them: per class the denominator, the single-transcript
short cut, the skipped classes whose denominator vanishes.  The reference adds into alphaOut class after class, the checker sums a transcript's
terms in its canonical order (SPEC §B): equal to rounding, not bit for bit.  The HIP kernels are bit-exact with the checker (tests/test_em.py).
Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc
from salmon_amd import api
from conftest import random_eq_classes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libem_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libem_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path)
    L.ref_em_update.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ref_truncate.argtypes = [C.c_void_p, C.c_uint32, C.c_double]; L.ref_truncate.restype = C.c_double
    return L


@pytest.mark.parametrize("M,E,zeros", [(64, 300, False), (3000, 20000, False), (3000, 20000, True)])
def test_em_step_follows_emupdate(built, M, E, zeros):
    L = _ref(); rng = np.random.default_rng(M + E + zeros)
    eq = random_eq_classes(M, E, seed=E, max_size=9)
    eff = rng.uniform(50, 5000, M); a0 = rng.uniform(0, 50, M)
    if zeros: a0[rng.random(M) < 0.6] = 0.0                                    # classes whose every transcript is at zero: skipped, their reads go nowhere
    o = api.em_opts(use_vbem=0); t = eq.table(); txp = api.make_txp_in(eff)
    O = orc.lib(); O.orc_em_combined_weights.argtypes = [C.c_void_p] * 4
    cw = np.zeros(len(eq.tid)); O.orc_em_combined_weights(C.byref(t), C.byref(txp), C.byref(o), cw.ctypes.data)
    off = np.ascontiguousarray(eq.off, np.uint64); tid = np.ascontiguousarray(eq.tid, np.uint32); cnt = np.ascontiguousarray(eq.count, np.uint64)
    alpha = a0.copy()
    for it in range(3):
        want = np.zeros(M); L.ref_em_update(E, off.ctypes.data, tid.ctypes.data, cw.ctypes.data, cnt.ctypes.data, M, alpha.ctypes.data, want.ctypes.data)
        got = orc.em_steps(eq, eff, alpha, 1, o)
        assert np.allclose(got, want, rtol=1e-11, atol=1e-9), (it, np.abs(got - want).max())
        if zeros and it == 0: assert want.sum() < float(cnt.sum())               # mass really was dropped with the dead classes
        alpha = want
    # truncateCountVector: what the optimiser does to the final alphas (CollapsedEMOptimizer.cpp:1017-1023)
    a = alpha.copy(); s = L.ref_truncate(a.ctypes.data, M, 1e-8); b = np.where(alpha <= 1e-8, 0.0, alpha)
    assert np.array_equal(a, b) and abs(s - b.sum()) < 1e-9 * max(1.0, b.sum())
