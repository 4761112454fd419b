"""[r5] normalizeAlphas + projectToPolytope (row a14) against the reference's own cluster code: include/salmon/internal/quant/TranscriptCluster.hpp
(projectToPolytope :46-102, merge) and ClusterForest.hpp (mergeClusters, updateCluster, getClusters) compiled from where they lie under /root/reference
into oracle/_ref/libpolytope_ref.so (oracle/ref_polytope_shim.cpp; boost::dynamic_bitset / disjoint_sets / Transcript stood in for under oracle/_stub/poly;
the 40 lines of normalizeAlphas, which live in the uncompilable SalmonUtils.cpp, are restated around those objects in the shim).  The checker's
orc_normalize_alphas — and the product's host implementation, which is bit-exact with it (tests/test_normalize.py) — gives the same projected counts:
the reference walks a cluster's members in the order its list splices left them, the checker in ascending id, so equal to rounding, not bit for bit.
Skipped where the library was not built."""
import ctypes as C, os
import numpy as np
import pytest
import orc
from test_normalize import clustered_classes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libpolytope_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libpolytope_ref.so not built (no /root/reference on this machine)")
    L = C.CDLL(path); vp = C.c_void_p
    L.ref_normalize_alphas.argtypes = [C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp]
    return L


@pytest.mark.parametrize("M,E,seed", [(50, 40, 1), (3000, 5000, 2), (40000, 90000, 3), (3000, 900, 4)])
def test_projected_counts_follow_the_reference_clusters(built, M, E, seed):
    L = _ref(); eq = clustered_classes(M, E, seed); rng = np.random.default_rng(seed + 100)
    lm = np.log(rng.random(M) * 50 + 1e-6)
    unseen = np.ones(M, bool); unseen[eq.tid] = False; lm[unseen] = -np.inf
    tc = rng.integers(0, 400, M).astype(np.uint64); uq = (tc * rng.random(M) * 0.6).astype(np.uint64)
    got = orc.normalize_alphas(M, eq, lm, uq, tc)
    off = np.ascontiguousarray(eq.off, np.uint64); tid = np.ascontiguousarray(eq.tid, np.uint32); cnt = np.ascontiguousarray(eq.count, np.uint64)
    want = np.zeros(M); ncl = C.c_uint64(0)
    L.ref_normalize_alphas(M, E, off.ctypes.data, tid.ctypes.data, cnt.ctypes.data, lm.ctypes.data, uq.ctypes.data, tc.ctypes.data, int(cnt.sum()), want.ctypes.data, C.byref(ncl))
    assert ncl.value < M                                                          # clusters really merged
    clamped = (want == tc) | (want == uq)
    assert clamped.any() and (~clamped & (want > 0)).any()                        # bound and free members both occur: the projection ran
    assert np.array_equal(got == 0, want == 0) and np.allclose(got, want, rtol=1e-9, atol=1e-9), np.abs(got - want).max()
